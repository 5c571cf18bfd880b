"""The library says what it was built from (pf_build_info, ABI 4), and a fallen-back build cannot pass for the
intended one.

``phyloformer_amd/build.py`` compiles pf_lib.hip with ``-mllvm -amdgpu-sched-strategy=iterative-ilp`` (1-7 % faster
kernels than hipcc's default strategy).  A hipcc that cannot do that used to get a silent retry with the default
strategy; now the build FAILS unless ``PF_ALLOW_SCHED_FALLBACK=1``, and a library built that way reports
``sched_fallback: true`` - which bench.py copies onto its line, and whose other ``kernel_hash`` makes the committed
PMC traffic figures ``null``.  hipcc cross-compiles without a GPU: this runs in the build container.
"""
import io
import json
import os
import stat
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_build_info_of_the_product_library():
    from phyloformer_amd import build, engine
    build.build()
    info = engine.build_info()
    assert info["abi"] == engine.ABI_VERSION == 5 and info["arch"] == "gfx950"
    assert info["sched_strategy"] == "iterative-ilp" and info["sched_fallback"] is False
    assert "amdgpu-sched-strategy=iterative-ilp" in info["flags"]["pf_lib.hip"]
    assert "sched-strategy" not in info["flags"]["pf_precise.hip"]          # its own unit: hipcc crashes on it otherwise
    assert info["source_hash"] == build.source_hash() and info["kernel_hash"] == build.kernel_hash()
    assert "HIP" in info["hipcc"] and "clang" in info["hipcc"]


@pytest.fixture(scope="module")
def fallback_lib(tmp_path_factory):
    """A hipcc that dies on the scheduling flag (as ROCm 7.2's did on one k_rowfin variant in round 3)."""
    from phyloformer_amd import build
    d = tmp_path_factory.mktemp("fallback")
    real = build.hipcc_path()
    fake = d / "hipcc"
    fake.write_text(f'#!/bin/bash\nfor a in "$@"; do case "$a" in *amdgpu-sched-strategy*) echo "fake hipcc: scheduler crash" >&2; exit 70;; esac; done\n'
                    f'exec {real} "$@"\n')
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    old = {k: os.environ.get(k) for k in ("HIPCC", "PF_ALLOW_SCHED_FALLBACK")}
    os.environ["HIPCC"] = str(fake)
    os.environ.pop("PF_ALLOW_SCHED_FALLBACK", None)
    try:
        with pytest.raises(RuntimeError, match="PF_ALLOW_SCHED_FALLBACK"):
            build.build(force=True, out=str(d / "refused.so"))                 # loud: no library without the opt-in
        assert not (d / "refused.so").exists()
        os.environ["PF_ALLOW_SCHED_FALLBACK"] = "1"
        lib = build.build(force=True, out=str(d / "libfallback.so"))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return lib


def test_fallback_build_is_marked_in_the_library_and_on_the_bench_line(fallback_lib, tmp_path):
    from phyloformer_amd import build, engine
    info = engine.build_info(fallback_lib)
    assert info["sched_fallback"] is True and info["sched_strategy"] == "default"
    assert "sched-strategy" not in info["flags"]["pf_lib.hip"]
    assert info["kernel_hash"] != build.kernel_hash() and info["source_hash"] == build.source_hash()

    # ... and on the line bench.py prints for an engine that sits on that library
    import bench
    from helpers.fake_engine import FakeEngine, FakeWeights

    class OnFallbackLibrary(FakeEngine):
        def build_info(self):
            return info

    args = bench.parse_args(["--steps", "2", "--warmup", "1", "--batch", "2", "--n-seqs", "6", "--n-sites", "45", "--no-power",
                             "--no-parity", "--no-configs", "--no-cpu-baseline", "--shard", "alignments"])
    out = io.StringIO()
    bench.run(args, 0, 1, 0, None, lambda dev: OnFallbackLibrary(0, []), FakeWeights(), out=out)
    line = json.loads(out.getvalue())
    assert line["config"]["build"]["sched_fallback"] is True and line["config"]["build"]["sched_strategy"] == "default"
    # the committed PMC counters were taken with another kernel_hash: not reported as this library's
    assert line["roofline"]["traffic"] is None and "stale" in line["roofline"]["traffic_source"]
