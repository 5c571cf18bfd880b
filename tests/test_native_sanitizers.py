"""AddressSanitizer + UBSan build of the native HOST code, fuzzed.

``csrc/pf_hostio.cpp`` parses untrusted files in place of /root/reference/phyloformer/data.py:11-31 and writes the
PHYLIP text of infer_alns.py:14-25; ``csrc/pf_host_prep.h`` is the host half of the engine (fp16 fragment packing,
LayerNorm folding, the residue-pair table, the shape-only launch plans).  Both are plain C++: this test compiles them
with ``g++ -fsanitize=address,undefined -fno-sanitize-recover`` into a test-only library and drives it with
hypothesis from a child process that has libasan preloaded (``tests/native/fuzz_host.py``): truncated files, CR/LF,
0xFF and NUL bytes, buffers exactly as large as declared (an overrun lands in a red zone), `idx_cap` / `max_seqs` /
`cap` one short, missing files and directories, 64 threads on 42 files - against the Python mirrors of the reference
(``fasta.py``, ``phylip.py``).  No GPU; GPU sanitizers are not available on the pool.
"""
import os
import shutil
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    out = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.mark.skipif(shutil.which("g++") is None or _tool("libasan.so") is None, reason="g++ / libasan not available")
def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    lib = str(tmp_path / "libpf_host_asan.so")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-Wall", "-Wextra", "-Werror",
           os.path.join(REPO, "phyloformer_amd", "csrc", "pf_hostio.cpp"),
           os.path.join(REPO, "tests", "native", "pf_host_prep_shim.cpp"), "-o", lib, "-lpthread"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    env = dict(os.environ, LD_PRELOAD=_tool("libasan.so"), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", PYTHONPATH=REPO)
    run = subprocess.run([sys.executable, os.path.join(REPO, "tests", "native", "fuzz_host.py"), lib, "150"], env=env,
                         capture_output=True, text=True, timeout=900)
    tail = (run.stdout + run.stderr)[-4000:]
    assert run.returncode == 0 and "fuzz_host: clean" in run.stdout, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
