"""Build ``libphyloformer_amd.so`` (HIP, gfx950) in-tree.

    python -m phyloformer_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU, so this also runs in the build container.
The shared object lands next to this file and is git-ignored; it travels to
the GPU box with the working-tree snapshot.

Every translation unit is compiled on its own and the objects are linked together with a generated
``pf_build_info.cpp``: the library can say which compiler, which flags (including the scheduling strategy that
REALLY compiled ``pf_lib.hip``) and which sources it was built from (``pf_build_info()``, ABI 4) - ``bench.py``
copies that onto its JSON line.
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
from typing import Dict, List, Optional

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libphyloformer_amd.so")
ARCH = "gfx950"

# -amdgpu-sched-strategy=iterative-ilp: with the default (max-occupancy) strategy hipcc sinks the software-
# pipelined LDS fragment reads of k_main's hidden loop down to their first use (read - wait - MFMA, one at a
# time, a single fragment buffer) whenever code outside the loop changes - +7 % on the loop between two builds
# with identical loop source (round 3); the iterative ILP strategy keeps the read pairs
# ahead of the MFMAs and is 1.3 % faster on the forward (k_colstats gains too).
SCHED = ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
COMMON = ["-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# source -> extra flags.  pf_precise.hip (float64 kernels) and pf_mha.hip (softmax attention operator) are their own
# translation units with the default strategy: iterative-ilp crashes hipcc's register allocator on them (ROCm 7.2,
# kp_head / k_mha_qkv).
UNITS: Dict[str, List[str]] = {
    "pf_lib.hip": [f"--offload-arch={ARCH}", "-fno-slp-vectorize"] + SCHED,
    "pf_precise.hip": [f"--offload-arch={ARCH}"],
    "pf_mha.hip": [f"--offload-arch={ARCH}", "-fno-slp-vectorize"],
    "pf_hostio.cpp": [],
}
HEADERS = ["pf_device.hip.h", "pf_mha.hip.h", "pf_precise.hip.h", "pf_precise_host.hip.h", "pf_layout.h", "pf_host_prep.h"]
# what decides the bits and the speed of the dominant kernels (k_main, k_colstats): the PMC traffic file under
# profiles/ is tied to this hash (bench.py: a mismatch means the counters are stale -> traffic null)
KERNEL_FILES = ["pf_device.hip.h", "pf_layout.h"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _deps() -> List[str]:
    return ([os.path.join(CSRC, u) for u in UNITS] + [os.path.join(CSRC, h) for h in HEADERS] +
            [os.path.join(INCLUDE, "phyloformer_amd.h"), os.path.abspath(__file__)])


def is_stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps())


def _sha(paths: List[str], extra: str = "") -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as fh:
            h.update(fh.read())
    h.update(extra.encode())
    return h.hexdigest()[:16]


def source_hash() -> str:
    """sha256 (16 hex digits) over every source and header the library is built from."""
    return _sha([d for d in _deps() if not d.endswith("build.py")])


def kernel_hash(lib_flags: Optional[List[str]] = None) -> str:
    """Hash of what decides k_main / k_colstats: their source and the flags pf_lib.hip is compiled with."""
    flags = COMMON + (UNITS["pf_lib.hip"] if lib_flags is None else lib_flags)
    return _sha([os.path.join(CSRC, f) for f in KERNEL_FILES], " ".join(flags))


def _hipcc_version(hipcc: str) -> str:
    try:
        out = subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError):
        return "unknown"
    hip = next((l.split(":", 1)[1].strip() for l in out.splitlines() if l.startswith("HIP version")), "?")
    clang = next((l.strip() for l in out.splitlines() if "clang version" in l), "?")
    return f"HIP {hip}; {clang}"[:200]


def build(force: bool = False, verbose: bool = False, out: str = LIB, extra: Optional[List[str]] = None) -> str:
    """Compile and link.  ``PF_ALLOW_SCHED_FALLBACK=1`` lets a hipcc that cannot compile pf_lib.hip with the
    iterative-ILP strategy fall back to the default one (1-7 % slower kernels) - the library
    then reports ``sched_fallback: true`` and bench.py carries it; without the variable the build FAILS.
    ``PF_BUILD_FORCE_FALLBACK=1`` (tests) takes the fallback without trying."""
    if not force and not is_stale(out):
        return out
    extra = list(extra or [])            # A/B builds (tools/build_variant.py): -D switches for the HIP units
    hipcc = hipcc_path()
    allow = os.environ.get("PF_ALLOW_SCHED_FALLBACK") == "1"
    forced = os.environ.get("PF_BUILD_FORCE_FALLBACK") == "1"
    tmp = tempfile.mkdtemp(prefix="pf_build_")
    try:
        objs, used = [], {}
        for unit, unit_flags in UNITS.items():
            uf = unit_flags + (extra if unit.endswith(".hip") else [])
            attempts = [uf]
            if unit == "pf_lib.hip":
                no_sched = [f for f in uf if f not in SCHED]
                attempts = [no_sched] if forced else ([uf, no_sched] if allow else [uf])
            obj = os.path.join(tmp, unit + ".o")
            res = None
            for flags in attempts:
                cmd = [hipcc, *COMMON, *flags, "-c", os.path.join(CSRC, unit), "-o", obj]
                if verbose:
                    cmd.append("-Rpass-analysis=kernel-resource-usage")
                    print(" ".join(cmd), file=sys.stderr)
                res = subprocess.run(cmd, capture_output=True, text=True)
                if res.returncode == 0:
                    used[unit] = flags
                    break
                sys.stderr.write(f"phyloformer_amd.build: hipcc failed on {unit} with {' '.join(flags)}\n")
            if verbose or res.returncode != 0:
                sys.stderr.write(res.stdout + res.stderr)
            if res.returncode != 0:
                hint = ("" if unit != "pf_lib.hip" or allow else
                        " (set PF_ALLOW_SCHED_FALLBACK=1 to accept the default scheduling strategy: slower kernels)")
                raise RuntimeError(f"hipcc failed on {unit} with exit code {res.returncode}{hint}")
            objs.append(obj)
        fallback = not all(f in used["pf_lib.hip"] for f in SCHED)
        if fallback:
            sys.stderr.write("phyloformer_amd.build: WARNING - pf_lib.hip was compiled with hipcc's DEFAULT scheduling "
                             "strategy (sched_fallback: true in pf_build_info)\n")
        info = {
            "abi": 5, "arch": ARCH, "hipcc": _hipcc_version(hipcc),
            "sched_strategy": "default" if fallback else "iterative-ilp", "sched_fallback": fallback,
            "flags": {u: " ".join(COMMON + f) for u, f in used.items()},
            "source_hash": source_hash(), "kernel_hash": kernel_hash(used["pf_lib.hip"]),
        }
        gen = os.path.join(tmp, "pf_build_info.cpp")
        with open(gen, "w") as fh:
            text = json.dumps(info, sort_keys=True)
            fh.write('extern "C" const char* pf_build_info(void) { return R"PFBI(' + text + ')PFBI"; }\n')
        part = f"{out}.{os.getpid()}.tmp"          # (two builders at once - two pytest processes - must not share it)
        cmd = [hipcc, *COMMON, "-shared", *objs, gen, "-o", part, "-ldl"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stdout + res.stderr)
            raise RuntimeError(f"link failed with exit code {res.returncode}")
        os.replace(part, out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
