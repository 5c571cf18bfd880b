"""Build ``libphyloformer_amd.so`` (HIP, gfx950) in-tree.

    python -m phyloformer_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU, so this also runs in the build container.
The shared object lands next to this file and is git-ignored; it travels to
the GPU box with the working-tree snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libphyloformer_amd.so")
SOURCES = [os.path.join(CSRC, "pf_lib.hip"), os.path.join(CSRC, "pf_hostio.cpp")]
DEPS = SOURCES + [os.path.join(CSRC, "pf_device.hip.h"), os.path.join(CSRC, "pf_mha.hip.h"),
                  os.path.join(os.path.dirname(HERE), "include", "phyloformer_amd.h")]
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    # -amdgpu-sched-strategy=iterative-ilp: with the default (max-occupancy) strategy hipcc sinks the software-
    # pipelined LDS fragment reads of k_main's hidden loop down to their first use (read - wait - MFMA, one at a
    # time, a single fragment buffer) whenever code outside the loop changes - +7 % on the loop between two builds
    # with identical loop source (round 3); the iterative ILP strategy keeps the read pairs
    # ahead of the MFMAs and is 1.3 % faster on the forward (k_colstats gains too).
    sched = ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
    base = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC",
            "-fno-slp-vectorize", "-Wno-unused-value"]
    tail = [*SOURCES, "-o", LIB + ".tmp", "-ldl"]
    if verbose:
        tail.append("-Rpass-analysis=kernel-resource-usage")
    res = None
    # the iterative-ILP strategy is a less travelled path of the compiler (it crashed on one variant of k_rowfin
    # during round 3): if hipcc fails with it, build with the default strategy rather than not at all
    for flags in (sched, []):
        cmd = base + flags + tail
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode == 0:
            break
        if flags:
            sys.stderr.write("phyloformer_amd.build: hipcc failed with " + " ".join(flags) +
                             "; retrying with the default scheduling strategy\n")
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed with exit code {res.returncode}")
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
