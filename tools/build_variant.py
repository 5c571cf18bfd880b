#!/usr/bin/env python3
"""Build a variant of the native library next to the product one, for A/B runs inside ONE GPU call
(tools/flag_compare.py, tools/kernel_ab.py take the file names):

    python tools/build_variant.py lib_x.so [--src DIR] [hipcc flags, e.g. -DPF_RUN_MAX=4]

--src DIR: repository root to take phyloformer_amd/csrc and include/ from (e.g. a `git worktree` of another
commit); default this checkout.  The product library (libphyloformer_amd.so) is never touched."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phyloformer_amd import build as B

def main():
    args = sys.argv[1:]
    out = os.path.join(REPO, "phyloformer_amd", args.pop(0))
    src = REPO
    if args and args[0] == "--src":
        src = os.path.abspath(args[1]); args = args[2:]
    sources = [os.path.join(src, "phyloformer_amd", "csrc", f) for f in ("pf_lib.hip", "pf_hostio.cpp")]
    sched = [] if "--default-sched" in args else ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
    args = [a for a in args if a != "--default-sched"]
    if "--sched" in args:                       # another scheduling strategy: --sched max-ilp
        k = args.index("--sched")
        sched = ["-mllvm", f"-amdgpu-sched-strategy={args[k + 1]}"]
        del args[k:k + 2]
    cmd = [B.hipcc_path(), f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC", "-fno-slp-vectorize",
           "-Wno-unused-value", *sched, *args, *sources, "-o", out, "-ldl"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode:
        sys.stderr.write(res.stderr[-3000:]); sys.exit(1)
    print(out)
main()
