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
    if args and args[0] == "--src":
        src = os.path.abspath(args[1]); args = args[2:]
        B.CSRC = os.path.join(src, "phyloformer_amd", "csrc")          # another checkout's sources, this checkout's recipe
        B.INCLUDE = os.path.join(src, "include")
        B.UNITS = {u: f for u, f in B.UNITS.items() if os.path.exists(os.path.join(B.CSRC, u))}
        B.HEADERS = [h for h in B.HEADERS if os.path.exists(os.path.join(B.CSRC, h))]
    if "--default-sched" in args:
        args = [a for a in args if a != "--default-sched"]
        B.UNITS["pf_lib.hip"] = [f for f in B.UNITS["pf_lib.hip"] if f not in B.SCHED]
    if "--sched" in args:                       # another scheduling strategy: --sched max-ilp
        k = args.index("--sched")
        B.UNITS["pf_lib.hip"] = [f for f in B.UNITS["pf_lib.hip"] if f not in B.SCHED] + ["-mllvm", f"-amdgpu-sched-strategy={args[k + 1]}"]
        B.SCHED = B.UNITS["pf_lib.hip"][-2:]
        del args[k:k + 2]
    print(B.build(force=True, out=out, extra=args))
main()
