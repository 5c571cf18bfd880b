#!/usr/bin/env python3
"""Instruction histogram of one kernel (or of its innermost / a chosen loop) from hipcc's gfx950 assembly:

    python tools/isa_histogram.py 'k_main<1, true>' [--loop N] [--scratch]

Compiles csrc/pf_lib.hip to assembly with the product flags (no GPU needed), cuts out the kernel whose demangled name
contains the argument and prints mnemonic counts.  --loop N: only the N-th "Loop Header" block of the listing (0 = the
first; the FFN hidden loop of k_main<MID> is the loop with 24 v_mfma); --scratch: list scratch_ (spill) instructions
with their line numbers.  Backs DESIGN.md section 9's per-hidden-value instruction count and the round-4 spill analysis."""
import collections, os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phyloformer_amd import build as B


def main():
    want = sys.argv[1]
    loop = int(sys.argv[sys.argv.index("--loop") + 1]) if "--loop" in sys.argv else None
    asm = os.path.join(tempfile.gettempdir(), "pf_lib_isa.s")
    cmd = [B.hipcc_path(), *B.COMMON, *B.UNITS["pf_lib.hip"], "-S", "--cuda-device-only", "-o", asm, os.path.join(B.CSRC, "pf_lib.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode:
        sys.stderr.write(res.stderr[-3000:]); sys.exit(1)
    lines = open(asm).read().splitlines()
    body, name = None, None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m and body is None:
            dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            if want in dem:
                name, body = dem, []
        if body is not None:
            body.append(ln)
            if ln.strip() == "s_endpgm":
                break
    if body is None:
        sys.exit(f"no kernel matching {want!r}")
    if "--scratch" in sys.argv:
        for i, ln in enumerate(body):
            if "scratch_" in ln:
                print(i, ln.strip())
    loops = []
    for i, ln in enumerate(body):
        if "Loop Header" in ln:
            j = i
            while j >= 0 and not re.match(r"^\.LBB\d+_\d+:", body[j]):
                j -= 1
            label = body[j].split(":")[0].strip()
            ends = [k for k in range(i + 1, len(body)) if re.search(r"s_cbranch\w*\s+" + re.escape(label) + r"\b", body[k])]
            if ends:
                loops.append((j, ends[-1], label))
    for n, (a0, a1, label) in enumerate(loops):
        print(f"  loop {n}: {label}, lines {a0}-{a1}, {sum('v_mfma' in x for x in body[a0:a1 + 1])} MFMA")
    if loop is not None:
        body = body[loops[loop][0]:loops[loop][1] + 1]
    hist = collections.Counter()
    for ln in body:
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\b", ln)
        if m and not ln.strip().startswith(";"):
            hist[m.group(1)] += 1
    print(f"{name}" + (f", loop {loop}" if loop is not None else "") + f": {sum(hist.values())} instructions")
    for k, v in hist.most_common():
        print(f"  {v:5d}  {k}")


main()
