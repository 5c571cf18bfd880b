#!/usr/bin/env python3
"""GPU box: where does the DEFAULT (split-bf16) path lose the 1e-4 bound?  Sweeps small shapes with the float64 path
switched off (precise = 0) and prints, per (N, L), the worst |GPU - fp32 oracle|, |fp32 oracle - fp64 oracle| and
|GPU - fp64 oracle| over checkpoints / seeds / input kinds.  The thresholds of pf_precise_host.hip.h::use_precise come
from this table (profiles/r05_precise_sweep.txt).
    python tests/dev/precise_sweep.py [out.json]"""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import pf_oracle as O
from phyloformer_amd.engine import Engine
from phyloformer_amd.msa_sim import simulate_batch
from phyloformer_amd.weights import load_weights

NS = [2, 3, 4, 5, 6, 8, 12, 20, 40]
LS = [1, 2, 3, 5, 7, 10, 15, 16, 20, 24, 32, 33, 48, 64, 100, 200]
CK = ["pf", "pf_base", "pf_indel", "pf_cherry", "pf_selreg"]


def main():
    ws = {n: load_weights(os.path.join(REPO, "models", n + ".ckpt")) for n in CK}
    eng = {n: Engine(ws[n], 0) for n in CK}
    for e in eng.values():
        e.set_option("precise", 0)
    rows = []
    t0 = time.time()
    for n in NS:
        for l in LS:
            if n * (n - 1) // 2 * l > 80_000:
                continue
            worst = {"gpu_f32": 0.0, "f32_f64": 0.0, "gpu_f64": 0.0, "scale": 0.0, "over": 0, "cases": 0}
            for ci, ck in enumerate(CK):
                for kind in range(3):       # simulated, gapped, uniformly random residues
                    seed = 7919 * n + 104729 * l + 31 * ci + kind
                    if kind == 2:
                        idx = np.random.default_rng(seed).integers(0, 22, (2, n, l)).astype(np.uint8)
                    else:
                        idx = simulate_batch(2, n, l, seed=seed, gaps=(kind == 1))
                    got = eng[ck].forward(idx)
                    f32 = O.forward_batch(ws[ck].tensors, idx)
                    f64 = O.forward_batch(ws[ck].tensors, idx, dtype=np.float64)
                    a, b, c = (float(np.abs(got - f32).max()), float(np.abs(f32 - f64).max()),
                               float(np.abs(got - f64).max()))
                    worst["gpu_f32"] = max(worst["gpu_f32"], a)
                    worst["f32_f64"] = max(worst["f32_f64"], b)
                    worst["gpu_f64"] = max(worst["gpu_f64"], c)
                    worst["scale"] = max(worst["scale"], float(np.abs(f32).max()))
                    worst["over"] += int(a > max(1e-4, 2 * b))
                    worst["cases"] += 1
            rows.append({"N": n, "L": l, "tokens": n * (n - 1) // 2 * l, **worst})
            r = rows[-1]
            print(f"N={n:3d} L={l:4d} tokens={r['tokens']:6d}: GPU-f32 {r['gpu_f32']:.2e}  f32-f64 {r['f32_f64']:.2e}  "
                  f"GPU-f64 {r['gpu_f64']:.2e}  max|d| {r['scale']:.3g}  over {r['over']}/{r['cases']}", flush=True)
    print(f"{time.time() - t0:.0f} s")
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            json.dump(rows, fh, indent=1)
    for e in eng.values():
        e.close()


main()
