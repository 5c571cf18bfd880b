#!/usr/bin/env python3
"""Randomised soak on the GPU box (not part of the suite): random shapes / batches / checkpoints / gap patterns through
pf_forward, against the numpy oracle (fp32) and against the engine itself in other schedules (one stream, alone,
emulated shards).  Prints the worst cases; exits non-zero on a violation.
    python tests/dev/soak.py [cases] [seed]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import pf_oracle as O
from phyloformer_amd.engine import Engine
from phyloformer_amd.msa_sim import simulate_batch
from phyloformer_amd.weights import load_weights


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
    names = ["pf", "pf_base", "pf_indel", "pf_cherry", "pf_selreg"]
    ws = {n: load_weights(os.path.join(REPO, "models", n + ".ckpt")) for n in names}
    engines = {n: Engine(ws[n], 0) for n in names}
    worst, bad = [], 0
    t0 = time.time()
    for c in range(cases):
        name = names[int(rng.integers(len(names)))]
        N = int(rng.choice([2, 3, 4, 5, 7, 9, 12, 17, 24, 33, 40]))
        L = int(rng.choice([1, 2, 7, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 200, 257]))
        B = int(rng.integers(1, 6))
        if N * (N - 1) // 2 * L * B > 400_000:
            B = 1
        idx = simulate_batch(B, N, L, seed=int(rng.integers(1 << 30)), gaps=bool(rng.integers(2)))
        e = engines[name]
        got = e.forward(idx)
        want = O.forward_batch(ws[name].tensors, idx)
        err = float(np.abs(got - want).max())
        scale = max(1.0, float(np.abs(want).max()))
        ok = np.isfinite(got).all() and err <= 1e-4 * max(1.0, scale / 5)       # in-distribution bound 1e-4; OOD scales (tests/test_gpu_parity.py)
        # schedule invariance: alone / one stream / emulated shards give the batch's bits (shards: to 2e-5 relative)
        alone = np.stack([e.forward(a) for a in idx])
        e.set_option("two_streams", 0)
        one = e.forward(idx)
        e.set_option("two_streams", 1)
        inv = np.array_equal(alone, got) and np.array_equal(one, got)
        sh = e.forward_shards_emulated(idx, int(rng.integers(2, 6))) if L >= 2 else got
        shard_ok = float(np.abs(sh - got).max()) <= 3e-5 * scale
        worst.append((err / scale, err, scale, name, B, N, L))
        if not (ok and inv and shard_ok):
            bad += 1
            print(f"VIOLATION case {c}: {name} B={B} N={N} L={L}: err {err:.3e} (max ref {scale:.3g}), invariance {inv}, shards {shard_ok}", flush=True)
    worst.sort(reverse=True)
    print(f"{cases} cases in {time.time() - t0:.0f} s, {bad} violations; worst relative errors:")
    for rel, err, scale, name, B, N, L in worst[:8]:
        print(f"  {name:10s} B={B} N={N:3d} L={L:4d}: abs {err:.3e}  max |ref| {scale:.3g}  rel {rel:.2e}")
    for e in engines.values():
        e.close()
    sys.exit(1 if bad else 0)


main()
