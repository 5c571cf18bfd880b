#!/usr/bin/env python3
"""Round 6: the adversarial study (guard_study.py: 5-60 sequences x 20-200 sites) at the sizes the model is used at -
80-150 sequences x 250-500 sites, 8 kinds of input, all five checkpoints: 32 cases.

    python tests/dev/large_shape_study.py oracles ORC.npz     # anywhere (CPU, ~1 h on 8 cores): fp32 / fp64 oracle outputs
    python tests/dev/large_shape_study.py gen GPU.npz         # GPU box: as the product routes (PF_STUDY_FORCED=1: precise = 0)
    python tests/dev/large_shape_study.py score ORC.npz GPU.npz [GPU2.npz ...]

The oracle is pf_oracle_torch (the reference's op order; pinned against the reference's goldens, and in float64 equal to
the numpy oracle's float64 to 4e-16) - the numpy oracle needs 7 minutes per case here."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests", "dev"))
import guard_study as G

SHAPES = [(80, 300), (100, 500), (120, 250), (150, 400)]
KINDS = ["sim", "gapped", "noise50", "uniform22", "uniform20", "alphabet2", "gap_columns", "all_x"]


def cases():
    rng = np.random.default_rng(6160)
    c = 0
    for (n, l) in SHAPES:
        for kind in KINDS:
            yield c, G.CK[c % len(G.CK)], kind, n, l, G.make_case(kind, n, l, np.random.default_rng(int(rng.integers(1 << 31))))
            c += 1


def oracles(out):
    from oracle import pf_oracle_torch as T
    from phyloformer_amd.weights import load_weights
    w = {ck: load_weights(os.path.join(REPO, "models", ck + ".ckpt")).tensors for ck in G.CK}
    res = {}
    for c, ck, kind, n, l, idx in cases():
        t0 = time.time()
        res[f"f32_{c}"] = T.forward(w[ck], idx)
        res[f"f64_{c}"] = T.forward(w[ck], idx, dtype=np.float64)
        print(f"{c} {ck} {kind} {n}x{l}: largest distance {float(res[f'f32_{c}'].max()):.2f}, fp32's own error "
              f"{float(np.abs(res[f'f32_{c}'] - res[f'f64_{c}']).max()):.2e} ({time.time() - t0:.0f} s)", flush=True)
        np.savez_compressed(out, **res)


def gen(out):
    from phyloformer_amd.engine import Engine
    from phyloformer_amd.weights import load_weights
    eng = {n: Engine(load_weights(os.path.join(REPO, "models", n + ".ckpt")), 0) for n in G.CK}
    forced = os.environ.get("PF_STUDY_FORCED") == "1"
    for e in eng.values():
        e.set_option("precise", 0 if forced else -1)
    res, nre = {}, 0
    for c, ck, kind, n, l, idx in cases():
        eng[ck].profile_reset()
        res[f"gpu{c}"] = eng[ck].forward(idx)
        nre += eng[ck].rechecked_count()
    np.savez_compressed(out, **res)
    print(f"{c + 1} cases, {nre} recomputed by the range re-check -> {out}")


def score(orc_path, gpu_paths):
    orc = np.load(orc_path)
    for gp in gpu_paths:
        z = np.load(gp)
        print("==", gp)
        over = 0
        for c, ck, kind, n, l, _ in cases():
            f32, f64, gpu = orc[f"f32_{c}"], orc[f"f64_{c}"], z[f"gpu{c}"]
            e, own = float(np.abs(gpu - f32).max()), float(np.abs(f32 - f64).max())
            bad = e > max(1e-4, 2 * own)
            over += bad
            print(f"  {n:3d} x {l:3d} {kind:12s} {ck:10s} largest distance {float(f32.max()):6.2f}  |GPU-f32| {e:.2e}  fp32's own "
                  f"{own:.2e}  |GPU-f64| {float(np.abs(gpu - f64).max()):.2e}{'  OVER' if bad else ''}")
        print(f"  over max(1e-4, 2 x fp32's own): {over} of {c + 1}")


if __name__ == "__main__":
    {"oracles": lambda: oracles(sys.argv[2]), "gen": lambda: gen(sys.argv[2]), "score": lambda: score(sys.argv[2], sys.argv[3:])}[sys.argv[1]]()
