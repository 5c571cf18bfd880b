#!/usr/bin/env python3
"""Which alignments are ill-conditioned for the default (split-bf16) kernels?  Two halves:

    python tests/dev/guard_study.py gen OUT.npz        # GPU box: default kernels (precise = 0, guard off) on ~700
                                                       # seeded inputs of realistic and adversarial kinds
    python tests/dev/guard_study.py analyze OUT.npz    # anywhere: fp32 / fp64 oracles, error against the largest
                                                       # distance the GPU itself predicted

The GPU's error tracks 20 x the fp32 reference's own distance from float64, and both grow with the magnitude of the
residual stream, which the largest predicted distance of the alignment follows.  The study picks the threshold of the
result guard (pf_lib.hip: alignments whose largest distance exceeds it are recomputed in float64).
"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
CK = ["pf", "pf_base", "pf_indel", "pf_cherry", "pf_selreg"]
SHAPES = [(5, 64), (6, 200), (8, 48), (8, 200), (12, 24), (12, 100), (20, 20), (20, 100), (20, 200), (33, 40),
          (40, 33), (40, 100), (60, 64)]


def make_case(kind, n, l, rng):
    from phyloformer_amd.msa_sim import simulate_alignment
    sim = simulate_alignment(n, l, rng=rng, gaps=False)
    if kind == "sim":
        return sim
    if kind == "gapped":
        return simulate_alignment(n, l, rng=rng, gaps=True)
    if kind.startswith("noise"):                 # a fraction of the entries replaced by uniformly random residues
        f = float(kind[5:]) / 100.0
        m = rng.random((n, l)) < f
        return np.where(m, rng.integers(0, 22, (n, l)), sim).astype(np.uint8)
    if kind == "uniform22":
        return rng.integers(0, 22, (n, l)).astype(np.uint8)
    if kind == "uniform20":
        return rng.integers(0, 20, (n, l)).astype(np.uint8)
    if kind == "alphabet2":
        return rng.choice(rng.choice(20, 2, replace=False), (n, l)).astype(np.uint8)
    if kind == "alphabet4":
        return rng.choice(rng.choice(20, 4, replace=False), (n, l)).astype(np.uint8)
    if kind == "identical":
        return np.repeat(sim[:1], n, axis=0)
    if kind == "two_distinct":                   # two unrelated sequences, each repeated
        two = rng.integers(0, 20, (2, l)).astype(np.uint8)
        return two[rng.integers(0, 2, n)]
    if kind == "two_related":
        return sim[:2][rng.integers(0, 2, n)]
    if kind == "gap_columns":                    # half of the columns are gaps in every sequence
        out = sim.copy()
        out[:, rng.random(l) < 0.5] = 21
        return out
    if kind == "gap_sequence":                   # one sequence is all gaps
        out = sim.copy()
        out[int(rng.integers(n))] = 21
        return out
    if kind == "all_x":
        out = sim.copy()
        out[rng.random((n, l)) < 0.5] = 20
        return out
    raise ValueError(kind)


KINDS = ["sim", "sim", "gapped", "noise5", "noise10", "noise20", "noise35", "noise50", "noise75", "uniform22", "uniform20",
         "alphabet2", "alphabet4", "identical", "two_distinct", "two_related", "gap_columns", "gap_sequence", "all_x"]


def cases():
    rng = np.random.default_rng(5150)
    c = 0
    for (n, l) in SHAPES:
        for ki, kind in enumerate(KINDS):
            for rep in range(3):
                ck = CK[(c + rep) % len(CK)]
                yield c, ck, kind, n, l, make_case(kind, n, l, np.random.default_rng(int(rng.integers(1 << 31))))
                c += 1


def gen(out):
    from phyloformer_amd.engine import Engine
    from phyloformer_amd.weights import load_weights
    eng = {n: Engine(load_weights(os.path.join(REPO, "models", n + ".ckpt")), 0) for n in CK}
    routed = os.environ.get("PF_STUDY_ROUTED") == "1"     # 1: as the product routes (float64 path for small shapes)
    for e in eng.values():
        e.set_option("precise", -1 if routed else 0)
    res = {}
    t0 = time.time()
    for c, ck, kind, n, l, idx in cases():
        res[f"idx{c}"] = idx
        res[f"gpu{c}"] = eng[ck].forward(idx)
    np.savez_compressed(out, **res)
    print(f"{c + 1} cases in {time.time() - t0:.1f} s -> {out}")


def _oracle(args):
    c, ck, kind, n, l, idx = args
    from oracle import pf_oracle as O
    from phyloformer_amd.weights import load_weights
    w = load_weights(os.path.join(REPO, "models", ck + ".ckpt")).tensors
    return c, O.forward(w, idx), O.forward(w, idx, dtype=np.float64)


def analyze(path):
    import multiprocessing as mp
    z = np.load(path)
    todo = list(cases())
    for c, ck, kind, n, l, idx in todo:
        assert np.array_equal(z[f"idx{c}"], idx), c
    with mp.Pool(8) as pool:
        orc = {c: (a, b) for c, a, b in pool.imap_unordered(_oracle, todo, chunksize=4)}
    rows = []
    for c, ck, kind, n, l, idx in todo:
        gpu = z[f"gpu{c}"]
        f32, f64 = orc[c]
        rows.append((float(np.abs(gpu).max()), float(np.abs(gpu - f32).max()), float(np.abs(f32 - f64).max()),
                     float(np.abs(gpu - f64).max()), kind, ck, n, l))
    rows.sort()
    print("max|d| (GPU)  |GPU-f32|   |f32-f64|   |GPU-f64|   viol  kind          ckpt       N    L")
    for d, a, b, g, kind, ck, n, l in rows:
        print(f"{d:10.3f}  {a:.3e}  {b:.3e}  {g:.3e}  {'OVER' if a > max(1e-4, 2 * b) else '    '}  {kind:13s} {ck:10s} {n:3d} {l:4d}")
    print("\nby largest predicted distance:")
    edges = [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 24, 1e9]
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = [r for r in rows if lo <= r[0] < hi]
        if sel:
            print(f"  {lo:5.0f} <= max|d| < {hi:5.0f}: {len(sel):4d} cases, worst |GPU-f32| {max(r[1] for r in sel):.3e}, "
                  f"worst |f32-f64| {max(r[2] for r in sel):.3e}, over the bound {sum(r[1] > max(1e-4, 2 * r[2]) for r in sel)}")


if __name__ == "__main__":
    (gen if sys.argv[1] == "gen" else analyze)(sys.argv[2])
