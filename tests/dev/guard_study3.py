#!/usr/bin/env python3
"""fp32 / fp64 oracle outputs of guard_study.py's 741 cases, SAVED (the first pass kept only the error norms), so that any
later GPU output file - default kernels forced, or as the product routes - can be scored without the oracles' hour.
    OMP_NUM_THREADS=1 python tests/dev/guard_study3.py oracles OUT.npz
    python tests/dev/guard_study3.py score ORACLES.npz GPU.npz [GPU2.npz ...]"""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1"); os.environ.setdefault("OPENBLAS_NUM_THREADS", "1"); os.environ.setdefault("MKL_NUM_THREADS", "1")
import collections
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests", "dev"))
import guard_study as G


def oracles(out):
    import multiprocessing as mp
    todo = list(G.cases())
    res = {}
    with mp.Pool(8) as pool:
        for c, f32, f64 in pool.imap_unordered(G._oracle, todo, chunksize=2):
            res[f"f32_{c}"], res[f"f64_{c}"] = f32, f64
    np.savez_compressed(out, **res)


def score(orc_path, gpu_paths):
    orc = np.load(orc_path)
    todo = [(c, ck, kind, n, l) for c, ck, kind, n, l, _ in G.cases()]
    for gp in gpu_paths:
        z = np.load(gp)
        by = collections.defaultdict(list)
        for c, ck, kind, n, l in todo:
            gpu, f32, f64 = z[f"gpu{c}"], orc[f"f32_{c}"], orc[f"f64_{c}"]
            e, own = float(np.abs(gpu - f32).max()), float(np.abs(f32 - f64).max())
            by[kind].append((e, own, e > max(1e-4, 2 * own), n, l))
        print(f"== {gp}")
        tot = 0
        for kind in sorted(by):
            v = by[kind]
            w = max(v)
            over = sum(x[2] for x in v)
            tot += over
            print(f"  {kind:13s} n={len(v):3d} worst |GPU-f32| {w[0]:.2e} ({w[3]}x{w[4]}, fp32's own {w[1]:.1e})  over the bound: {over}")
        print(f"  total over max(1e-4, 2 x fp32's own): {tot} of {len(todo)}")


if __name__ == "__main__":
    if sys.argv[1] == "oracles":
        oracles(sys.argv[2])
    else:
        score(sys.argv[2], sys.argv[3:])
