#!/usr/bin/env python3
"""Second half of guard_study.py done properly: per case, the fp32 oracle WITH taps (token RMS / max |x| after every
sub-block), the fp64 oracle, and the GPU's error, saved to an .npz for indicator fitting.
    OMP_NUM_THREADS=1 python tests/dev/guard_study2.py IN.npz OUT.npz"""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1"); os.environ.setdefault("OPENBLAS_NUM_THREADS", "1"); os.environ.setdefault("MKL_NUM_THREADS", "1")
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests", "dev"))
import guard_study as G


def work(args):
    c, ck, kind, n, l, idx = args
    from oracle import pf_oracle as O
    from phyloformer_amd.weights import load_weights
    w = load_weights(os.path.join(REPO, "models", ck + ".ckpt")).tensors
    stats = {}

    def tap(name, x):
        if x.ndim == 3:
            rms = np.sqrt((x.astype(np.float64) ** 2).mean(-1))
            stats[name] = (float(rms.max()), float(np.abs(x).max()), float(rms.mean()))
        elif name == "logits":
            stats[name] = (float(np.abs(x).max()), float(x.max()), float(x.mean()))
    f32 = O.forward(w, idx, tap=tap)
    f64 = O.forward(w, idx, dtype=np.float64)
    names = ["embed"] + [f"block{b}.{s}" for b in range(6) for s in ("row", "col", "ffn")] + ["logits"]
    return c, f32, f64, np.array([stats[k] for k in names])


def main(inp, out):
    import multiprocessing as mp
    z = np.load(inp)
    todo = list(G.cases())
    res = {}
    with mp.Pool(8) as pool:
        for c, f32, f64, st in pool.imap_unordered(work, todo, chunksize=2):
            gpu = z[f"gpu{c}"]
            res[f"m{c}"] = np.array([np.abs(gpu).max(), np.abs(gpu - f32).max(), np.abs(f32 - f64).max(), np.abs(gpu - f64).max()])
            res[f"s{c}"] = st
    meta = np.array([(c, ck, kind, n, l) for c, ck, kind, n, l, _ in todo], dtype=object)
    np.savez_compressed(out, meta=meta, **res)


main(sys.argv[1], sys.argv[2])
