#!/usr/bin/env python3
"""Diagnostic run on the GPU box: prints per-stage errors and timings instead of asserting."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from oracle import pf_oracle as O  # noqa: E402
from phyloformer_amd.engine import Engine  # noqa: E402
from phyloformer_amd.weights import load_weights  # noqa: E402
import devmath  # noqa: E402


def main():
    w = load_weights(os.path.join(REPO, "models/pf.ckpt"))
    e = Engine(w, 0)
    print(e.device_info())
    out = e.selftest()
    lane = np.arange(64)
    print("pair_sum ok", np.array_equal(out[0:64], (lane % 32) * 2 + 32.0), out[0:4], out[32:36])
    print("pair_other ok", np.array_equal(out[64:128], (lane ^ 32).astype(np.float32)), out[64:68], out[96:100])
    print("row16 ok", np.array_equal(out[128:192], np.repeat(lane.reshape(4, 16).sum(1), 16)), out[128:192:16])
    print("half32 ok", np.array_equal(out[192:256], np.repeat(lane.reshape(2, 32).sum(1), 32)), out[192:256:16])
    d1 = out[256:1280].reshape(64, 16)
    d2 = out[1280:2304].reshape(64, 16)
    ok1 = ok2 = True
    for l in range(64):
        for r in range(16):
            m, n = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31
            ok1 &= d1[l, r] == m + 32 * ((n % 16) % 8)
            ok2 &= d2[l, r] == n % 16
    print("mfma layout ok", ok1, ok2)
    if not (ok1 and ok2):
        print("d1 lane0", d1[0], "lane1", d1[1], "lane32", d1[32])
        print("d2 lane0", d2[0], "lane1", d2[1], "lane17", d2[17], "lane32", d2[32])

    g = np.load(os.path.join(REPO, "tests/golden/taps_tiny.npz"))
    wt = w.tensors
    e.set_option("debug_keep", 1)
    d = e.forward(g["idx"])
    P, L = 10, 16
    x0 = e.debug_read("x0").reshape(P, L, 64)
    print("x0 exact", np.array_equal(x0, g["embed"]), np.abs(x0 - g["embed"]).max())
    x_in = g["embed"]
    for k in range(6):
        srow = e.debug_read(f"srow{k}").reshape(P, 72)
        want, _ = devmath.expected_srow(wt, k, x_in)
        mrow = e.debug_read(f"mrow{k}").reshape(P, 4, 64)   # 4 heads; the bias row is not stored
        wantm = devmath.expected_mrow(wt, k, want, L)
        ctx = e.debug_read(f"ctx{k}").reshape(L, 64)
        wantc, _ = devmath.expected_ctx(wt, k, g[f"block{k}.row"])
        xk = e.debug_read(f"x{k + 1}").reshape(P, L, 64)
        ref = g[f"block{k}.ffn"]
        print(f"block {k}: srow rel {np.abs(srow - want).max() / np.abs(want).max():.2e} "
              f"(skv {np.abs(srow[:, :64] - want[:, :64]).max():.2e} sq {np.abs(srow[:, 64:68] - want[:, 64:68]).max():.2e} "
              f"sk {np.abs(srow[:, 68:] - want[:, 68:]).max():.2e}) "
              f"mrow rel {np.abs(mrow - wantm).max() / np.abs(wantm).max():.2e} "
              f"ctx rel {np.abs(ctx - wantc).max() / np.abs(wantc).max():.2e} "
              f"x rel {np.abs(xk - ref).max() / np.abs(ref).max():.2e} (max {np.abs(ref).max():.1f})")
        x_in = ref
    print("tiny dist err", np.abs(d - g["dist"]).max(), "max", np.abs(g["dist"]).max())
    e.set_option("debug_keep", 0)

    cfg = np.load(os.path.join(REPO, "tests/golden/configs.npz"))
    for tag in ("c2", "c3"):
        idx = cfg[tag + "_idx"]
        t0 = time.time()
        got = e.forward(idx)
        t1 = time.time()
        got = e.forward(idx)
        t2 = time.time()
        print(tag, idx.shape, "err", np.abs(got - cfg[tag + "_dist"]).max(), "first", round(t1 - t0, 4), "second", round(t2 - t1, 4))
    # per-kernel timing at the headline shape
    e.set_option("profile", 1)
    idx = np.repeat(cfg["c3_idx"], 4, axis=0)
    e.forward(idx)
    e.profile_reset()
    t0 = time.time()
    e.forward(idx)
    dt = time.time() - t0
    print(f"60x500 x4: {dt * 1e3:.2f} ms wall -> {4 / dt:.1f} aln/s")
    for k in ("embed", "rowfin", "colstats", "colfin", "main", "allreduce"):
        n, ms = e.profile_get(k)
        print(f"  {k:9s} launches {n:3d} total {ms:8.3f} ms")
    e.close()


if __name__ == "__main__":
    main()
