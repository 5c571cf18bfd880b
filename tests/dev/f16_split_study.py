#!/usr/bin/env python3
"""CPU study (oracle only, float64 carrier; not on the product path) behind round 6's switch of the split MFMA operands
from bf16 to fp16: the DEVICE's own contraction set - not every matmul of the reference - with
each operand split into two 16-bit limbs, products hi*hi + hi*lo + lo*hi, everything else in float64, so that the
number printed is the operand format's share of the error and nothing else.

The five split contractions of a block (pf_device.hip.h):
    row statistics   [Wv'; Wq'; Wk'] x~          (block 0: exact, from the residue-pair table)
    row mix          M_base^T (+ bias rows)  x  (q' L / S_q | 1 | 1)
    column apply     (Wo 2^4)  x  (q'_c (x) ctx 2^-4)
    FFN 1 / FFN 2    W1' a x~ ,  W2 / (2a) g
(k_colstats, k_rowfin, k_colfin and the head are fp32 VALU on the device: float64 here.)

Formats: bf16 (rounds 1-5) | fp16 with subnormals (what gfx950 does: tools/f16_probe.hip) | fp16 with subnormal limbs
flushed to zero (what a flushing matrix core would do) | flushed + every operand scaled by the power of two that brings
its largest entry to 2^13 (the static-scale fallback the judge proposed).  Reported: max |d - exact| over the pairs.
    python tests/dev/f16_split_study.py"""
import os, sys
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import oracle.pf_oracle as O                                   # noqa: E402
from phyloformer_amd import weights as Wt                      # noqa: E402
from phyloformer_amd.msa_sim import simulate_batch             # noqa: E402

A = np.sqrt(np.log2(np.e) / 2)


def bf16(x):
    m, e = np.frexp(x)
    return np.ldexp(np.round(m * 256.0) / 256.0, e)


def f16(x, flush):
    with np.errstate(over="ignore"):
        h = np.asarray(x, np.float64).astype(np.float16).astype(np.float64)
    assert np.isfinite(h).all(), "fp16 overflow"
    if flush:
        h = np.where(np.abs(h) < 2.0 ** -14, 0.0, h)
    return h


def split(x, fmt):
    if fmt == "exact":
        return x, np.zeros_like(x)
    if fmt == "bf16":
        hi = bf16(x)
        return hi, bf16(x - hi)
    flush = fmt != "f16"
    s = 1.0
    if fmt == "f16 flushed, scaled":
        mx = float(np.abs(x).max())
        s = 2.0 ** (13 - np.ceil(np.log2(mx))) if mx > 0 else 1.0
    hi = f16(x * s, flush)
    lo = f16(x * s - hi, flush)
    return hi / s, lo / s


def mm3(x, W, fmt):
    """x[..., K] @ W[M, K].T with both operands split, lo * lo dropped"""
    x2 = np.ascontiguousarray(x).reshape(-1, x.shape[-1])
    xh, xl = split(x2, fmt)
    Wh, Wl = split(W, fmt)
    out = xh @ Wh.T + xh @ Wl.T + xl @ Wh.T
    return out.reshape(x.shape[:-1] + (W.shape[0],))


def norm(x):
    mu = x.mean(-1, keepdims=True)
    xc = x - mu
    return xc / np.sqrt((xc * xc).mean(-1, keepdims=True) + O.LN_EPS)


def gelu2a(xa):
    """2 a gelu(h) for xa = a h (pf_device.hip.h::gelu_scaled, exact here)"""
    h = xa / A
    return 2 * A * O.gelu(h)


def fold(w, p, a):
    g, b = w[p + f"{a}_norm.weight"], w[p + f"{a}_norm.bias"]
    out = {}
    for n in ("q", "k", "v"):
        Wm = w[p + f"{a}_attention.{n}_proj.weight"]
        out[n] = (Wm * g[None, :], w[p + f"{a}_attention.{n}_proj.bias"] + Wm @ b)
    return out


def device_forward(w, idx, fmt):
    N, L = idx.shape
    table = O.embedding_table(w, np.dtype(np.float64))
    e = table[idx]
    pi, pj = O.pair_index(N)
    x = e[pi] + e[pj]                                                     # [P, L, 64]
    P = x.shape[0]
    for k in range(6):
        p = f"attention_blocks.{k}."
        fr, fc = fold(w, p, "row"), fold(w, p, "col")
        xn = norm(x)
        f0 = "exact" if k == 0 else fmt                                  # block 0: the residue-pair table (fp64-built)
        q = O.elu_plus_one(mm3(xn, fr["q"][0], f0) + fr["q"][1])
        kk = O.elu_plus_one(mm3(xn, fr["k"][0], f0) + fr["k"][1])
        v = mm3(xn, fr["v"][0], f0)
        skv = (kk[..., :, None] * v.reshape(P, L, 4, 16)).sum(1)         # [P, 4, 16]
        sq, sk = q.sum(1), kk.sum(1)                                     # [P, 4]
        vbar = (skv + fr["v"][1].reshape(4, 16)[None] * sk[..., None]) / sk[..., None]
        Wo = w[p + "row_attention.out_proj.weight"].reshape(64, 4, 16)
        mbase = np.einsum("chd,phd->phc", Wo, vbar)                      # [P, 4, 64]
        rq = L / sq                                                      # [P, 4]
        bo = w[p + "row_attention.out_proj.bias"]
        # row mix per pair: A = M_base^T [64 x 4], B = q' rq [L x 4]
        qb = q * rq[:, None, :]
        y = np.empty_like(x)
        for r in range(P):
            y[r] = mm3(qb[r], mbase[r].T, fmt)
        x_exact_row = x + bo + np.einsum("plh,phc->plc", qb, mbase)      # what k_colstats forms in fp32 from mrow
        x = x + bo + y
        # column statistics (VALU on the device)
        xc = norm(x_exact_row)
        qc = O.elu_plus_one(xc @ fc["q"][0].T + fc["q"][1])
        kc = O.elu_plus_one(xc @ fc["k"][0].T + fc["k"][1])
        z = np.einsum("plh,plc->lhc", kc, xc)
        skc, sqc = kc.sum(0), qc.sum(0)
        skv = np.einsum("hdc,lhc->lhd", fc["v"][0].reshape(4, 16, 64), z) + fc["v"][1].reshape(4, 16)[None] * skc[..., None]
        ctx = (skv / skc[..., None] * (P / sqc)[..., None]).reshape(L, 64)
        o = np.repeat(qc, 16, axis=-1) * ctx[None] / 16.0                # q'_c (x) ctx 2^-4
        x = x + w[p + "col_attention.out_proj.bias"] + mm3(o, w[p + "col_attention.out_proj.weight"] * 16.0, fmt)
        # FFN
        g, b = w[p + "ffn_norm.weight"], w[p + "ffn_norm.bias"]
        W1, b1 = w[p + "ffn.0.weight"].reshape(256, 64), w[p + "ffn.0.bias"]
        W2, b2 = w[p + "ffn.3.weight"].reshape(64, 256), w[p + "ffn.3.bias"]
        h = mm3(norm(x), W1 * g[None, :] * A, fmt) + (b1 + W1 @ b) * A
        x = x + b2 + mm3(gelu2a(h), W2 / (2 * A), fmt)
    hw, hb = w["pwFNN.0.weight"].reshape(-1), w["pwFNN.0.bias"]
    return O.softplus(x @ hw + hb).mean(1)


def cases():
    load = lambda n: {k: v.astype(np.float64) for k, v in Wt.load_weights(os.path.join(REPO, f"models/{n}.ckpt")).tensors.items()}  # noqa: E731
    wpf, wind = load("pf"), load("pf_indel")
    rng = np.random.default_rng(6)
    sim = simulate_batch(1, 12, 24, seed=4, gaps=True)[0]
    noisy = simulate_batch(1, 8, 48, seed=9)[0]
    m = rng.random(noisy.shape) < 0.75
    noisy = np.where(m, rng.integers(0, 22, noisy.shape), noisy).astype(np.uint8)
    return [("5x16 uniform random (pf)", wpf, rng.integers(0, 22, (5, 16)).astype(np.uint8)),
            ("20x20 uniform random (pf)", wpf, rng.integers(0, 20, (20, 20)).astype(np.uint8)),
            ("40x33 uniform over 22 (pf_indel)", wind, rng.integers(0, 22, (40, 33)).astype(np.uint8)),
            ("12x24 simulated, gapped (pf_indel)", wind, sim),
            ("8x48, 75 % noise (pf)", wpf, noisy),
            ("20x100 two-letter (pf)", wpf, rng.choice([3, 11], (20, 100)).astype(np.uint8)),
            ("24x120 simulated (pf)", wpf, simulate_batch(1, 24, 120, seed=2)[0])]


def main():
    fmts = ["bf16", "f16", "f16 flushed", "f16 flushed, scaled"]
    print(f"{'input (checkpoint)':38s} {'fp32 reference':>15s} " + " ".join(f"{f:>20s}" for f in fmts) + "   max |d|")
    for label, w, idx in cases():
        exact = device_forward(w, idx, "exact")
        ref64 = O.forward(w, idx, dtype=np.float64)
        assert np.abs(exact - ref64).max() <= 1e-9 * max(1.0, np.abs(ref64).max()), "the restated device algebra is off"
        f32 = O.forward({k: v.astype(np.float32) for k, v in w.items()}, idx)
        row = [float(np.abs(device_forward(w, idx, f) - exact).max()) for f in fmts]
        print(f"{label:38s} {float(np.abs(f32 - ref64).max()):15.2e} " + " ".join(f"{d:20.2e}" for d in row) +
              f"   {np.abs(ref64).max():.3g}", flush=True)


if __name__ == "__main__":
    main()
