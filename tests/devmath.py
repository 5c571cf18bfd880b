"""Expected values of the device path's intermediate buffers, from oracle activations (float64).

The device works with LayerNorm-without-affine activations and projections with
gamma/beta folded in (phyloformer_amd/csrc/pf_device.hip.h header); these helpers
restate that algebra in numpy so a GPU test can localise a kernel bug to one buffer.
"""
import numpy as np

from oracle import pf_oracle as O


def _norm(x):
    mu = x.mean(-1, keepdims=True)
    xc = x - mu
    return xc / np.sqrt((xc * xc).mean(-1, keepdims=True) + O.LN_EPS)


def _fold(w, p, a):
    g, b = w[p + f"{a}_norm.weight"].astype(np.float64), w[p + f"{a}_norm.bias"].astype(np.float64)
    out = {}
    for n in ("q", "k", "v"):
        Wm = w[p + f"{a}_attention.{n}_proj.weight"].astype(np.float64)
        out[n] = (Wm * g[None, :], w[p + f"{a}_attention.{n}_proj.bias"].astype(np.float64) + Wm @ b)
    return out


def expected_srow(w, k, x_in):
    """srow feeding block k from the residual stream entering block k: [P][72]."""
    f = _fold(w, f"attention_blocks.{k}.", "row")
    xn = _norm(x_in.astype(np.float64))
    q = O.elu_plus_one(xn @ f["q"][0].T + f["q"][1])
    kk = O.elu_plus_one(xn @ f["k"][0].T + f["k"][1])
    v = xn @ f["v"][0].T                                   # no bias: added in k_rowfin
    skv = (kk[..., :, None] * v.reshape(v.shape[:-1] + (4, 16))).sum(1).reshape(-1, 64)
    return np.concatenate([skv, q.sum(1), kk.sum(1)], axis=1), q


def expected_mrow(w, k, srow, L):
    p = f"attention_blocks.{k}."
    f = _fold(w, p, "row")
    Wo = w[p + "row_attention.out_proj.weight"].astype(np.float64)
    skv, sq, sk = srow[:, :64], srow[:, 64:68], srow[:, 68:72]
    skr = np.repeat(sk, 16, axis=1)
    ctx = (skv + f["v"][1][None, :] * skr) / skr * (L / np.repeat(sq, 16, axis=1))
    M = np.einsum("chd,phd->phc", Wo.reshape(64, 4, 16), ctx.reshape(-1, 4, 16))
    bias = np.broadcast_to(w[p + "row_attention.out_proj.bias"].astype(np.float64), (M.shape[0], 1, 64))
    return np.concatenate([M, bias], axis=1)               # [P][5][64]


def expected_ctx(w, k, x_row):
    """ctx of block k from the stream after block k's row attention: [L][64], and q'_col."""
    f = _fold(w, f"attention_blocks.{k}.", "col")
    xn = _norm(x_row.astype(np.float64))
    q = O.elu_plus_one(xn @ f["q"][0].T + f["q"][1])
    kk = O.elu_plus_one(xn @ f["k"][0].T + f["k"][1])
    z = np.einsum("plh,plc->lhc", kk, xn)
    sk, sq = kk.sum(0), q.sum(0)                           # [L][4]
    skv = np.einsum("hdc,lhc->lhd", f["v"][0].reshape(4, 16, 64), z) + f["v"][1].reshape(4, 16)[None] * sk[..., None]
    P = x_row.shape[0]
    ctx = skv / sk[..., None] * (P / sq)[..., None]
    return ctx.reshape(-1, 64), q
