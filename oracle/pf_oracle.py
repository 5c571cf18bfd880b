"""ORACLE — test infrastructure only, never on the product path.

NumPy restatement of the Phyloformer inference forward pass
(one-hot MSA → pairwise distance vector).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module; ``phyloformer_amd`` itself never does.

Parity status: PINNED.  The reference ships no golden vectors for this path
(SURVEY.md §8c), so the oracle is pinned against outputs of the reference
itself, produced in the build container by ``oracle/gen_golden.py`` (which
imports /root/reference on CPU) and committed under ``tests/golden/``:
end-to-end distances for the 20 test MSAs × 5 checkpoints, sub-block taps on a
tiny case, and the synthetic benchmark configurations.
``tests/test_oracle.py`` checks this file against them.

Layout: *token-major* ``x[P, L, E]`` (pair, site, channel) instead of the
reference's channels-first ``[B, E, P, L]``; the arithmetic per element is the
same.  Each function cites the reference lines it restates.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import numpy as np

try:  # vectorised erf; scipy is present in the image, math.erf is the fallback
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)

LN_EPS = 1e-5          # nn.LayerNorm default (model.py:64-66)
MAX_SEQS = 200         # SEQ2PAIR = seq2pair(200), model.py:39


def pair_index(n: int):
    """Pairs (i, j), i < j, lexicographic — model.py:13-17 (= triu_indices(n, n, 1))."""
    return np.triu_indices(n, k=1)


def embedding_table(w: Dict[str, np.ndarray], dtype) -> np.ndarray:
    """1×1 conv 22→E on a one-hot + bias + ReLU ≡ table lookup (model.py:138-143,173)."""
    W = w["embedding_block.0.weight"].astype(dtype)          # [E, 22]
    b = w["embedding_block.0.bias"].astype(dtype)            # [E]
    return np.maximum(W.T + b[None, :], 0)                   # [22, E]


def _mm(x, W):
    """``x @ W.T`` for ``x[..., K]``, ``W[M, K]`` through one 2-D BLAS call (numpy's
    stacked matmul is ~20x slower for these shapes)."""
    x2 = np.ascontiguousarray(x).reshape(-1, x.shape[-1])
    return (x2 @ np.ascontiguousarray(W.T)).reshape(x.shape[:-1] + (W.shape[0],))


def layer_norm(x, g, b):
    """nn.LayerNorm(E): biased variance, eps inside the sqrt (model.py:90,96,102)."""
    mu = x.mean(axis=-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True)
    return xc / np.sqrt(var + x.dtype.type(LN_EPS)) * g + b


def elu_plus_one(x):
    """attention.py:179-180: elu(x) + 1 = x + 1 (x > 0) or exp(x) (x <= 0)."""
    return np.where(x > 0, x + 1, np.exp(np.minimum(x, 0)))


def gelu(x):
    """nn.GELU() default = exact erf form (model.py:77)."""
    return x * x.dtype.type(0.5) * (1 + _erf(x / x.dtype.type(math.sqrt(2.0)))).astype(x.dtype)


def softplus(x):
    """nn.Softplus(beta=1, threshold=20) (model.py:163)."""
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20))))


def attention_stats(xn, w, pre, axis, n_heads):
    """Statistics of ScaledLinearAttention over ``axis`` (attention.py:163-190).

    Returns q' (per token) and the three sums the block needs:
    ``S_q[h] = Σ q'``, ``S_k[h] = Σ k'``, ``S_kv[h, d] = Σ k'·v``.
    These are the quantities a site-sharded run all-reduces.
    """
    E = xn.shape[-1]
    D = E // n_heads
    dt = xn.dtype
    q = elu_plus_one(_mm(xn, w[pre + "q_proj.weight"].astype(dt)) + w[pre + "q_proj.bias"].astype(dt))
    k = elu_plus_one(_mm(xn, w[pre + "k_proj.weight"].astype(dt)) + w[pre + "k_proj.bias"].astype(dt))
    v = _mm(xn, w[pre + "v_proj.weight"].astype(dt)) + w[pre + "v_proj.bias"].astype(dt)
    vh = v.reshape(v.shape[:-1] + (n_heads, D))               # channel h*D+d (attention.py:175)
    s_q = q.sum(axis=axis, keepdims=True)
    s_k = k.sum(axis=axis, keepdims=True)
    s_kv = (k[..., None] * vh).sum(axis=axis, keepdims=True)
    return q, s_q, s_k, s_kv


def attention_apply(q, s_q, s_k, s_kv, w, pre, count):
    """attention.py:183-195: q/mean(q) · (Σk'v / Σk') → merge heads → out_proj."""
    dt = q.dtype
    qn = q / (s_q / dt.type(count))                          # q / q.mean(dim=-2)
    ctx = s_kv / s_k[..., None]                              # (k / k.sum)ᵀ @ v
    o = qn[..., None] * ctx                                  # [..., H, D]
    o = o.reshape(o.shape[:-2] + (o.shape[-2] * o.shape[-1],))
    return _mm(o, w[pre + "out_proj.weight"].astype(dt)) + w[pre + "out_proj.bias"].astype(dt)


def _all_sum(parts):
    out = parts[0].copy()
    for p in parts[1:]:
        out += p
    return out


def forward(weights: Dict[str, np.ndarray], idx: np.ndarray, *, n_blocks: int = 6,
            n_heads: int = 4, dtype=np.float32, shards: int = 1,
            tap: Optional[Callable[[str, np.ndarray], None]] = None,
            max_seqs: Optional[int] = MAX_SEQS) -> np.ndarray:
    """Phyloformer.forward (model.py:166-187) for one alignment.

    idx: ``uint8[N, L]`` residue indices.  Returns ``dtype[P]``.
    ``shards > 1`` evaluates the site-sharded algorithm (each shard owns a
    contiguous block of sites; row-attention statistics and the final per-pair
    site sums are summed across shards) — the CPU stand-in for the RCCL path.
    """
    idx = np.asarray(idx)
    if idx.ndim != 2:
        raise ValueError(f"idx must be [N, L], got shape {idx.shape}")
    N, L = idx.shape
    if max_seqs is not None and N > max_seqs:
        # adaptable_seq2pair, model.py:24-28
        raise ValueError(f"n_seqs must be smaller or equal to {max_seqs} "
                         "(or pre-compute a larger global_seq2pair)")
    if N < 2 or L < 1:
        raise ValueError(f"need at least 2 sequences and 1 site, got N={N}, L={L}")
    if idx.max(initial=0) >= 22:
        raise ValueError("residue index out of range")
    dt = np.dtype(dtype)
    w = weights
    table = embedding_table(w, dt)
    e = table[idx]                                           # [N, L, E]
    pi, pj = pair_index(N)
    x_full = e[pi] + e[pj]                                   # model.py:175
    step = -(-L // max(shards, 1))                          # ceil-split, like the device path
    bounds = np.minimum(np.arange(max(shards, 1) + 1) * step, L)
    xs = [x_full[:, bounds[s]:bounds[s + 1], :] for s in range(len(bounds) - 1)]
    xs = [x for x in xs if x.shape[1] > 0]
    P = x_full.shape[0]
    if tap:
        tap("embed", np.concatenate(xs, axis=1))

    for b in range(n_blocks):
        p = f"attention_blocks.{b}."
        # ---- row attention: reduce over sites (axis 1), model.py:89-92
        g, bb = w[p + "row_norm.weight"].astype(dt), w[p + "row_norm.bias"].astype(dt)
        st = [attention_stats(layer_norm(x, g, bb), w, p + "row_attention.", 1, n_heads) for x in xs]
        s_q = _all_sum([s[1] for s in st])                   # ← all-reduce #b
        s_k = _all_sum([s[2] for s in st])
        s_kv = _all_sum([s[3] for s in st])
        xs = [x + attention_apply(s[0], s_q, s_k, s_kv, w, p + "row_attention.", L)
              for x, s in zip(xs, st)]
        if tap:
            tap(f"block{b}.row", np.concatenate(xs, axis=1))
        # ---- column attention: reduce over pairs (axis 0), model.py:95-98 — shard-local
        g, bb = w[p + "col_norm.weight"].astype(dt), w[p + "col_norm.bias"].astype(dt)
        nx = []
        for x in xs:
            q, a, k_, kv = attention_stats(layer_norm(x, g, bb), w, p + "col_attention.", 0, n_heads)
            nx.append(x + attention_apply(q, a, k_, kv, w, p + "col_attention.", P))
        xs = nx
        if tap:
            tap(f"block{b}.col", np.concatenate(xs, axis=1))
        # ---- feed-forward, model.py:101-104
        g, bb = w[p + "ffn_norm.weight"].astype(dt), w[p + "ffn_norm.bias"].astype(dt)
        W1, b1 = w[p + "ffn.0.weight"].astype(dt), w[p + "ffn.0.bias"].astype(dt)
        W2, b2 = w[p + "ffn.3.weight"].astype(dt), w[p + "ffn.3.bias"].astype(dt)
        xs = [x + (_mm(gelu(_mm(layer_norm(x, g, bb), W1) + b1), W2) + b2) for x in xs]
        if tap:
            tap(f"block{b}.ffn", np.concatenate(xs, axis=1))

    hw, hb = w["pwFNN.0.weight"].astype(dt).reshape(-1), w["pwFNN.0.bias"].astype(dt)
    logits = [_mm(x, hw[None, :])[..., 0] + hb for x in xs]                       # model.py:182
    if tap:
        tap("logits", np.concatenate(logits, axis=1))
    part = [softplus(z).sum(axis=1) for z in logits]
    d = _all_sum(part) / dt.type(L)                          # model.py:185; final all-reduce
    if tap:
        tap("dist", d)
    return d.astype(dt)


def forward_rank(weights: Dict[str, np.ndarray], idx_local: np.ndarray, L_total: int,
                 allreduce: Callable[[np.ndarray], np.ndarray], *, n_blocks: int = 6,
                 n_heads: int = 4, dtype=np.float32) -> np.ndarray:
    """One rank of the site-sharded forward: ``idx_local`` is ``uint8[N, Lloc]`` (Lloc may be 0).

    ``allreduce(a)`` must return the element-wise sum of ``a`` over all ranks.  It is called
    ``n_blocks + 1`` times with arrays of identical shape on every rank: the fused row
    statistics ``[P, 72]`` (S_kv | S_q | S_k) once per block and the site sums ``[P]`` at the
    end — the collective schedule of the device path.
    """
    idx_local = np.asarray(idx_local)
    N, Lloc = idx_local.shape
    dt = np.dtype(dtype)
    w = weights
    pi, pj = pair_index(N)
    P = len(pi)
    e = embedding_table(w, dt)[idx_local]
    x = e[pi] + e[pj]                                        # [P, Lloc, E]
    for b in range(n_blocks):
        p = f"attention_blocks.{b}."
        g, bb = w[p + "row_norm.weight"].astype(dt), w[p + "row_norm.bias"].astype(dt)
        q, s_q, s_k, s_kv = attention_stats(layer_norm(x, g, bb), w, p + "row_attention.", 1, n_heads)
        fused = np.concatenate([s_kv.reshape(P, -1), s_q.reshape(P, -1), s_k.reshape(P, -1)], axis=1)
        fused = allreduce(np.ascontiguousarray(fused, dtype=dt))
        s_kv = fused[:, :-2 * n_heads].reshape(P, 1, n_heads, -1)
        s_q = fused[:, -2 * n_heads:-n_heads].reshape(P, 1, n_heads)
        s_k = fused[:, -n_heads:].reshape(P, 1, n_heads)
        if Lloc:
            x = x + attention_apply(q, s_q, s_k, s_kv, w, p + "row_attention.", L_total)
        g, bb = w[p + "col_norm.weight"].astype(dt), w[p + "col_norm.bias"].astype(dt)
        if Lloc:
            q, a, k_, kv = attention_stats(layer_norm(x, g, bb), w, p + "col_attention.", 0, n_heads)
            x = x + attention_apply(q, a, k_, kv, w, p + "col_attention.", P)
        g, bb = w[p + "ffn_norm.weight"].astype(dt), w[p + "ffn_norm.bias"].astype(dt)
        W1, b1 = w[p + "ffn.0.weight"].astype(dt), w[p + "ffn.0.bias"].astype(dt)
        W2, b2 = w[p + "ffn.3.weight"].astype(dt), w[p + "ffn.3.bias"].astype(dt)
        x = x + (_mm(gelu(_mm(layer_norm(x, g, bb), W1) + b1), W2) + b2)
    hw, hb = w["pwFNN.0.weight"].astype(dt).reshape(-1), w["pwFNN.0.bias"].astype(dt)
    part = softplus(_mm(x, hw[None, :])[..., 0] + hb).sum(axis=1).astype(dt)
    return (allreduce(np.ascontiguousarray(part)) / dt.type(L_total)).astype(dt)


def forward_batch(weights, idx: np.ndarray, **kw) -> np.ndarray:
    """``uint8[B, N, L]`` → ``[B, P]``; alignments are independent (model.py:166-187)."""
    idx = np.asarray(idx)
    if idx.ndim == 2:
        return forward(weights, idx, **kw)
    return np.stack([forward(weights, a, **kw) for a in idx])


# ---------------------------------------------------------------------------
# Restructured algebra — what the device kernels evaluate.  Mathematically
# identical to the functions above; used by tests to bound the re-association
# error of the device formulation in fp32 and to localise kernel bugs.
# ---------------------------------------------------------------------------

def collapsed_stats(xn, w, pre, axis):
    """q', k' and Z[h, c] = Σ k'[h]·x̂[c] — the V projection pulled out of the sum."""
    dt = xn.dtype
    q = elu_plus_one(_mm(xn, w[pre + "q_proj.weight"].astype(dt)) + w[pre + "q_proj.bias"].astype(dt))
    k = elu_plus_one(_mm(xn, w[pre + "k_proj.weight"].astype(dt)) + w[pre + "k_proj.bias"].astype(dt))
    s_q = q.sum(axis=axis)
    s_k = k.sum(axis=axis)
    z = np.einsum("...h,...c->...hc", k, xn).sum(axis=axis)
    return q, s_q, s_k, z


def collapsed_mix(s_q, s_k, z, w, pre, count, n_heads):
    """M[h, c]: out_proj folded with the normalised context, so y = Σ_h q'[h]·M[h] + b_o."""
    dt = z.dtype
    Wv, bv = w[pre + "v_proj.weight"].astype(dt), w[pre + "v_proj.bias"].astype(dt)
    Wo = w[pre + "out_proj.weight"].astype(dt)
    E = Wv.shape[0]
    D = E // n_heads
    Wvh = Wv.reshape(n_heads, D, E)
    s_kv = np.einsum("hdc,...hc->...hd", Wvh, z) + bv.reshape(n_heads, D) * s_k[..., None]
    ctx = s_kv / s_k[..., None] * (dt.type(count) / s_q)[..., None]
    Woh = Wo.reshape(E, n_heads, D)
    return np.einsum("chd,...hd->...hc", Woh, ctx)           # [..., H, E]
