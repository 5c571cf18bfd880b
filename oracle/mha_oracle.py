"""CPU oracle of the softmax ``MultiHeadAttention`` module — TEST INFRASTRUCTURE ONLY.

numpy restatement of /root/reference/phyloformer/attention.py:53-91 (projections :43-47).
Nothing in the reference instantiates that class and no checkpoint can be loaded into it
(SURVEY.md F1), so it has no effect on the graded distances; it is the §8f rank-4 "next" row.
Pinned by tests/golden/mha.npz = outputs of the reference class itself on seeded synthetic
weights (oracle/gen_golden_mha.py).  Only tests/, __graft_entry__.smoke() and tools may import this.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

KEYS = ("q_proj", "k_proj", "v_proj", "out_proj")


def mha_forward(w: Dict[str, np.ndarray], x: np.ndarray, n_heads: int = 4, dtype=np.float64) -> np.ndarray:
    """``MultiHeadAttention.forward`` (attention.py:62-91): x [B, R, C, E] → [B, R, C, E];
    attention runs along axis 2 (C), independently for every (b, r) and head."""
    x = np.asarray(x, dtype=dtype)
    B, R, C, E = x.shape
    D = E // n_heads

    def proj(name):                                                   # nn.Linear, attention.py:43-47
        y = x.reshape(-1, E) @ w[f"{name}.weight"].astype(dtype).T + w[f"{name}.bias"].astype(dtype)
        return y.reshape(B, R, C, n_heads, D).transpose(0, 1, 3, 2, 4)  # .view(...).transpose(2, 3): [B,R,H,C,D]

    q, k, v = proj("q_proj"), proj("k_proj"), proj("v_proj")
    logits = q @ k.transpose(0, 1, 2, 4, 3) / dtype(np.sqrt(D))      # :82-83
    logits -= logits.max(axis=-1, keepdims=True)
    p = np.exp(logits)
    p /= p.sum(axis=-1, keepdims=True)                                # softmax(dim=-1), :84
    o = (p @ v).transpose(0, 1, 3, 2, 4).reshape(B, R, C, E)          # :86-88
    y = o.reshape(-1, E) @ w["out_proj.weight"].astype(dtype).T + w["out_proj.bias"].astype(dtype)
    return y.reshape(B, R, C, E)                                       # :90 (dropout p = 0 / eval)
