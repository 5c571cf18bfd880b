"""Distance vector → square PHYLIP matrix text.

Host-side mirror of ``vec_to_phylip`` (/root/reference/infer_alns.py:14-25):
the ``P = N(N-1)/2`` predictions fill the strict upper triangle in
``triu_indices(N, N, 1)`` order (pair ``(i, j)``, ``i < j``, lexicographic —
the same enumeration as ``seq2pair``, /root/reference/phyloformer/model.py:13-17),
the matrix is symmetrised and every row is written as ``id`` followed by
``" %.10f"`` per entry; the first line is ``N``.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np


def pair_indices(n: int) -> Tuple[np.ndarray, np.ndarray]:
    """``(i, j)`` index arrays of the ``n(n-1)/2`` pairs, lexicographic, ``i < j``."""
    return np.triu_indices(n, k=1)


def vec_to_matrix(preds: np.ndarray, n: int) -> np.ndarray:
    preds = np.asarray(preds)
    if preds.ndim == 0:
        preds = preds.reshape(1)  # N == 2: the reference squeezes to a 0-dim tensor
    i, j = pair_indices(n)
    if preds.shape[-1] != i.size:
        raise ValueError(f"expected {i.size} distances for {n} sequences, got {preds.shape}")
    dm = np.zeros(preds.shape[:-1] + (n, n), dtype=preds.dtype)
    dm[..., i, j] = preds
    return dm + np.swapaxes(dm, -1, -2)


def vec_to_phylip(preds: np.ndarray, ids: Sequence[str]) -> Tuple[np.ndarray, str]:
    n = len(ids)
    dm = vec_to_matrix(np.asarray(preds), n)
    lines = [f"{n}\n"]
    for name, row in zip(ids, dm):
        lines.append(f"{name} " + " ".join(f"{float(x):.10f}" for x in row) + "\n")
    return dm, "".join(lines)
