"""Neighbour joining (Saitou & Nei 1987) for the ``--trees`` flag of the CLI.

The reference delegates to ``skbio.tree.nj`` (/root/reference/infer_alns.py:62-64,
120-123); scikit-bio is not installed in this image, so the tree TEXT of skbio
is unpinned, but the algorithm is pinned against a reference-held neighbour
joining: FastME ``-m N`` from the reference checkout gives the same topology
(RF = 0) and, without clamping, the same branch lengths (<= 3e-8) on the reference's own distance
matrices of all 20 test MSAs (tests/golden/nj_fastme.json,
tests/test_treecmp.py::test_nj_matches_fastme_nj_goldens).  This is the textbook
algorithm with the scikit-bio defaults the reference relies on: negative branch
lengths are clamped to zero and the final three clusters are joined at a
trifurcating root.
O(N³) on N ≤ 200 taxa — host work, not on the device path.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def neighbor_joining(dm: np.ndarray, ids: Sequence[str], clamp_negative: bool = True) -> str:
    """Return a Newick string (terminated by ``;`` and a newline)."""
    d = np.array(dm, dtype=np.float64)
    n = d.shape[0]
    if d.shape != (n, n) or len(ids) != n:
        raise ValueError("distance matrix must be square and match ids")
    labels: List[str] = [str(i) for i in ids]
    if n == 1:
        return f"({labels[0]});\n"
    if n == 2:
        return f"({labels[0]}:{d[0, 1] / 2:.6g},{labels[1]}:{d[0, 1] / 2:.6g});\n"

    def fmt(x: float) -> str:
        if clamp_negative and x < 0:
            x = 0.0
        return repr(float(x))

    active = list(range(n))
    while len(active) > 3:
        m = len(active)
        sub = d[np.ix_(active, active)]
        r = sub.sum(axis=1)
        q = (m - 2) * sub - r[:, None] - r[None, :]
        np.fill_diagonal(q, np.inf)
        a, b = np.unravel_index(np.argmin(q), q.shape)
        if a > b:
            a, b = b, a
        ia, ib = active[a], active[b]
        dab = sub[a, b]
        la = 0.5 * dab + (r[a] - r[b]) / (2 * (m - 2))
        lb = dab - la
        new_label = f"({labels[ia]}:{fmt(la)},{labels[ib]}:{fmt(lb)})"
        # distances from the new node to every other active node
        dn = 0.5 * (d[ia, :] + d[ib, :] - dab)
        d[ia, :] = dn
        d[:, ia] = dn
        d[ia, ia] = 0.0
        labels[ia] = new_label
        active.pop(b)
    i, j, k = active
    li = 0.5 * (d[i, j] + d[i, k] - d[j, k])
    lj = 0.5 * (d[i, j] + d[j, k] - d[i, k])
    lk = 0.5 * (d[i, k] + d[j, k] - d[i, j])
    return f"({labels[i]}:{fmt(li)},{labels[j]}:{fmt(lj)},{labels[k]}:{fmt(lk)});\n"
