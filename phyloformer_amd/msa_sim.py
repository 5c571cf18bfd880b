"""Synthetic "LG+GC-like" alignments for benchmarks and parity fixtures.

The reference simulates with ``iqtree2 --alisim`` under LG+G
(/root/reference/alisim.py:83-114); that binary and the LG matrix are not
available offline, so this module is the build's own seeded
generator:

* random binary tree grown by successive leaf splitting, exponential branch
  lengths, rescaled so the tree diameter is log-uniform in [0.1, 5];
* per-site rate ~ Gamma(α, 1/α), α log-uniform in [0.3, 3] — the
  "gamma-continuous" (GC) part (alisim.py:23-26 samples α from an empirical
  list whose median is ≈1);
* substitutions with equal exchangeabilities and the stationary frequencies
  of the 20 reference test MSAs (an F81-style process, rate-normalised to one
  expected substitution per unit branch length);
* optional indels: Poisson events per branch (≈1 % per site per unit length),
  geometric block lengths, inherited by the whole subtree (cf. alisim.py:86-89).

The device computation is dense and data-independent, so timing does not depend
on the substitution model; only the error statistic does.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from .fasta import ALPHABET, GAP_INDEX

# amino-acid frequencies of data/testdata/msas (20 files), alphabet order ARNDCQEGHILKMFPSTWYV
PI = np.array([0.0820, 0.0547, 0.0432, 0.0534, 0.0120, 0.0404, 0.0703, 0.0589, 0.0217, 0.0610,
               0.0973, 0.0622, 0.0243, 0.0418, 0.0423, 0.0625, 0.0533, 0.0117, 0.0323, 0.0748])
PI = PI / PI.sum()


def random_tree(rng: np.random.Generator, n_leaves: int) -> Tuple[np.ndarray, np.ndarray]:
    """Return ``(parent, length)`` arrays of a rooted binary tree with ``2n-1`` nodes.

    Node 0 is the root; leaves are the nodes that never get split.
    """
    parent = [-1]
    length = [0.0]
    leaves = [0]
    while len(leaves) < n_leaves:
        k = int(rng.integers(len(leaves)))
        node = leaves.pop(k)
        for _ in range(2):
            parent.append(node)
            length.append(float(rng.exponential(1.0)))
            leaves.append(len(parent) - 1)
    return np.array(parent), np.array(length)


def _leaf_depths(parent, length):
    depth = np.zeros(len(parent))
    for v in range(1, len(parent)):       # children are always created after parents
        depth[v] = depth[parent[v]] + length[v]
    is_leaf = np.ones(len(parent), bool)
    is_leaf[parent[1:]] = False
    return depth, is_leaf


def simulate_alignment(n_seqs: int, n_sites: int, seed: int = 0, gaps: bool = False,
                       rng: Optional[np.random.Generator] = None) -> np.ndarray:
    """One alignment as ``uint8[n_seqs, n_sites]`` residue indices (0..19, 21 for gaps)."""
    if rng is None:
        rng = np.random.Generator(np.random.PCG64(seed))
    parent, length = random_tree(rng, n_seqs)
    depth, is_leaf = _leaf_depths(parent, length)
    # crude diameter: twice the deepest leaf; rescale to the target diameter
    diam = float(np.exp(rng.uniform(np.log(0.1), np.log(5.0))))
    length = length * (diam / max(2.0 * depth[is_leaf].max(), 1e-9))
    alpha = float(np.exp(rng.uniform(np.log(0.3), np.log(3.0))))
    rates = rng.gamma(alpha, 1.0 / alpha, size=n_sites)
    mu = 1.0 / (1.0 - float((PI ** 2).sum()))
    n_nodes = len(parent)
    seqs = np.empty((n_nodes, n_sites), dtype=np.uint8)
    seqs[0] = rng.choice(20, size=n_sites, p=PI)
    gapmask = np.zeros((n_nodes, n_sites), dtype=bool)
    for v in range(1, n_nodes):
        p_change = 1.0 - np.exp(-mu * rates * length[v])
        redraw = rng.random(n_sites) < p_change
        fresh = rng.choice(20, size=n_sites, p=PI)
        seqs[v] = np.where(redraw, fresh, seqs[parent[v]])
        gapmask[v] = gapmask[parent[v]]
        if gaps:
            n_ev = rng.poisson(0.01 * n_sites * length[v])
            for _ in range(int(n_ev)):
                start = int(rng.integers(n_sites))
                blk = int(rng.geometric(0.25))
                gapmask[v, start:start + blk] = True
    out = seqs[is_leaf].copy()
    if gaps:
        out[gapmask[is_leaf]] = GAP_INDEX
    return out


def simulate_batch(n_aln: int, n_seqs: int, n_sites: int, seed: int = 0,
                   gaps: bool = False) -> np.ndarray:
    """``uint8[n_aln, n_seqs, n_sites]`` — alignment ``b`` uses stream ``seed`` advanced ``b`` times."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return np.stack([simulate_alignment(n_seqs, n_sites, gaps=gaps, rng=rng) for _ in range(n_aln)])


def to_fasta(idx: np.ndarray, ids: Optional[List[str]] = None) -> str:
    ids = ids or [f"T{i + 1}" for i in range(idx.shape[0])]
    lut = np.frombuffer(ALPHABET, dtype=np.uint8)
    return "".join(f">{name}\n{lut[row].tobytes().decode()}\n" for name, row in zip(ids, idx))
