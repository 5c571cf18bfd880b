"""Tree comparison for the end-to-end check.

The reference's README (/root/reference/README.md:83-99) turns the predicted
distance matrices into trees with FastME and compares them with the true trees
using ``phylocompare`` (a binary that is missing from the reference checkout).
This module is the build's own comparison: a Newick reader, the bipartition
(split) set of an unrooted tree, the Robinson-Foulds distance — plain and
normalised by the number of internal edges of both trees — and the
Kuhner-Felsenstein branch-score distance.  Host-side, O(N²) on N ≤ 200 taxa.
"""
from __future__ import annotations

import math
from typing import Dict, FrozenSet, List, Optional, Tuple


class Node:
    __slots__ = ("name", "length", "children")

    def __init__(self, name: Optional[str] = None, length: Optional[float] = None):
        self.name = name
        self.length = length
        self.children: List["Node"] = []

    def is_leaf(self) -> bool:
        return not self.children


def parse_newick(text: str) -> Node:
    """Parse one Newick tree (quoted labels, comments in [] and internal labels accepted)."""
    s = text.strip()
    if not s.endswith(";"):
        raise ValueError("Newick string must end with ';'")
    pos = 0

    def skip():
        nonlocal pos
        while pos < len(s):
            if s[pos].isspace():
                pos += 1
            elif s[pos] == "[":
                end = s.find("]", pos)
                if end < 0:
                    raise ValueError("unterminated comment")
                pos = end + 1
            else:
                break

    def label() -> Optional[str]:
        nonlocal pos
        skip()
        if pos < len(s) and s[pos] == "'":
            out = []
            pos += 1
            while True:
                if pos >= len(s):
                    raise ValueError("unterminated quoted label")
                if s[pos] == "'":
                    if pos + 1 < len(s) and s[pos + 1] == "'":
                        out.append("'")
                        pos += 2
                        continue
                    pos += 1
                    break
                out.append(s[pos])
                pos += 1
            return "".join(out)
        start = pos
        while pos < len(s) and s[pos] not in "(),:;[" and not s[pos].isspace():
            pos += 1
        return s[start:pos] or None

    def length() -> Optional[float]:
        nonlocal pos
        skip()
        if pos < len(s) and s[pos] == ":":
            pos += 1
            skip()
            start = pos
            while pos < len(s) and s[pos] not in "(),;[" and not s[pos].isspace():
                pos += 1
            return float(s[start:pos])
        return None

    def subtree() -> Node:
        nonlocal pos
        skip()
        node = Node()
        if pos < len(s) and s[pos] == "(":
            pos += 1
            while True:
                node.children.append(subtree())
                skip()
                if pos >= len(s):
                    raise ValueError("unbalanced parentheses")
                if s[pos] == ",":
                    pos += 1
                    continue
                if s[pos] == ")":
                    pos += 1
                    break
                raise ValueError(f"unexpected character {s[pos]!r} at {pos}")
        node.name = label()
        node.length = length()
        return node

    root = subtree()
    skip()
    if pos >= len(s) or s[pos] != ";":
        raise ValueError(f"trailing characters at {pos}")
    return root


def leaf_names(root: Node) -> List[str]:
    out, stack = [], [root]
    while stack:
        n = stack.pop()
        if n.is_leaf():
            out.append(n.name)
        else:
            stack.extend(reversed(n.children))
    return out


def splits(root: Node) -> Dict[FrozenSet[str], float]:
    """Non-trivial bipartitions of the unrooted tree → branch length (0 when absent).

    A split is stored as the side that does not contain the lexicographically
    smallest leaf, so rooted and unrooted drawings of the same tree agree; the
    two edges at a bifurcating root are one unrooted edge and their lengths add.
    Trivial splits (one leaf against the rest) are kept under a 1-element key so
    the branch-score distance can use them.
    """
    leaves = leaf_names(root)
    if len(set(leaves)) != len(leaves):
        raise ValueError("duplicate leaf names")
    universe = frozenset(leaves)
    anchor = min(leaves)
    out: Dict[FrozenSet[str], float] = {}

    def visit(n: Node) -> FrozenSet[str]:
        below = frozenset([n.name]) if n.is_leaf() else frozenset().union(*[visit(c) for c in n.children])
        if n is not root:
            side = universe - below if anchor in below else below
            if 0 < len(side) < len(universe):
                out[side] = out.get(side, 0.0) + (n.length or 0.0)
        return below

    import sys
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 10000))
    try:
        visit(root)
    finally:
        sys.setrecursionlimit(old)
    return out


def _internal(sp: Dict[FrozenSet[str], float], n_leaves: int) -> set:
    return {k for k in sp if 1 < len(k) < n_leaves - 1}


def robinson_foulds(a: Node, b: Node) -> Tuple[int, float]:
    """(RF, normalised RF): number of internal splits in exactly one tree, and that number divided
    by the total number of internal splits of both trees (0 = same topology, 1 = nothing shared)."""
    la, lb = set(leaf_names(a)), set(leaf_names(b))
    if la != lb:
        raise ValueError(f"trees have different leaf sets ({len(la ^ lb)} names differ)")
    sa, sb = _internal(splits(a), len(la)), _internal(splits(b), len(la))
    rf = len(sa ^ sb)
    total = len(sa) + len(sb)
    return rf, (rf / total if total else 0.0)


def branch_score(a: Node, b: Node) -> float:
    """Kuhner-Felsenstein distance: sqrt of the summed squared branch-length differences over all
    splits (a split missing from one tree counts with length 0)."""
    la, lb = set(leaf_names(a)), set(leaf_names(b))
    if la != lb:
        raise ValueError("trees have different leaf sets")
    sa, sb = splits(a), splits(b)
    return math.sqrt(sum((sa.get(k, 0.0) - sb.get(k, 0.0)) ** 2 for k in set(sa) | set(sb)))


def patristic(root: Node) -> Tuple[List[str], "object"]:
    """(leaf names, path-length matrix) — what the reference's training labels are
    (dendropy patristic distances, /root/reference/phyloformer/data.py:45-52)."""
    import numpy as np
    names = leaf_names(root)
    index = {n: i for i, n in enumerate(names)}
    dm = np.zeros((len(names), len(names)))

    def visit(n: Node) -> Dict[int, float]:
        if n.is_leaf():
            return {index[n.name]: 0.0}
        merged: Dict[int, float] = {}
        for c in n.children:
            sub = {k: v + (c.length or 0.0) for k, v in visit(c).items()}
            for i, di in merged.items():
                for j, dj in sub.items():
                    dm[i, j] = dm[j, i] = di + dj
            merged.update(sub)
        return merged

    visit(root)
    return names, dm
