"""FASTA → residue-index matrix.

Host-side mirror of ``load_alignment`` (/root/reference/phyloformer/data.py:7-31).
The reference builds a one-hot ``int64[22, L, N]`` tensor; the embedding that
consumes it is a 1×1 convolution, i.e. a 22-row table lookup, so the device
path only ever needs the indices: ``uint8[N, L]`` with values ``0..21`` in the
alphabet order of data.py:7.

Parsing rules kept from the reference (data.py:18-26): the file is read as
bytes, every line is ``strip()``-ed, a line starting with ``>`` opens a record
whose id is the rest of that line, all other lines are appended to the current
record, a byte outside the alphabet raises ``KeyError``, ragged records raise
``ValueError``, residues before the first header ``IndexError`` and a file
without a single residue ``RuntimeError`` (the reference's ``one_hot`` refuses
the empty float tensor ``torch.tensor([])``, data.py:28).  Pinned against the
reference itself by ``tests/golden/fasta_edge.json`` (``oracle/gen_golden_fasta.py``).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

ALPHABET = b"ARNDCQEGHILKMFPSTWYVX-"  # data.py:7
N_ALPHABET = len(ALPHABET)
GAP_INDEX = ALPHABET.index(b"-")
UNKNOWN_INDEX = ALPHABET.index(b"X")

_LUT = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(ALPHABET):
    _LUT[_c] = _i


def encode_sequence(seq: bytes) -> np.ndarray:
    """Map residue bytes to alphabet indices; ``KeyError`` on an unknown byte (data.py:26)."""
    raw = np.frombuffer(seq, dtype=np.uint8)
    out = _LUT[raw]
    bad = np.flatnonzero(out == 255)
    if bad.size:
        raise KeyError(int(raw[bad[0]]))
    return out


def parse_fasta(data: bytes) -> Tuple[np.ndarray, List[str]]:
    """Parse FASTA bytes into ``(uint8[N, L], ids)``."""
    ids: List[str] = []
    chunks: List[List[np.ndarray]] = []
    for line in data.split(b"\n"):  # binary file iteration splits on \n only
        line = line.strip()
        if line.startswith(b">"):
            ids.append(line[1:].decode("utf8"))
            chunks.append([])
        elif line:
            if not chunks:
                # the reference indexes sequences[-1] on an empty list here
                raise IndexError("sequence data before the first '>' header")
            chunks[-1].append(encode_sequence(line))
    seqs = [np.concatenate(c) if c else np.zeros(0, np.uint8) for c in chunks]
    lengths = {s.size for s in seqs}
    if len(lengths) > 1:
        raise ValueError(
            f"expected sequences of equal length, got lengths {sorted(lengths)}")
    if not chunks or lengths == {0}:
        # the reference fails inside one_hot here (RuntimeError): torch.tensor([]) is a float tensor
        raise RuntimeError("no residues found (empty alignment)")
    return np.stack(seqs).astype(np.uint8), ids


def load_alignment(filepath) -> Tuple[np.ndarray, List[str]]:
    """Read a FASTA alignment → ``(uint8[N, L] indices, ids)``."""
    with open(filepath, "rb") as fh:
        return parse_fasta(fh.read())


def one_hot(indices: np.ndarray) -> np.ndarray:
    """Indices ``[N, L]`` → the reference's one-hot layout ``int64[22, L, N]`` (data.py:28-29)."""
    idx = np.asarray(indices)
    oh = np.zeros((N_ALPHABET,) + idx.shape[::-1], dtype=np.int64)
    n, l = idx.shape
    oh[idx.T, np.arange(l)[:, None], np.arange(n)[None, :]] = 1
    return oh


def from_one_hot(x: np.ndarray) -> np.ndarray:
    """One-hot ``[..., 22, L, N]`` (any dtype) → indices ``uint8[..., N, L]``."""
    x = np.asarray(x)
    if x.shape[-3] != N_ALPHABET:
        raise ValueError(f"expected {N_ALPHABET} channels on axis -3, got shape {x.shape}")
    idx = x.argmax(axis=-3).astype(np.uint8)  # [..., L, N]
    return np.ascontiguousarray(np.swapaxes(idx, -1, -2))
