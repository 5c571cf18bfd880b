"""ctypes bindings of the native file-format helpers (``csrc/pf_hostio.cpp``).

Same results and the same exception types as the pure-Python mirrors of the
reference in :mod:`fasta` (``load_alignment``, /root/reference/phyloformer/data.py:11-31)
and :mod:`phylip` (``vec_to_phylip``, /root/reference/infer_alns.py:14-25), ~10x
faster and with the GIL released, so the CLI's loader and writer threads run
beside the GPU instead of in front of it.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from .engine import load_library

PF_FASTA_EBYTE, PF_FASTA_ERAGGED, PF_FASTA_ENOHEADER, PF_FASTA_EEMPTY, PF_FASTA_ECAP = -16, -17, -18, -19, -20


def parse_fasta(data: bytes) -> Tuple[np.ndarray, List[str]]:
    """FASTA bytes → ``(uint8[N, L] residue indices, ids)``."""
    lib = load_library()
    n_max = data.count(b">") + 1
    idx = np.empty(len(data), dtype=np.uint8)
    spans = np.empty(2 * n_max, dtype=np.int64)
    n, l, detail = C.c_int32(0), C.c_int32(0), C.c_int64(0)
    rc = lib.pf_parse_fasta(data, len(data), idx.ctypes.data, idx.size, spans.ctypes.data, n_max,
                            C.byref(n), C.byref(l), C.byref(detail))
    if rc == PF_FASTA_EBYTE:
        raise KeyError(int(detail.value))
    if rc == PF_FASTA_ENOHEADER:
        raise IndexError("sequence data before the first '>' header")
    if rc == PF_FASTA_EEMPTY or (rc == 0 and l.value == 0):
        # same class as the reference, whose one_hot refuses the empty tensor (data.py:28)
        raise RuntimeError("no residues found (empty alignment)")
    if rc == PF_FASTA_ERAGGED:
        raise ValueError("expected sequences of equal length")
    if rc != 0:
        raise RuntimeError(f"pf_parse_fasta failed with status {rc}")
    ids = [data[int(spans[2 * i]):int(spans[2 * i] + spans[2 * i + 1])].decode("utf8") for i in range(n.value)]
    return idx[:n.value * l.value].reshape(n.value, l.value).copy(), ids


def load_alignment(filepath) -> Tuple[np.ndarray, List[str]]:
    with open(filepath, "rb") as fh:
        return parse_fasta(fh.read())


def format_phylip(preds: np.ndarray, ids: Sequence[str]) -> bytes:
    """Distance vector ``[P]`` + ids → PHYLIP text (utf-8 bytes), byte-identical to ``vec_to_phylip``."""
    lib = load_library()
    n = len(ids)
    p = np.ascontiguousarray(np.asarray(preds, dtype=np.float32).reshape(-1))
    if p.size != n * (n - 1) // 2:
        raise ValueError(f"expected {n * (n - 1) // 2} distances for {n} sequences, got {p.shape}")
    enc = [s.encode("utf8") for s in ids]
    if any(b"\0" in e for e in enc):
        raise ValueError("sequence id contains a NUL byte")
    arr = (C.c_char_p * n)(*enc)
    cap = sum(len(e) for e in enc) + n * (n * 24 + 2) + 32
    buf = C.create_string_buffer(cap)
    w = lib.pf_format_phylip(p.ctypes.data, n, arr, buf, cap)
    if w < 0:
        raise RuntimeError(f"pf_format_phylip failed with status {w}")
    if w > cap:                                      # astronomically large distances
        cap = int(w)
        buf = C.create_string_buffer(cap)
        w = lib.pf_format_phylip(p.ctypes.data, n, arr, buf, cap)
    return buf.raw[:w]
