"""ctypes bindings of the native file-format helpers (``csrc/pf_hostio.cpp``).

Same results and the same exception types as the pure-Python mirrors of the
reference in :mod:`fasta` (``load_alignment``, /root/reference/phyloformer/data.py:11-31)
and :mod:`phylip` (``vec_to_phylip``, /root/reference/infer_alns.py:14-25), ~10x
faster and with the GIL released, so the CLI's loader and writer threads run
beside the GPU instead of in front of it.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence, Tuple

import numpy as np

from .engine import load_library

PF_FASTA_EBYTE, PF_FASTA_ERAGGED, PF_FASTA_ENOHEADER, PF_FASTA_EEMPTY, PF_FASTA_ECAP, PF_FASTA_EUTF8 = -16, -17, -18, -19, -20, -21


PF_EIO = -6


def parse_error(rc: int, l: int, detail: int, path: str = "", data: bytes = b"") -> "BaseException | None":
    """The exception the reference raises where pf_parse_fasta returns ``rc`` (None = a valid alignment)."""
    if rc == PF_FASTA_EUTF8:
        # the header at offset `detail` is not UTF-8: let bytes.decode build the reference's exception (data.py:22)
        if not data and path:
            with open(path, "rb") as fh:
                data = fh.read()
        line = data[int(detail):].split(b"\n", 1)[0].strip()
        try:
            line.decode("utf8")
        except UnicodeDecodeError as exc:
            return exc
        return UnicodeDecodeError("utf-8", line, 0, 1, "invalid header")
    if rc == PF_FASTA_EBYTE:
        return KeyError(int(detail))
    if rc == PF_FASTA_ENOHEADER:
        return IndexError("sequence data before the first '>' header")
    if rc == PF_FASTA_EEMPTY or (rc == 0 and l == 0):
        # same class as the reference, whose one_hot refuses the empty tensor (data.py:28)
        return RuntimeError("no residues found (empty alignment)")
    if rc == PF_FASTA_ERAGGED:
        return ValueError("expected sequences of equal length")
    if rc == PF_EIO:
        import os
        return OSError(int(detail), os.strerror(int(detail)), path)
    if rc != 0:
        return RuntimeError(f"pf_parse_fasta failed with status {rc}")
    return None


def parse_fasta(data: bytes) -> Tuple[np.ndarray, List[str]]:
    """FASTA bytes → ``(uint8[N, L] residue indices, ids)``."""
    lib = load_library()
    n_max = data.count(b">") + 1
    idx = np.empty(len(data), dtype=np.uint8)
    spans = np.empty(2 * n_max, dtype=np.int64)
    n, l, detail = C.c_int32(0), C.c_int32(0), C.c_int64(0)
    rc = lib.pf_parse_fasta(data, len(data), idx.ctypes.data, idx.size, spans.ctypes.data, n_max,
                            C.byref(n), C.byref(l), C.byref(detail))
    exc = parse_error(rc, l.value, detail.value, data=data)
    if exc is not None:
        raise exc
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
    arr = (C.c_char_p * n)(*enc)
    lens = np.array([len(e) for e in enc], dtype=np.int64)      # explicit lengths: an id may hold NUL bytes
    cap = int(lens.sum()) + n * (n * 24 + 2) + 32
    buf = C.create_string_buffer(cap)
    w = lib.pf_format_phylip_n(p.ctypes.data, n, arr, lens.ctypes.data, buf, cap)
    if w < 0:
        raise RuntimeError(f"pf_format_phylip failed with status {w}")
    if w > cap:                                      # astronomically large distances
        cap = int(w)
        buf = C.create_string_buffer(cap)
        w = lib.pf_format_phylip_n(p.ctypes.data, n, arr, lens.ctypes.data, buf, cap)
    return buf.raw[:w]


def nj_newick(preds: np.ndarray, ids: Sequence[str], clamp_negative: bool = True) -> bytes:
    """Distance vector ``[P]`` + ids → Newick text of the neighbour-joining tree (utf-8 bytes), byte-identical to
    ``nj.neighbor_joining`` on the matrix ``vec_to_phylip`` builds (the CLI's ``--trees``, infer_alns.py:120-123)."""
    lib = load_library()
    n = len(ids)
    p = np.ascontiguousarray(np.asarray(preds, dtype=np.float32).reshape(-1))
    if p.size != n * (n - 1) // 2:
        raise ValueError(f"expected {n * (n - 1) // 2} distances for {n} sequences, got {p.shape}")
    enc = [s.encode("utf8") for s in ids]
    arr = (C.c_char_p * n)(*enc)
    lens = np.array([len(e) for e in enc], dtype=np.int64)
    need = lib.pf_nj_newick_n(p.ctypes.data, n, arr, lens.ctypes.data, int(clamp_negative), None, 0)
    if need < 0:
        raise RuntimeError(f"pf_nj_newick_n failed with status {need}")
    buf = C.create_string_buffer(int(need) + 1)
    w = lib.pf_nj_newick_n(p.ctypes.data, n, arr, lens.ctypes.data, int(clamp_negative), buf, need)
    return buf.raw[:w]


class FastaBatch:
    """``pf_fasta_batch_load``: a list of FASTA files read and parsed on native threads (GIL released for the
    whole call).  The parsed alignments stay in library memory; ``gather`` copies the residue indices of a
    same-shaped selection into one ``uint8[B, N, L]`` array and ``write_phylip`` formats and writes their
    distance matrices with the ids the batch holds - Python never touches residues or ids on that path."""

    def __init__(self, paths: Sequence[str], threads: int = 8):
        self._lib = load_library()
        self.paths = list(paths)
        self._h = C.c_void_p()
        enc = [os.fsencode(p) for p in self.paths]
        arr = (C.c_char_p * len(enc))(*enc)
        rc = self._lib.pf_fasta_batch_load(arr, len(enc), int(threads), C.byref(self._h))
        if rc != 0:
            raise MemoryError("pf_fasta_batch_load failed") if rc == -4 else RuntimeError(f"pf_fasta_batch_load: status {rc}")
        k = len(enc)
        self.status = np.empty(k, np.int32)
        self.n = np.empty(k, np.int32)
        self.l = np.empty(k, np.int32)
        self.detail = np.empty(k, np.int64)
        self._lib.pf_fasta_batch_infos(self._h, self.status.ctypes.data, self.n.ctypes.data, self.l.ctypes.data,
                                       self.detail.ctypes.data)

    def __len__(self):
        return len(self.paths)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pf_fasta_batch_free(self._h)
            self._h = None

    __del__ = close

    def error(self, i: int) -> "BaseException | None":
        """What ``load_alignment`` raises for file ``i`` (None: a valid alignment)."""
        return parse_error(int(self.status[i]), int(self.l[i]), int(self.detail[i]), self.paths[i])

    def ids(self, i: int) -> List[str]:
        out, p, ln = [], C.c_void_p(), C.c_int64()
        for s in range(int(self.n[i])):
            if self._lib.pf_fasta_batch_id(self._h, i, s, C.byref(p), C.byref(ln)) != 0:
                raise IndexError((i, s))
            out.append(C.string_at(p, ln.value).decode("utf8"))
        return out

    def indices(self, i: int) -> np.ndarray:
        return gather([(self, i)], int(self.n[i]), int(self.l[i]))[0]


def gather(entries: Sequence[Tuple["FastaBatch", int]], n: int, l: int) -> np.ndarray:
    """``uint8[B, N, L]`` of the ``(batch, file)`` entries, all of shape ``(n, l)``."""
    lib = load_library()
    k = len(entries)
    hs = (C.c_void_p * k)(*[e[0]._h for e in entries])
    fi = np.array([e[1] for e in entries], dtype=np.int32)
    out = np.empty((k, n, l), dtype=np.uint8)
    rc = lib.pf_fasta_batch_gather(hs, fi.ctypes.data, k, n, l, out.ctypes.data)
    if rc != 0:
        raise ValueError(f"pf_fasta_batch_gather: status {rc} (entries of another shape than {n} x {l}?)")
    return out


def write_phylip(entries: Sequence[Tuple["FastaBatch", int]], n: int, preds: np.ndarray, out_paths: Sequence[str],
                 threads: int = 8, tree_paths: "Sequence[str] | None" = None) -> None:
    """Format ``preds[B, P]`` and write ``out_paths`` - and, with ``tree_paths``, the neighbour-joining trees of the same
    distances - on native threads; ``OSError`` for the first file that failed."""
    lib = load_library()
    k = len(entries)
    p = np.ascontiguousarray(np.asarray(preds, dtype=np.float32).reshape(k, -1))
    if p.shape[1] != n * (n - 1) // 2 or len(out_paths) != k:
        raise ValueError(f"expected {k} x {n * (n - 1) // 2} distances and {k} paths")
    hs = (C.c_void_p * k)(*[e[0]._h for e in entries])
    fi = np.array([e[1] for e in entries], dtype=np.int32)
    paths = (C.c_char_p * k)(*[os.fsencode(q) for q in out_paths])
    trees = None
    if tree_paths is not None:
        if len(tree_paths) != k:
            raise ValueError(f"expected {k} tree paths")
        trees = (C.c_char_p * k)(*[os.fsencode(q) for q in tree_paths])
    status = np.zeros(k, dtype=np.int32)
    rc = lib.pf_phylip_write_batch(hs, fi.ctypes.data, k, n, p.ctypes.data, paths, trees, int(threads), status.ctypes.data)
    if rc != 0:
        raise ValueError(f"pf_phylip_write_batch: status {rc}")
    bad = np.flatnonzero(status)
    if bad.size:
        e = -int(status[bad[0]])
        raise OSError(e, os.strerror(e), out_paths[int(bad[0])] if tree_paths is None else f"{out_paths[int(bad[0])]} / {tree_paths[int(bad[0])]}")
