"""ctypes binding of ``libphyloformer_amd.so`` (C ABI in ``include/phyloformer_amd.h``).

This is the only way the Python host code reaches the device: plain pointers
and sizes, no torch op dispatch.  If the shared library is missing or no
gfx950 device is present the calls raise — there is deliberately no CPU
fallback on the product path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import numpy as np

from .weights import ModelWeights

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libphyloformer_amd.so")
ABI_VERSION = 5            # PF_ABI_VERSION of include/phyloformer_amd.h this binding was written against
UNIQUE_ID_BYTES = 256      # PF_UNIQUE_ID_BYTES: two ncclUniqueIds, one per communicator / stream

PF_OK, PF_EINVAL, PF_EHIP, PF_ERCCL, PF_ENOMEM, PF_ESTATE = 0, -1, -2, -3, -4, -5


def _refuse_single_sequence(B: int, N: int):
    """One sequence = no pair: the reference's forward raises RuntimeError there (attention.py:193 cannot view the empty
    tensor; pinned by tests/golden/cli_bad_entry.json), the C ABI answers PF_EINVAL for any N < 2 - the host mirror
    raises what the reference raises."""
    if N == 1 and B >= 1:
        raise RuntimeError(f"cannot reshape tensor of 0 elements into shape [{B}, -1, 0, 64] because the unspecified "
                           "dimension size -1 can be any value and is ambiguous")


class EngineError(RuntimeError):
    """A HIP / RCCL / allocation failure reported by the native library."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"[pf status {code}] {msg}")
        self.code = code


class pf_weights_t(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("n_heads", C.c_int32), ("embed_dim", C.c_int32),
                ("n_alphabet", C.c_int32), ("blob", C.POINTER(C.c_float)), ("blob_len", C.c_uint64)]


# name -> (restype, argtypes); every symbol declared in include/phyloformer_amd.h
_H = C.c_void_p
SIGNATURES = {
    "pf_abi_version": (C.c_int, []),
    "pf_build_info": (C.c_char_p, []),
    "pf_blob_len": (C.c_uint64, [C.c_int32, C.c_int32, C.c_int32]),
    "pf_create": (C.c_int, [C.POINTER(pf_weights_t), C.c_int, C.POINTER(_H)]),
    "pf_destroy": (C.c_int, [_H]),
    "pf_last_error": (C.c_char_p, [_H]),
    "pf_set_option": (C.c_int, [_H, C.c_char_p, C.c_int64]),
    "pf_forward": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "pf_forward_device": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "pf_forward_sharded": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_void_p]),
    "pf_forward_sharded_device": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_int32, C.c_void_p]),
    "pf_comm_unique_id": (C.c_int, [C.c_void_p]),
    "pf_comm_init": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_int32]),
    "pf_comm_destroy": (C.c_int, [_H]),
    "pf_comm_info": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int32)]),
    "pf_synchronize": (C.c_int, [_H]),
    "pf_get_stream": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "pf_device_malloc": (C.c_int, [_H, C.c_size_t, C.POINTER(C.c_void_p)]),
    "pf_device_free": (C.c_int, [_H, C.c_void_p]),
    "pf_memcpy_h2d": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_size_t]),
    "pf_memcpy_d2h": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_size_t]),
    "pf_profile_reset": (C.c_int, [_H]),
    "pf_profile_get": (C.c_int, [_H, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "pf_debug_read": (C.c_int64, [_H, C.c_char_p, C.c_void_p, C.c_int64]),
    "pf_device_info": (C.c_int, [_H, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32),
                                 C.POINTER(C.c_uint64)]),
    "pf_device_pci": (C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pf_selftest": (C.c_int, [_H, C.c_void_p]),
    "pf_create_bare": (C.c_int, [C.c_int, C.POINTER(_H)]),
    "pf_mha_create": (C.c_int, [_H, C.c_void_p, C.POINTER(C.c_void_p)]),
    "pf_mha_destroy": (C.c_int, [C.c_void_p]),
    "pf_mha_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "pf_mha_forward_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "pf_parse_fasta": (C.c_int, [C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32,
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "pf_format_phylip": (C.c_int64, [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.c_char_p, C.c_int64]),
    "pf_format_phylip_n": (C.c_int64, [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.c_void_p, C.c_char_p, C.c_int64]),
    "pf_fasta_batch_load": (C.c_int, [C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "pf_fasta_batch_free": (None, [C.c_void_p]),
    "pf_fasta_batch_count": (C.c_int32, [C.c_void_p]),
    "pf_fasta_batch_infos": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_fasta_batch_id": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "pf_fasta_batch_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "pf_nj_newick_n": (C.c_int64, [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.c_void_p, C.c_int32, C.c_char_p, C.c_int64]),
    "pf_phylip_write_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                        C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int32, C.c_void_p]),
    "pf_forward_shards_emulated": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                             C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen the native library and attach the prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("PHYLOFORMER_AMD_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise EngineError(PF_EHIP, f"native library not found at {p}; run "
                          "`python -m phyloformer_amd.build` (there is no CPU fallback)")
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.pf_abi_version() != ABI_VERSION:
        raise EngineError(PF_ESTATE, f"{p} has ABI version {lib.pf_abi_version()}, this binding expects "
                          f"{ABI_VERSION}; rebuild with `python -m phyloformer_amd.build --force`")
    if path is None:
        _lib = lib
    return lib


def build_info(path: Optional[str] = None) -> Dict[str, object]:
    """What the loaded library was built from (``pf_build_info``, ABI 4): compiler, flags, the scheduling
    strategy that really compiled the kernels, source and kernel hashes.  Needs no device."""
    import json
    return json.loads(load_library(path).pf_build_info().decode())


def _u8(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


class Engine:
    """One handle = one GPU + one stream + one set of prepared weights."""

    def __init__(self, weights: ModelWeights, device: int = 0):
        self._lib = load_library()
        self._h = _H()
        blob = weights.blob()
        expect = self._lib.pf_blob_len(weights.n_blocks, weights.n_heads, weights.embed_dim)
        if blob.size != expect:
            raise ValueError(f"weight blob has {blob.size} floats, library expects {expect}")
        w = pf_weights_t(weights.n_blocks, weights.n_heads, weights.embed_dim, 22,
                         blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size)
        rc = self._lib.pf_create(C.byref(w), device, C.byref(self._h))
        if rc != PF_OK:
            msg = (self._lib.pf_last_error(None) or b"").decode()
            self._h = _H()
            if rc == PF_EINVAL:
                raise ValueError(msg)
            raise EngineError(rc, msg)
        self.device = device
        self.world = 1
        self.rank = 0

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc >= 0:
            return rc
        msg = (self._lib.pf_last_error(self._h) or b"").decode()
        if rc == PF_EINVAL:
            raise ValueError(msg)   # the reference raises ValueError for N > 200 (model.py:24-28)
        raise EngineError(rc, msg)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.pf_destroy(self._h)
            self._h = _H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, key: str, value: int):
        self._check(self._lib.pf_set_option(self._h, key.encode(), int(value)))

    def device_info(self) -> Dict[str, object]:
        name = C.create_string_buffer(256)
        cu = C.c_int32()
        mem = C.c_uint64()
        self._check(self._lib.pf_device_info(self._h, name, 256, C.byref(cu), C.byref(mem)))
        return {"name": name.value.decode(), "cu_count": cu.value, "hbm_bytes": mem.value}

    def build_info(self) -> Dict[str, object]:
        """What the library behind this engine was built from (``pf_build_info``)."""
        import json
        return json.loads(self._lib.pf_build_info().decode())

    def device_pci(self) -> Tuple[int, int, int]:
        """(domain, bus, device) of this engine's GPU - the key rocm_smi finds it by (phyloformer_amd/smi.py)."""
        d, b, v = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._lib.pf_device_pci(self._h, C.byref(d), C.byref(b), C.byref(v)))
        return d.value, b.value, v.value

    # -- forward ----------------------------------------------------------------------------
    def forward(self, idx: np.ndarray) -> np.ndarray:
        """``uint8[B, N, L]`` (or ``[N, L]``) → ``float32[B, P]`` (or ``[P]``)."""
        idx = _u8(idx)
        single = idx.ndim == 2
        if single:
            idx = idx[None]
        if idx.ndim != 3:
            raise ValueError(f"idx must be [B, N, L] or [N, L], got shape {idx.shape}")
        B, N, L = idx.shape
        _refuse_single_sequence(B, N)
        out = np.empty((B, N * (N - 1) // 2), dtype=np.float32)
        self._check(self._lib.pf_forward(self._h, idx.ctypes.data, B, N, L, out.ctypes.data))
        return out[0] if single else out

    def forward_sharded(self, idx_local: np.ndarray, l_begin: int, l_end: int, L_total: int) -> np.ndarray:
        """This rank's sites ``[l_begin, l_end)`` of ``uint8[B, N, L_total]`` alignments."""
        idx = _u8(idx_local)
        single = idx.ndim == 2
        if single:
            idx = idx[None]
        B, N, Ll = idx.shape
        _refuse_single_sequence(B, N)
        if Ll != l_end - l_begin:
            raise ValueError(f"idx has {Ll} sites, expected {l_end - l_begin}")
        out = np.empty((B, N * (N - 1) // 2), dtype=np.float32)
        self._check(self._lib.pf_forward_sharded(self._h, idx.ctypes.data, B, N, l_begin, l_end,
                                                 L_total, out.ctypes.data))
        return out[0] if single else out

    def forward_shards_emulated(self, idx: np.ndarray, nshards: int) -> np.ndarray:
        """Site-sharded algorithm over ``nshards`` emulated ranks on this one GPU (tests)."""
        idx = _u8(idx)
        single = idx.ndim == 2
        if single:
            idx = idx[None]
        B, N, L = idx.shape
        out = np.empty((B, N * (N - 1) // 2), dtype=np.float32)
        self._check(self._lib.pf_forward_shards_emulated(self._h, idx.ctypes.data, B, N, L, nshards,
                                                         out.ctypes.data))
        return out[0] if single else out

    # -- device-resident variant (benchmark) ------------------------------------------------
    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self._lib.pf_device_malloc(self._h, nbytes, C.byref(p)))
        return p.value

    def free(self, ptr: int):
        self._check(self._lib.pf_device_free(self._h, C.c_void_p(ptr)))

    def h2d(self, dst: int, src: np.ndarray):
        src = np.ascontiguousarray(src)
        self._check(self._lib.pf_memcpy_h2d(self._h, C.c_void_p(dst), src.ctypes.data, src.nbytes))

    def d2h(self, dst: np.ndarray, src: int):
        assert dst.flags["C_CONTIGUOUS"]
        self._check(self._lib.pf_memcpy_d2h(self._h, dst.ctypes.data, C.c_void_p(src), dst.nbytes))

    def forward_device(self, d_idx: int, B: int, N: int, L: int, d_out: int):
        self._check(self._lib.pf_forward_device(self._h, C.c_void_p(d_idx), B, N, L, C.c_void_p(d_out)))

    def forward_sharded_device(self, d_idx: int, B: int, N: int, l_begin: int, l_end: int,
                               L_total: int, d_out: int):
        self._check(self._lib.pf_forward_sharded_device(
            self._h, C.c_void_p(d_idx), B, N, l_begin, l_end, L_total, C.c_void_p(d_out)))

    def synchronize(self):
        self._check(self._lib.pf_synchronize(self._h))

    # -- RCCL -------------------------------------------------------------------------------
    def unique_id(self) -> bytes:
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        rc = self._lib.pf_comm_unique_id(buf)
        if rc != PF_OK:
            raise EngineError(rc, (self._lib.pf_last_error(None) or b"").decode())
        return buf.raw

    def comm_init(self, unique_id: Optional[bytes], rank: int, world: int):
        buf = C.create_string_buffer(unique_id, UNIQUE_ID_BYTES) if unique_id else None
        self._check(self._lib.pf_comm_init(self._h, buf, rank, world))
        self.rank, self.world = rank, world

    def comm_destroy(self):
        self._check(self._lib.pf_comm_destroy(self._h))
        self.rank, self.world = 0, 1

    def collective_count(self) -> int:
        """All-reduces issued by this handle since the last ``profile_reset`` (always counted)."""
        return self.profile_get("collectives")[0]

    def rechecked_count(self) -> int:
        """Alignments the range re-check (option "recheck_above") recomputed in float64 since ``profile_reset``."""
        return self.profile_get("rechecked")[0]

    def comm_info(self) -> dict:
        """Path and version of the librccl the native library resolved (loads it if necessary)."""
        buf = C.create_string_buffer(512)
        ver = C.c_int32()
        rc = self._lib.pf_comm_info(buf, 512, C.byref(ver))
        if rc != PF_OK:
            raise EngineError(rc, (self._lib.pf_last_error(None) or b"").decode())
        return {"library": buf.value.decode(), "version": int(ver.value)}

    # -- profiling / debugging --------------------------------------------------------------
    def profile_reset(self):
        self._check(self._lib.pf_profile_reset(self._h))

    def profile_get(self, kernel: str) -> Tuple[int, float]:
        n = C.c_int64()
        ms = C.c_double()
        self._check(self._lib.pf_profile_get(self._h, kernel.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def debug_read(self, name: str) -> np.ndarray:
        n = self._check(self._lib.pf_debug_read(self._h, name.encode(), None, 0))
        out = np.empty(n, dtype=np.float32)
        self._check(self._lib.pf_debug_read(self._h, name.encode(), out.ctypes.data, n))
        return out

    def selftest(self) -> np.ndarray:
        out = np.empty(2304, dtype=np.float32)
        self._check(self._lib.pf_selftest(self._h, out.ctypes.data))
        return out
