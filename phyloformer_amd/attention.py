"""Host-side mirror of the reference's softmax ``MultiHeadAttention`` module.

Same constructor arguments and call surface as
/root/reference/phyloformer/attention.py:53-91 (``MultiHeadAttention(nb_heads, embed_dim)``,
``forward(input[B, R, C, E]) -> [B, R, C, E]``, attention along axis 2); the arithmetic runs in
``libphyloformer_amd.so`` (kernels ``k_mha_qkv / k_mha_attn / k_mha_out``, csrc/pf_mha.hip.h) on one
MI355X.  The class is dead code in the reference — nothing instantiates it and no
shipped checkpoint fits it — so it is not used by ``Phyloformer``; this is the §8f rank-4 row.
There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np

from .engine import PF_EINVAL, PF_OK, EngineError, load_library

_PARAMS = ("q_proj", "k_proj", "v_proj", "out_proj")


class _MhaWeights(C.Structure):
    _fields_ = [("n_heads", C.c_int32), ("embed_dim", C.c_int32)] + \
               [(n, C.POINTER(C.c_float)) for n in ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo")]


class MultiHeadAttention:
    """``MultiHeadAttention(nb_heads=4, embed_dim=64)`` + ``load_state_dict`` + ``__call__``."""

    def __init__(self, nb_heads: int = 4, embed_dim: int = 64, dropout: float = 0.0, device: int = 0,
                 engine=None):
        if embed_dim % nb_heads != 0:
            # attention.py:27-31
            raise ValueError("Embed dim and QK dim (if specified) mus tbe divisible by the number of heads.\n"
                             f"Embed: {embed_dim}, QK: {embed_dim} -> n_heads: {nb_heads}")
        self.nb_heads, self.embed_dim, self.head_dim = nb_heads, embed_dim, embed_dim // nb_heads
        self._lib = load_library()
        self._own_handle = engine is None
        self._h = C.c_void_p()
        if engine is None:
            rc = self._lib.pf_create_bare(device, C.byref(self._h))
            if rc != PF_OK:
                raise EngineError(rc, (self._lib.pf_last_error(None) or b"").decode())
        else:
            self._h = engine._h
            self._engine = engine          # keep the parent alive
        self._m = C.c_void_p()
        self._keep = None

    def _err(self, rc):
        msg = (self._lib.pf_last_error(self._h) or b"").decode()
        if rc == PF_EINVAL:
            raise ValueError(msg)
        raise EngineError(rc, msg)

    def load_state_dict(self, sd: Dict[str, np.ndarray]):
        arrs = {}
        for p in _PARAMS:
            for kind, shape in (("weight", (self.embed_dim, self.embed_dim)), ("bias", (self.embed_dim,))):
                v = sd[f"{p}.{kind}"]
                v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
                if v.shape != shape:
                    raise ValueError(f"{p}.{kind}: expected shape {shape}, got {v.shape}")
                arrs[f"{p}.{kind}"] = np.ascontiguousarray(v, dtype=np.float32)
        ptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        w = _MhaWeights(self.nb_heads, self.embed_dim,
                        ptr(arrs["q_proj.weight"]), ptr(arrs["q_proj.bias"]),
                        ptr(arrs["k_proj.weight"]), ptr(arrs["k_proj.bias"]),
                        ptr(arrs["v_proj.weight"]), ptr(arrs["v_proj.bias"]),
                        ptr(arrs["out_proj.weight"]), ptr(arrs["out_proj.bias"]))
        if self._m:
            self._lib.pf_mha_destroy(self._m)
            self._m = C.c_void_p()
        rc = self._lib.pf_mha_create(self._h, C.byref(w), C.byref(self._m))
        if rc != PF_OK:
            self._err(rc)
        return self

    def eval(self):
        return self

    def forward(self, x) -> np.ndarray:
        if not self._m:
            raise RuntimeError("no weights loaded")
        a = x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
        if a.ndim != 4 or a.shape[-1] != self.embed_dim:
            raise ValueError(f"expected input [B, R, C, {self.embed_dim}], got {a.shape}")
        a = np.ascontiguousarray(a, dtype=np.float32)
        y = np.empty_like(a)
        B, R, Cc, _E = a.shape
        rc = self._lib.pf_mha_forward(self._m, a.ctypes.data, B, R, Cc, y.ctypes.data)
        if rc != PF_OK:
            self._err(rc)
        return y

    __call__ = forward

    def forward_device(self, d_x: int, B: int, R: int, Cc: int, d_y: int):
        rc = self._lib.pf_mha_forward_device(self._m, C.c_void_p(d_x), B, R, Cc, C.c_void_p(d_y))
        if rc != PF_OK:
            self._err(rc)

    def close(self):
        if getattr(self, "_m", None):
            self._lib.pf_mha_destroy(self._m)
            self._m = C.c_void_p()
        if getattr(self, "_own_handle", False) and getattr(self, "_h", None):
            self._lib.pf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
