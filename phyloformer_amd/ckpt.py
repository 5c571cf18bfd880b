"""Torch-free reader for Phyloformer ``.ckpt`` files.

Drop-in surface for the checkpoint handling of the reference CLI
(/root/reference/infer_alns.py:71-82): ``torch.load`` → ``state_dict`` →
strip the ``model.`` prefix → drop the stale ``model.seq2pair`` entry.

A ``.ckpt`` is a zip archive (stored, not deflated) holding
``<name>/data.pkl`` (protocol-2 pickle), ``<name>/data/<key>`` raw
little-endian storages, ``<name>/byteorder`` and ``<name>/version``.  The
pickle only needs three globals (``collections.OrderedDict``,
``torch._utils._rebuild_tensor_v2`` and ``torch.<X>Storage``); everything
else is refused, so loading an untrusted checkpoint cannot run code.
"""
from __future__ import annotations

import collections
import pickle
import zipfile
from typing import Dict, Tuple

import numpy as np

__all__ = ["load_ckpt", "load_state_dict", "CheckpointError"]


class CheckpointError(ValueError):
    """Raised when a checkpoint cannot be decoded or has unexpected shapes."""


_STORAGE_DTYPES = {
    "FloatStorage": np.dtype("<f4"),
    "DoubleStorage": np.dtype("<f8"),
    "HalfStorage": np.dtype("<f2"),
    "LongStorage": np.dtype("<i8"),
    "IntStorage": np.dtype("<i4"),
    "ShortStorage": np.dtype("<i2"),
    "CharStorage": np.dtype("i1"),
    "ByteStorage": np.dtype("u1"),
    "BoolStorage": np.dtype("?"),
}


class _StorageType:
    def __init__(self, name: str):
        self.name = name
        self.dtype = _STORAGE_DTYPES[name]


class _Opaque:
    """Placeholder for objects we do not need (Lightning callbacks, optimizer state …)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        pass

    def __call__(self, *a, **k):
        return _Opaque()


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False,
                       backward_hooks=None, metadata=None):
    arr = storage
    size = tuple(int(s) for s in size)
    stride = tuple(int(s) for s in stride)
    storage_offset = int(storage_offset)
    # as_strided does no bounds checking: a crafted pickle must not be able to read outside its storage
    if (not isinstance(arr, np.ndarray) or arr.ndim != 1 or len(size) != len(stride) or storage_offset < 0
            or any(n < 0 for n in size) or any(st < 0 for st in stride)):
        raise CheckpointError(f"malformed tensor record (offset {storage_offset}, size {size}, stride {stride})")
    last = storage_offset + sum((n - 1) * st for n, st in zip(size, stride))
    if all(n > 0 for n in size) and last >= arr.size:
        raise CheckpointError(f"tensor record reaches element {last} of a storage with {arr.size} elements")
    if any(n == 0 for n in size):
        return np.zeros(size, dtype=arr.dtype)
    if len(size) == 0:
        return np.array(arr[storage_offset]).copy()
    itemsize = arr.dtype.itemsize
    view = np.lib.stride_tricks.as_strided(
        arr[storage_offset:], shape=size, strides=tuple(s * itemsize for s in stride),
        writeable=False)
    return np.ascontiguousarray(view)


def _rebuild_parameter(data, requires_grad=False, backward_hooks=None):
    return data


class _Unpickler(pickle.Unpickler):
    def __init__(self, fh, zf: zipfile.ZipFile, prefix: str):
        super().__init__(fh)
        self._zf = zf
        self._prefix = prefix
        self._cache: Dict[str, np.ndarray] = {}

    def find_class(self, module, name):
        if module == "collections" and name == "OrderedDict":
            return collections.OrderedDict
        if module == "torch._utils" and name == "_rebuild_tensor_v2":
            return _rebuild_tensor_v2
        if module == "torch._utils" and name == "_rebuild_parameter":
            return _rebuild_parameter
        if module == "torch" and name in _STORAGE_DTYPES:
            return _StorageType(name)
        # Lightning checkpoints carry optimizer/callback state that the
        # inference path never looks at: decode it to inert placeholders.
        return _Opaque

    def persistent_load(self, pid):
        if not isinstance(pid, tuple) or pid[0] != "storage":
            raise CheckpointError(f"unsupported persistent id {pid!r}")
        _, stype, key, _location, numel = pid
        if not isinstance(stype, _StorageType):
            raise CheckpointError(f"unsupported storage type {stype!r}")
        key = str(key)
        if key not in self._cache:
            raw = self._zf.read(f"{self._prefix}/data/{key}")
            arr = np.frombuffer(raw, dtype=stype.dtype)
            if arr.size < int(numel):
                raise CheckpointError(f"storage {key} truncated: {arr.size} < {numel}")
            self._cache[key] = arr
        return self._cache[key]


def load_ckpt(path) -> dict:
    """Decode a ``.ckpt`` into plain Python: tensors become ``numpy`` arrays."""
    try:
        zf = zipfile.ZipFile(path)
    except zipfile.BadZipFile as e:
        raise CheckpointError(f"{path}: not a torch zip checkpoint ({e})") from e
    with zf:
        pkl = [n for n in zf.namelist() if n.endswith("/data.pkl")]
        if len(pkl) != 1:
            raise CheckpointError(f"{path}: expected exactly one data.pkl, found {pkl}")
        prefix = pkl[0][: -len("/data.pkl")]
        bo = f"{prefix}/byteorder"
        if bo in zf.namelist() and zf.read(bo).strip() != b"little":
            raise CheckpointError(f"{path}: only little-endian checkpoints are supported")
        with zf.open(pkl[0]) as fh:
            obj = _Unpickler(fh, zf, prefix).load()
    return obj


def load_state_dict(path) -> Tuple[Dict[str, np.ndarray], dict]:
    """Return ``(state_dict, hyper_parameters)`` the way the reference CLI consumes them.

    Mirrors /root/reference/infer_alns.py:71-82: keys lose their ``model.``
    prefix and ``model.seq2pair`` (a stale buffer of shape ``(1225, 50)``) is
    dropped.  A bare ``state_dict`` (no Lightning wrapper) is accepted too.
    """
    obj = load_ckpt(path)
    if not isinstance(obj, dict):
        raise CheckpointError(f"{path}: top-level object is {type(obj).__name__}, expected dict")
    sd = obj["state_dict"] if "state_dict" in obj else obj
    hp = obj.get("hyper_parameters", {}) if "state_dict" in obj else {}
    out: Dict[str, np.ndarray] = {}
    for k, v in sd.items():
        if k == "model.seq2pair" or not isinstance(v, np.ndarray):
            continue
        out[k.replace("model.", "")] = v
    if not isinstance(hp, dict):
        hp = {}
    return out, dict(hp)
