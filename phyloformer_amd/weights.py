"""Model weights: validation and the flat fp32 blob handed across the C ABI.

The blob is the only weight format the native library sees
(``pf_weights_t`` in ``include/phyloformer_amd.h``).  Everything derived from it
— the ReLU'd embedding table, fp16 hi/lo splits, MFMA operand permutations —
is produced inside ``pf_create`` on the host.

Tensor names and shapes follow the reference ``state_dict``
(/root/reference/phyloformer/model.py:60-85,138-164,
/root/reference/phyloformer/attention.py:43-47), prefix ``model.`` stripped.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

from .ckpt import CheckpointError, load_state_dict

N_ALPHABET = 22

# (name suffix, shape as a function of (E, H, F=4E)) in blob order, per attention sub-block
_ATTN_FIELDS = [
    ("{a}_norm.weight", lambda E, H: (E,)),
    ("{a}_norm.bias", lambda E, H: (E,)),
    ("{a}_attention.q_proj.weight", lambda E, H: (H, E)),
    ("{a}_attention.q_proj.bias", lambda E, H: (H,)),
    ("{a}_attention.k_proj.weight", lambda E, H: (H, E)),
    ("{a}_attention.k_proj.bias", lambda E, H: (H,)),
    ("{a}_attention.v_proj.weight", lambda E, H: (E, E)),
    ("{a}_attention.v_proj.bias", lambda E, H: (E,)),
    ("{a}_attention.out_proj.weight", lambda E, H: (E, E)),
    ("{a}_attention.out_proj.bias", lambda E, H: (E,)),
]


def blob_layout(n_blocks: int, n_heads: int, embed_dim: int) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered ``(state_dict key, logical shape)`` list defining the flat blob."""
    E, H = embed_dim, n_heads
    out: List[Tuple[str, Tuple[int, ...]]] = [
        ("embedding_block.0.weight", (E, N_ALPHABET)),
        ("embedding_block.0.bias", (E,)),
    ]
    for b in range(n_blocks):
        p = f"attention_blocks.{b}."
        for a in ("row", "col"):
            for name, shp in _ATTN_FIELDS:
                out.append((p + name.format(a=a), shp(E, H)))
        out += [
            (p + "ffn_norm.weight", (E,)),
            (p + "ffn_norm.bias", (E,)),
            (p + "ffn.0.weight", (4 * E, E)),
            (p + "ffn.0.bias", (4 * E,)),
            (p + "ffn.3.weight", (E, 4 * E)),
            (p + "ffn.3.bias", (E,)),
        ]
    out += [("pwFNN.0.weight", (E,)), ("pwFNN.0.bias", (1,))]
    return out


@dataclass
class ModelWeights:
    n_blocks: int
    n_heads: int
    embed_dim: int
    tensors: Dict[str, np.ndarray]  # logical shapes (1×1 conv kernels squeezed), float32

    @property
    def n_params(self) -> int:
        return int(sum(v.size for v in self.tensors.values()))

    def blob(self) -> np.ndarray:
        parts = [self.tensors[k].reshape(-1) for k, _ in
                 blob_layout(self.n_blocks, self.n_heads, self.embed_dim)]
        return np.ascontiguousarray(np.concatenate(parts).astype(np.float32))

    def __getitem__(self, key: str) -> np.ndarray:
        return self.tensors[key]


def from_state_dict(sd: Dict[str, np.ndarray]) -> ModelWeights:
    """Validate a (prefix-stripped) state dict; dimensions are read from the tensors.

    The reference constructor ignores the checkpoint's ``hyper_parameters``
    (their names do not match its arguments, model.py:112-123) and relies on
    the defaults 6/4/64; here the architecture is derived from the tensor
    shapes so a mismatch is an error instead of silent garbage.
    """
    try:
        emb = np.asarray(sd["embedding_block.0.weight"])
    except KeyError as e:
        raise CheckpointError("state dict has no embedding_block.0.weight") from e
    E = int(emb.shape[0])
    if emb.reshape(E, -1).shape[1] != N_ALPHABET:
        raise CheckpointError(f"embedding expects {N_ALPHABET} input channels, got {emb.shape}")
    n_blocks = 0
    while f"attention_blocks.{n_blocks}.ffn.0.weight" in sd:
        n_blocks += 1
    if n_blocks == 0:
        raise CheckpointError("state dict has no attention blocks")
    H = int(np.asarray(sd["attention_blocks.0.row_attention.q_proj.weight"]).shape[0])
    tensors: Dict[str, np.ndarray] = {}
    for key, shape in blob_layout(n_blocks, H, E):
        if key not in sd:
            raise CheckpointError(f"missing tensor {key}")
        t = np.asarray(sd[key], dtype=np.float32)
        if int(np.prod(t.shape)) != int(np.prod(shape)):
            raise CheckpointError(f"{key}: shape {t.shape} does not match expected {shape}")
        tensors[key] = np.ascontiguousarray(t.reshape(shape))
    return ModelWeights(n_blocks, H, E, tensors)


def load_weights(path) -> ModelWeights:
    sd, _hp = load_state_dict(path)
    return from_state_dict(sd)


def random_weights(seed: int = 0, n_blocks: int = 6, n_heads: int = 4,
                   embed_dim: int = 64, scale: float = 1.0) -> ModelWeights:
    """Random-init weights with torch-like fan-in scaling (tests only)."""
    rng = np.random.default_rng(seed)
    tensors = {}
    for key, shape in blob_layout(n_blocks, n_heads, embed_dim):
        if key.endswith("norm.weight"):
            t = 1.0 + 0.1 * rng.standard_normal(shape)
        elif key.endswith("norm.bias"):
            t = 0.1 * rng.standard_normal(shape)
        else:
            fan_in = shape[-1] if len(shape) > 1 else embed_dim
            bound = scale / np.sqrt(fan_in)
            t = rng.uniform(-bound, bound, size=shape)
        tensors[key] = t.astype(np.float32)
    return ModelWeights(n_blocks, n_heads, embed_dim, tensors)
