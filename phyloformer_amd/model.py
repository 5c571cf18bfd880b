"""Host-side mirror of the reference ``Phyloformer`` module for inference.

Same surface as /root/reference/phyloformer/model.py:109-187 where it matters
to ``infer_alns.py``: construct, load a state dict, call on an alignment, get
the distance vector.  The arithmetic runs in ``libphyloformer_amd.so`` on one
MI355X (``engine.Engine``); this class only adapts shapes and error types.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import fasta
from .engine import Engine
from .weights import ModelWeights, from_state_dict, load_weights

MAX_SEQS = 200  # SEQ2PAIR = seq2pair(200), model.py:39


class Phyloformer:
    """``Phyloformer(...)`` + ``load_state_dict`` + ``.to(device)`` + ``.eval()`` in one object."""

    def __init__(self, weights: Optional[ModelWeights] = None, device: int = 0, engine_factory=None, **_ignored):
        # The reference constructor swallows unknown keyword arguments (model.py:122);
        # checkpoint hyper-parameters are read from the tensors instead.
        # engine_factory(weights, device): the engine to bind - ``Engine`` unless a caller passes another one
        # explicitly (infer_alns.py hands its test hook through here; the library API never reads the environment)
        self.device = device
        self._engine_factory = engine_factory or Engine
        self.weights: Optional[ModelWeights] = None
        self.engine: Optional[Engine] = None
        if weights is not None:
            self._bind(weights)

    @classmethod
    def from_checkpoint(cls, path, device: int = 0, engine_factory=None) -> "Phyloformer":
        return cls(load_weights(path), device=device, engine_factory=engine_factory)

    def load_state_dict(self, state_dict: Dict[str, np.ndarray], strict: bool = False):
        """Accepts the prefix-stripped dict of infer_alns.py:75-82 (numpy arrays or tensors)."""
        sd = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v))
              for k, v in state_dict.items() if k != "seq2pair"}
        self._bind(from_state_dict(sd))
        return self

    def _bind(self, weights: ModelWeights):
        if self.engine is not None:
            self.engine.close()
        self.weights = weights
        self.engine = self._engine_factory(weights, self.device)

    def eval(self):
        return self

    def to(self, device):
        return self

    @staticmethod
    def _as_indices(x) -> np.ndarray:
        a = x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
        if a.dtype == np.uint8 and a.ndim in (2, 3):
            return a
        if a.ndim in (3, 4) and a.shape[-3] == fasta.N_ALPHABET:
            # the reference's input: one-hot float [B, 22, L, N] (infer_alns.py:112)
            return fasta.from_one_hot(a)
        if a.ndim in (2, 3) and np.issubdtype(a.dtype, np.integer):
            if a.min(initial=0) < 0 or a.max(initial=0) >= fasta.N_ALPHABET:
                raise ValueError("residue index out of range 0..21")
            return a.astype(np.uint8)
        raise ValueError(f"expected uint8 indices [B, N, L] or one-hot [B, 22, L, N], got {a.shape} {a.dtype}")

    def forward(self, x) -> np.ndarray:
        """Distance vector(s): ``[P]`` for one alignment, ``[B, P]`` for a batch.

        Like the reference's ``torch.squeeze`` (model.py:185) a batch of one
        one-hot alignment yields ``[P]`` and ``N == 2`` yields a 0-dim array.
        """
        if self.engine is None:
            raise RuntimeError("no weights loaded")
        idx = self._as_indices(x)
        out = self.engine.forward(idx)
        one_hot_input = getattr(x, "ndim", idx.ndim) == 4
        return np.squeeze(out) if one_hot_input or out.shape[-1] == 1 else out

    __call__ = forward

    def close(self):
        if self.engine is not None:
            self.engine.close()
            self.engine = None
