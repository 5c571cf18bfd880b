#!/usr/bin/env python3
"""Generate tests/golden/mha.npz by running the REFERENCE class MultiHeadAttention
(/root/reference/phyloformer/attention.py:53-91) on CPU with seeded synthetic weights.
Build container only; writes data (weights, inputs, expected outputs), no reference source.

    python oracle/gen_golden_mha.py
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")


def main():
    import torch
    from phyloformer.attention import MultiHeadAttention
    torch.manual_seed(20260101)
    out = {}
    # (name, B, R, C, gain on q/k weights — larger gain = peakier softmax)
    for name, B, R, C, gain in [("small", 1, 3, 50, 1.0), ("ragged", 2, 2, 97, 3.0), ("peaky", 1, 1, 160, 8.0),
                                ("one_key", 1, 2, 1, 1.0)]:
        m = MultiHeadAttention(nb_heads=4, embed_dim=64).eval()
        with torch.no_grad():
            m.q_proj.weight.mul_(gain)
            m.k_proj.weight.mul_(gain)
            m.q_proj.bias.normal_(0, 0.3)
            m.k_proj.bias.normal_(0, 0.3)
            x = torch.randn(B, R, C, 64) * 1.5
            y = m(x)
        for k, v in m.state_dict().items():
            out[f"{name}/{k}"] = v.numpy().astype(np.float32)
        out[f"{name}/x"] = x.numpy().astype(np.float32)
        out[f"{name}/y"] = y.numpy().astype(np.float32)
        print(name, tuple(x.shape), "max|y| %.3f" % float(y.abs().max()))
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "mha.npz"), **out)


if __name__ == "__main__":
    main()
