#!/usr/bin/env python3
"""CPU study (oracle only, float64 carrier; not on the product path): would CHEAPER matrix passes than the three
bf16 ones keep the distances under 1e-4?  k_main spends 37 % of a tile's energy in 264 MFMAs = 3 bf16 passes
(hi*hi + lo*hi + hi*lo); gfx950's int8 MFMA runs at twice the bf16 rate (MI355X_MICROARCH.md), so

  A  "bf16 + i8 cross":  hi*hi stays one bf16 pass, the two cross terms - each 2^-8 of the product - go through int8
     operands (per-token / per-weight-row scale, 7 bits + sign): 1 + 2 * 0.5 = 2 pass-equivalents instead of 3;
  B  "i8 two limbs":     both operands as 16-bit fixed point = two balanced int8 limbs, products a1 b1, a1 b0, a0 b1:
     3 * 0.5 = 1.5 pass-equivalents;
  C  "bf16 x 3":         what the device does today (the yardstick: its error here is the budget already spent).

Each scheme replaces the big GEMMs of the oracle (FFN 64 -> 256 -> 64, v_proj, out_proj; the device's algebra is
re-associated, so this is a proxy - the same proxy tests/dev/precision_study.py used for the 2-pass question in
round 1).  Reported: max-abs error of the final distances against the exact float64 forward.
Run: python tests/dev/limb_study.py"""
import os, sys
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import oracle.pf_oracle as O                                   # noqa: E402
from phyloformer_amd import weights as Wt                      # noqa: E402
from phyloformer_amd.msa_sim import simulate_batch             # noqa: E402


def bf16(x):
    """round-to-nearest-even to 8 significant bits (exponent range is not an issue here)"""
    m, e = np.frexp(x)
    return np.ldexp(np.round(m * 256.0) / 256.0, e)


def q_i8(x, axis):
    """symmetric int8 with one scale per vector along `axis` (the K axis): values k * s, k in -127..127"""
    s = np.abs(x).max(axis=axis, keepdims=True) / 127.0
    s = np.where(s == 0, 1.0, s)
    return np.round(x / s) * s


def q_i16_limbs(x, axis):
    """16-bit fixed point with one scale per K-vector, as two balanced int8 limbs a1 * 256 + a0 (returned scaled)"""
    s = np.abs(x).max(axis=axis, keepdims=True) / 32767.0
    s = np.where(s == 0, 1.0, s)
    k = np.round(x / s)
    a1 = np.floor((k + 128) / 256.0)
    a0 = k - 256 * a1
    return a1 * 256 * s, a0 * s


orig = O._mm


def scheme(name):
    def mm(x, W):
        if W.shape[0] < 64:                   # q/k (64 -> 4) and the head stay as they are
            return orig(x, W)
        x2 = np.ascontiguousarray(x).reshape(-1, x.shape[-1])
        if name == "bf16x3":
            xh = bf16(x2); xl = bf16(x2 - xh); Wh = bf16(W); Wl = bf16(W - Wh)
            out = xh @ Wh.T + xl @ Wh.T + xh @ Wl.T
        elif name == "bf16+i8cross":
            xh = bf16(x2); xl = x2 - xh; Wh = bf16(W); Wl = W - Wh
            out = xh @ Wh.T + q_i8(xl, 1) @ q_i8(Wh, 1).T + q_i8(xh, 1) @ q_i8(Wl, 1).T
        elif name == "bf16+i8cross(lo exact)":   # only the hi operands of the cross terms quantised (lo parts as bf16)
            xh = bf16(x2); xl = bf16(x2 - xh); Wh = bf16(W); Wl = bf16(W - Wh)
            out = xh @ Wh.T + xl @ q_i8(Wh, 1).T + q_i8(xh, 1) @ Wl.T
        elif name == "i8x2limbs":
            x1, x0 = q_i16_limbs(x2, 1); W1, W0 = q_i16_limbs(W, 1)
            out = x1 @ W1.T + x1 @ W0.T + x0 @ W1.T
        else:
            raise ValueError(name)
        return out.reshape(x.shape[:-1] + (W.shape[0],))
    return mm


def cases():
    wpf = {k: v.astype(np.float64) for k, v in Wt.load_weights(os.path.join(REPO, "models/pf.ckpt")).tensors.items()}
    wind = {k: v.astype(np.float64) for k, v in Wt.load_weights(os.path.join(REPO, "models/pf_indel.ckpt")).tensors.items()}
    z = np.load(os.path.join(REPO, "tests/golden/configs.npz"))
    taps = np.load(os.path.join(REPO, "tests/golden/taps_tiny.npz"))
    return [("configs[1] 20x200 x2 (pf)", wpf, z["c2_idx"][:2]),
            ("configs[2] 60x500, 24 x 200 corner (pf)", wpf, z["c3_idx"][:, :24, :200]),
            ("OOD tiny_taps 5x16 (pf)", wpf, taps["idx"][None]),
            ("OOD 4x32 gapped (pf_indel)", wind, simulate_batch(2, 4, 32, seed=3, gaps=True)),
            ("OOD 24x33 (pf)", wpf, simulate_batch(1, 24, 33, seed=18, gaps=True))]


def main():
    names = ["bf16x3", "bf16+i8cross", "bf16+i8cross(lo exact)", "i8x2limbs"]
    print(f"{'case':42s} " + " ".join(f"{n:>24s}" for n in names) + "   max |d|")
    for label, w, idx in cases():
        O._mm = orig
        ref = np.stack([O.forward(w, a, dtype=np.float64) for a in idx])
        row = []
        for n in names:
            O._mm = scheme(n)
            got = np.stack([O.forward(w, a, dtype=np.float64) for a in idx])
            row.append(float(np.abs(got - ref).max()))
        O._mm = orig
        print(f"{label:42s} " + " ".join(f"{d:24.3e}" for d in row) + f"   {np.abs(ref).max():.3g}", flush=True)


if __name__ == "__main__":
    main()
