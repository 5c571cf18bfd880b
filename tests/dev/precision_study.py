"""Numerical study (CPU, oracle only): how much final-distance error does rounding the *activation*
operand of the big GEMMs (FFN 64->256->64, Wv', Wo) to a short format cost?  Used to decide between the
3-pass split-bf16 MFMA scheme and cheaper 2-pass schemes.  Not part of the product path."""
import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import oracle.pf_oracle as O
from phyloformer_amd import weights as Wt

def rnd(x, fmt):
    if fmt == "f64": return x
    if fmt == "f16": return x.astype(np.float16).astype(x.dtype)
    if fmt.startswith("m"):         # keep n mantissa bits (incl. implicit), round to nearest
        n = int(fmt[1:])
        m, e = np.frexp(x)
        return np.ldexp(np.round(m * 2.0**n) / 2.0**n, e)
    raise ValueError(fmt)

orig = O._mm
def make(fmt, wfmt):
    def mm(x, W):
        if W.shape[0] >= 64:                  # FFN / v_proj / out_proj, not q/k/head
            return orig(rnd(x, fmt), rnd(W, wfmt))
        return orig(x, W)
    return mm

def main():
    w = {k: v.astype(np.float64) for k, v in Wt.load_weights("models/pf.ckpt").tensors.items()}
    z = np.load("tests/golden/configs.npz")
    cases = [("c2", z["c2_idx"][:2]), ("c3", z["c3_idx"][:1, :24, :200])]
    for name, idx in cases:
        O._mm = orig
        ref = np.stack([O.forward(w, a, dtype=np.float64) for a in idx])
        for fmt, wfmt in [("m8","f64"),("f16", "f64"), ("m11", "f64"), ("m12","f64"),("m16", "m16"), ("m16","f64")]:
            O._mm = make(fmt, wfmt)
            out = np.stack([O.forward(w, a, dtype=np.float64) for a in idx])
            print(name, fmt, wfmt, "max abs err %.3g" % np.abs(out - ref).max(), "max ref %.3g" % ref.max(), flush=True)
main()
