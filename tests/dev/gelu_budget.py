#!/usr/bin/env python3
"""CPU study (oracle only, float64; not on the product path): how much final-distance error does a CHEAPER erf-GELU
polynomial cost?  VERDICT r03 / next 3b.

The device evaluates  2a gelu(h) = x + |x| (1 - 2^P(|x|)),  x = a h,  a^2 = log2(e) / 2,  with P a degree-5 fit of
log2 erfc(|h| / sqrt 2)  (pf_device.hip.h::gelu_scaled: 5 FMA + v_exp + sub + FMA per hidden value).  A degree-4 / 3
fit would save one / two FMAs of the ten VALU instructions a hidden value costs (energy table, DESIGN section 9:
-6 % of the hidden loop for two).  This script
  1. refits P for degrees 3, 4, 5 (minimax of the GELU's absolute error over |h| <= 8, leading coefficient
     constrained negative so that the tail vanishes without a clamp) and prints each fit's max |gelu error|;
  2. injects each fit into the oracle's GELU - everything else exact, float64 - and reports the max-abs change of the
     final distances on the BASELINE goldens and on the three worst out-of-distribution parity cases of
     tests/test_gpu_parity.py (tiny_taps 5x16, 4x32 gapped pf_indel, 24x33), where the GPU's own error is 6-9e-5 of
     the 1e-4 budget.
Run: python tests/dev/gelu_budget.py      (about five minutes on 8 cores)"""
import math, os, sys
import numpy as np
from scipy.optimize import minimize
from scipy.special import erfc

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import oracle.pf_oracle as O                                   # noqa: E402
from phyloformer_amd import weights as Wt                      # noqa: E402
from phyloformer_amd.msa_sim import simulate_batch             # noqa: E402

A = math.sqrt(math.log2(math.e) / 2)
DEVICE_P5 = [-0.00107098569, 0.0136151873, -0.084594565, -0.637684925, -1.35494915, -0.00003762]   # highest first

H = np.concatenate([np.linspace(0, 8, 40001), np.linspace(8, 60, 2001)])


def gelu_err(coef, h=H):
    """|gelu_fit(h) - gelu(h)| for h >= 0 (the form is exact in the sign: the error is even in h)."""
    p = np.polyval(coef, A * h)
    return 0.5 * h * np.abs(np.exp2(np.minimum(p, 0.0)) - erfc(h / math.sqrt(2)))


def fit(deg):
    hh = np.linspace(1e-3, 6.5, 4000)
    tgt = np.log2(erfc(hh / math.sqrt(2)))
    wgt = hh * erfc(hh / math.sqrt(2))
    c0 = np.polyfit(A * hh, tgt, deg, w=wgt)                   # weighted least squares start
    def cost(c):
        e = gelu_err(c)
        return e.max() + (1e3 * c[0] if c[0] > 0 else 0.0)     # negative leading coefficient: 2^P -> 0 in the tail
    best = c0
    for _ in range(6):
        r = minimize(cost, best, method="Nelder-Mead", options={"xatol": 1e-12, "fatol": 1e-14, "maxiter": 40000, "maxfev": 40000})
        best = r.x
    return best


def gelu_with(coef):
    def g(x):
        h = np.abs(x)
        q2 = np.exp2(np.polyval(coef, A * h))                  # ~ erfc(|h| / sqrt 2)
        return (0.5 * (x + h * (1.0 - q2))).astype(x.dtype)
    return g


def cases():
    wpf = {k: v.astype(np.float64) for k, v in Wt.load_weights(os.path.join(REPO, "models/pf.ckpt")).tensors.items()}
    wind = {k: v.astype(np.float64) for k, v in Wt.load_weights(os.path.join(REPO, "models/pf_indel.ckpt")).tensors.items()}
    z = np.load(os.path.join(REPO, "tests/golden/configs.npz"))
    taps = np.load(os.path.join(REPO, "tests/golden/taps_tiny.npz"))
    out = [("configs[1] 20x200 x3 (pf)", wpf, z["c2_idx"]),
           ("configs[2] 60x500, 30 x 250 corner (pf)", wpf, z["c3_idx"][:, :30, :250]),
           ("OOD tiny_taps 5x16 (pf)", wpf, taps["idx"][None] if taps["idx"].ndim == 2 else taps["idx"]),
           ("OOD 4x32 gapped (pf_indel)", wind, simulate_batch(2, 4, 32, seed=3, gaps=True)),
           ("OOD 24x33 (pf)", wpf, simulate_batch(1, 24, 33, seed=18, gaps=True)),
           ("OOD 33x12 (pf)", wpf, simulate_batch(1, 33, 12, seed=19))]
    from phyloformer_amd.fasta import load_alignment
    for name in ("0_20_tips", "1_40_tips"):
        idx, _ids = load_alignment(os.path.join(REPO, "data/testdata/msas", name + ".fa"))
        out.append((f"test MSA {name} (pf)", wpf, np.asarray(idx)[None]))
    return out


def main():
    fits = {"deg 5 (device)": np.array(DEVICE_P5)}
    for d in (5, 4, 3):
        fits[f"deg {d} refit"] = fit(d)
    print("fit: max |gelu error| over |h| <= 60, coefficients (highest first)")
    for name, c in fits.items():
        print(f"  {name:16s} {gelu_err(c).max():.3e}   " + " ".join(f"{v:.9g}" for v in c))
    exact = O.gelu
    print("\nmax-abs change of the final distances with the fit in place of erf-GELU (float64 oracle):")
    print(f"  {'case':44s} " + " ".join(f"{n:>16s}" for n in fits) + "   max |d|")
    worst = {n: 0.0 for n in fits}
    for label, w, idx in cases():
        O.gelu = exact
        ref = np.stack([O.forward(w, a, dtype=np.float64) for a in idx])
        row = []
        for n, c in fits.items():
            O.gelu = gelu_with(c)
            got = np.stack([O.forward(w, a, dtype=np.float64) for a in idx])
            d = float(np.abs(got - ref).max())
            worst[n] = max(worst[n], d)
            row.append(d)
        O.gelu = exact
        print(f"  {label:44s} " + " ".join(f"{d:16.3e}" for d in row) + f"   {np.abs(ref).max():.3g}", flush=True)
    print(f"  {'worst':44s} " + " ".join(f"{worst[n]:16.3e}" for n in fits))


if __name__ == "__main__":
    main()
