#!/usr/bin/env python3
"""Joules per forward / per k_main tile of the real kernels, with k_main's phases switched off one at a time.

    python3 tools/energy_forward.py [seconds per variant]      ->  gpurun_out/energy_forward.json

Every variant runs the headline step (60 x 500, batch 16, one stream so that launches are serial) back to
back for a few seconds between two reads of the socket energy counter (phyloformer_amd/smi.py: librocm_smi64
in-process).  `ablate` bits (pf_device.hip.h, results invalid - energy only): 4 no next-row statistics phase,
8 no apply phase, 16 no FFN, 32 copy only.  The k_main share of a variant's energy is taken from the HIP-event
time share of the same run (profile = 1 brackets every kernel).
"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from phyloformer_amd.engine import Engine  # noqa: E402
from phyloformer_amd.msa_sim import simulate_batch  # noqa: E402
from phyloformer_amd.smi import Smi  # noqa: E402
from phyloformer_amd.weights import load_weights  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    B, N, L = 16, 60, 500
    P = N * (N - 1) // 2
    w = load_weights(os.path.join(REPO, "models", "pf.ckpt"))
    idx = simulate_batch(8, N, L, seed=3)
    idx = np.ascontiguousarray(idx[np.arange(B) % 8])
    rows = []
    with Engine(w, device=0) as eng, Smi(0) as smi:
        d_idx, d_out = eng.malloc(idx.nbytes), eng.malloc(B * P * 4)
        eng.h2d(d_idx, idx)
        eng.set_option("two_streams", 0)
        t0 = time.perf_counter()
        j0 = smi.energy_j()
        time.sleep(2.0)
        idle_w = (smi.energy_j() - j0) / (time.perf_counter() - t0)
        tiles = B * P * ((L + 31) // 32)
        for name, ab in (("full", 0), ("no next-row statistics", 4), ("no apply", 8), ("no FFN", 16),
                         ("FFN only = no apply, no statistics", 12),
                         ("copy only (load, store)", 32 + 16 + 8), ("no apply, no FFN", 24)):
            eng.set_option("ablate", ab)
            for _ in range(3):
                eng.forward_device(d_idx, B, N, L, d_out)
            eng.synchronize()
            # pass 1: energy, unprofiled
            n, j0, t0 = 0, smi.energy_j(), time.perf_counter()
            pw, ck = [], []
            while time.perf_counter() - t0 < seconds:
                for _ in range(4):
                    eng.forward_device(d_idx, B, N, L, d_out)
                eng.synchronize()
                n += 4
                pw.append(smi.power_w())
                ck.append(smi.sclk_mhz())
            dt = time.perf_counter() - t0
            joules = smi.energy_j() - j0
            # pass 2: time share of k_main (HIP events around every kernel serialise the stream a little)
            eng.set_option("profile", 1)
            eng.profile_reset()
            for _ in range(4):
                eng.forward_device(d_idx, B, N, L, d_out)
            eng.synchronize()
            prof = {k: eng.profile_get(k) for k in ("embed", "rowfin", "colstats", "colfin", "main")}
            eng.set_option("profile", 0)
            tot_ms = sum(v[1] for v in prof.values())
            main_ms = prof["main"][1] / max(prof["main"][0], 1)
            rows.append({"variant": name, "ablate": ab, "forwards": n, "seconds": round(dt, 3),
                         "ms_per_forward": round(dt / n * 1e3, 3), "joules_per_forward": round(joules / n, 3),
                         "avg_w": round(joules / dt, 1), "median_sampled_w": float(np.median(pw)),
                         "median_sclk_mhz": float(np.median(ck)),
                         "k_main_ms_per_launch": round(main_ms, 4),
                         "k_main_time_share": round(prof["main"][1] / tot_ms, 4),
                         "uj_per_tile_whole_forward": round(joules / n / (6 * tiles) * 1e6, 3)})
            print(rows[-1], flush=True)
        eng.set_option("ablate", 0)
        eng.free(d_idx)
        eng.free(d_out)
    out = {"shape": f"{N}x{L} batch {B}", "idle_w": round(idle_w, 1), "tiles_per_launch": tiles, "rows": rows}
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "energy_forward.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
