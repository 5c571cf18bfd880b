#!/usr/bin/env python3
"""Throughput at every BASELINE.json configuration (device-resident inputs, 1 GPU), batch 1 and batched.

    python tools/configs_bench.py [out.json]

Prints one line per case and writes them as JSON (default gpurun_out/configs.json; the copy that is judged
lives in profiles/r<NN>_configs.json).  `frac_mfma` / `frac_hbm` are the whole-forward roofline fractions of
SURVEY.md section 8d: 602,240 algorithmic flop and 3,328 algorithmic bytes per token against 2.5 PFLOP/s
dense bf16 and 8 TB/s (the split-fp16 scheme issues three MFMA passes, so frac_mfma tops out at 1/3).
"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phyloformer_amd.engine import Engine
from phyloformer_amd.msa_sim import simulate_batch
from phyloformer_amd.weights import load_weights

FLOPS_PER_TOKEN, BYTES_PER_TOKEN = 602240, 3328
PEAK_TFLOPS, PEAK_TBS = 2500.0, 8.0


def run(tag, ckpt, n, l, B, gaps=False, reps=5):
    e = Engine(load_weights(os.path.join(REPO, "models", ckpt)), 0)
    base = simulate_batch(min(B, 4), n, l, seed=2, gaps=gaps)
    idx = np.ascontiguousarray(base[np.arange(B) % base.shape[0]])
    P = n * (n - 1) // 2
    d_idx = e.malloc(idx.nbytes)
    d_out = e.malloc(B * P * 4)
    e.h2d(d_idx, idx)
    t0 = time.perf_counter()
    for _ in range(3):
        e.forward_device(d_idx, B, n, l, d_out)
    e.synchronize()
    # at least 0.3 s of timed work (a 0.4 ms forward timed five times mostly measures the clock ramping up)
    reps = max(reps, int(0.3 / ((time.perf_counter() - t0) / 3)))
    for _ in range(reps // 2):
        e.forward_device(d_idx, B, n, l, d_out)
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        e.forward_device(d_idx, B, n, l, d_out)
    e.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # the host-buffer entry point (PCIe-inclusive, one synchronisation per call): what a caller with one
    # alignment at a time sees
    e.forward(idx)
    t0 = time.perf_counter()
    for _ in range(reps):
        e.forward(idx)
    dth = (time.perf_counter() - t0) / reps
    rec_reps = reps
    tok = B * P * l
    rec = {"config": tag, "ckpt": ckpt, "n_seqs": n, "n_sites": l, "gapped": gaps, "batch": B, "timed_forwards": rec_reps,
           "ms_per_forward": round(dt * 1e3, 4), "alignments_per_s": round(B / dt, 2),
           "alignments_per_s_host_buffers": round(B / dth, 2), "gtoken_per_s": round(tok / dt / 1e9, 4),
           "tflops_algorithmic": round(FLOPS_PER_TOKEN * tok / dt / 1e12, 2),
           "frac_mfma": round(FLOPS_PER_TOKEN * tok / dt / 1e12 / PEAK_TFLOPS, 4),
           "frac_hbm": round(BYTES_PER_TOKEN * tok / dt / 1e12 / PEAK_TBS, 4)}
    print(f"{tag:3s} {ckpt:14s} {n:3d} x {l:4d} batch {B:5d}: {dt * 1e3:9.3f} ms  {B / dt:10.1f} aln/s "
          f"({B / dth:9.1f} with host buffers)  {tok / dt / 1e9:5.2f} Gtoken/s  frac_mfma {rec['frac_mfma']:.3f}  "
          f"frac_hbm {rec['frac_hbm']:.3f}", flush=True)
    e.free(d_idx)
    e.free(d_out)
    e.close()
    return rec


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "configs.json")
    recs = [
        run("C2", "pf.ckpt", 20, 200, 1),
        run("C2", "pf.ckpt", 20, 200, 64),
        run("C2", "pf.ckpt", 20, 200, 1024),
        run("C3", "pf.ckpt", 60, 500, 1),
        run("C3", "pf.ckpt", 60, 500, 16),
        run("C4", "pf.ckpt", 60, 2000, 1),
        run("C4", "pf.ckpt", 60, 2000, 4),
        run("C5", "pf_indel.ckpt", 200, 500, 1, gaps=True),
        run("C5", "pf_indel.ckpt", 200, 500, 2, gaps=True),
    ]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as fh:
        json.dump({"device": "MI355X (gfx950), 1 GPU", "peaks": {"mfma_bf16_dense_tflops": PEAK_TFLOPS, "hbm_tb_s": PEAK_TBS},
                   "cases": recs}, fh, indent=1)


if __name__ == "__main__":
    main()
