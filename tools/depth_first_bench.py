#!/usr/bin/env python3
"""Breadth-first (one batch of 16, two half-batches on two streams: the default) against depth-first processing - k
engines on k host threads, each pushing LONE alignments through all six blocks back to back, device-resident - so that
an alignment's 227 MB residual stream has a chance to stay in the 256 MB Infinity Cache between the kernels that touch it.
    python tools/depth_first_bench.py [seconds per point]"""
import os, sys, threading, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phyloformer_amd.engine import Engine
from phyloformer_amd.msa_sim import simulate_batch
from phyloformer_amd.weights import load_weights


def run(engs, bufs, B, seconds, N=60, L=500):
    stop = time.perf_counter() + seconds
    counts = [0] * len(engs)

    def work(i):
        e, (d_idx, d_out) = engs[i], bufs[i]
        while time.perf_counter() < stop:
            for _ in range(8):
                e.forward_device(d_idx, B, N, L, d_out)
            e.synchronize()
            counts[i] += 8 * B
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(engs))]
    for t in ths: t.start()
    for t in ths: t.join()
    return sum(counts) / (time.perf_counter() - t0)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    w = load_weights(os.path.join(REPO, "models", "pf.ckpt"))
    N, L = 60, 500
    P = N * (N - 1) // 2
    idx16 = np.ascontiguousarray(np.resize(simulate_batch(8, N, L, seed=3), (16, N, L)))
    engs = [Engine(w, 0) for _ in range(4)]
    bufs = []
    for e in engs:
        d_idx, d_out = e.malloc(idx16.nbytes), e.malloc(16 * P * 4)
        e.h2d(d_idx, idx16)
        bufs.append((d_idx, d_out))
    for rep in range(2):
        print(f"breadth-first, batch 16, one engine (two half-batches on two streams): {run(engs[:1], bufs[:1], 16, seconds):7.1f} alignments/s", flush=True)
        for k in (1, 2, 3, 4):
            for e in engs[:k]:
                e.set_option("two_streams", 0)
            print(f"depth-first, {k} engine(s) x lone alignments:                          {run(engs[:k], bufs[:k], 1, seconds):7.1f} alignments/s", flush=True)
            for e in engs[:k]:
                e.set_option("two_streams", 1)
        for b in (2, 4):
            print(f"2 engines x batch {b}:                                                   {run(engs[:2], bufs[:2], b, seconds):7.1f} alignments/s", flush=True)
    for e in engs:
        e.close()


main()
