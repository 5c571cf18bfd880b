#!/usr/bin/env python3
"""Headline benchmark: alignments/s of the Phyloformer forward on synthetic LG+GC-like MSAs.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A *step* is one pass of the hot path (``pf_forward_device`` / ``pf_forward_sharded_device``) over one
batch of synthetic alignments whose residue indices are already resident in HBM.  Workload:
BASELINE.json configs[2], the headline 60-leaf / 500-site shape, ``pf.ckpt`` weights.

N > 1 (default ``--shard sites``): the global batch is ``batch x N`` alignments and every alignment is
*site-sharded* over the N ranks (rank r holds 500/N sites of every pair); row-attention statistics are
all-reduced once per block and the site sums once at the end with RCCL (7 collectives per step).
Per-GPU work is fixed as N grows -> "weak".  ``--shard alignments`` shards whole alignments instead (no
collective).  The ranks meet through ``phyloformer_amd.rendezvous.TcpGroup`` (standard library: the
launcher's RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT, no torch import in a GPU rank).

Rank 0 prints ONE JSON line (contract in the task statement).  ``value`` is the HBM-resident rate;
``value_host_buffers`` is the same step through ``pf_forward`` with host buffers (H2D of the indices, D2H of
the distances and one stream synchronisation per call) - the PCIe-inclusive rate SURVEY.md §8d defines.
Extra objects: ``roofline`` (dominant kernel ``k_main``: algorithmic flops / HIP-event time vs the dense
bf16 MFMA peak), ``cpu_baseline`` (torch op-order port of the reference on this host's cores: 1 warm-up +
3 timed forwards, median, at 60x500 and at 20x200; N = 1, rank 0 only) and ``power`` (rocm-smi samples
taken during the timed region: the forward runs at the chip's power limit).
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16
# algorithmic flops per token per k_main launch (MAC = 2 flops):
#   FFN 64->256->64 = 65,536; column out_proj 64x64 = 8,192; next block's row v/q/k
#   projection 72x64 = 9,216; row mix apply (4 heads + bias) x 64 = 640.  The last block
#   swaps the row projection for the 64->1 head (128).
FLOPS_MAIN_MID = 65536 + 8192 + 9216 + 640
FLOPS_MAIN_LAST = 65536 + 8192 + 640 + 128
FLOPS_ALG_PER_TOKEN = 602240           # whole forward, SURVEY.md §8d
PMC_FILE = os.path.join("profiles", "pmc_k_main.json")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="alignments per GPU per step")
    ap.add_argument("--n-seqs", type=int, default=60)
    ap.add_argument("--n-sites", type=int, default=500)
    ap.add_argument("--ckpt", default=os.path.join(REPO, "models", "pf.ckpt"))
    ap.add_argument("--shard", choices=["sites", "alignments"], default="sites")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip HIP-event bracketing of kernels")
    ap.add_argument("--no-power", action="store_true", help="skip the rocm-smi power / clock samples")
    ap.add_argument("--one-stream", action="store_true",
                    help="run every pass with the batch on one stream (engine options two_streams / overlap = 0): "
                         "launches are serial, so a rocprofv3 --stats summary of this command agrees with roofline.avg_launch_ms")
    ap.add_argument("--force-dist", action="store_true",
                    help="test aid: run the N>1 code path (rendezvous, RCCL communicator, 7 all-reduces "
                         "per step) even with one rank")
    return ap.parse_args(argv)


def pmc_traffic(tokens_per_launch):
    """HBM bytes per k_main launch from the committed PMC passes (profiles/pmc_k_main.json, written by
    tools/pmc.sh on the GPU box: separate --pmc runs for FETCH_SIZE and WRITE_SIZE, KiB units,
    FETCH_SIZE doubled for 16-byte-per-lane streaming reads as MI355X_MICROARCH.md prescribes),
    scaled by tokens if the profiled batch differed.  NOT measured by this run.  None if absent."""
    path = os.path.join(REPO, PMC_FILE)
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        p = json.load(fh)
    return round(p["hbm_bytes_per_token"] * tokens_per_launch)


def cpu_baseline(w, shapes=((60, 500), (20, 200)), repeats=3):
    """Reference-op-order torch port on the host cores: per shape one warm-up at the SAME
    shape, then ``repeats`` timed forwards; the median is reported.  Headline shape first."""
    import torch
    from oracle import pf_oracle_torch
    from phyloformer_amd.msa_sim import simulate_batch
    # 32 threads: the fastest of 16/32/64/256 on the 2 x 64-core GPU host (9.9 s vs 59.6 s with all
    # 256 hardware threads, where the OpenMP pool oversubscribes) - tests/dev/cpu_threads.py
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    legs = []
    for n_seqs, n_sites in shapes:
        idx = simulate_batch(1, n_seqs, n_sites, seed=3)[0]
        pf_oracle_torch.forward(w.tensors, idx)                       # warm-up, same shape
        times = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            pf_oracle_torch.forward(w.tensors, idx)
            times.append(time.perf_counter() - t0)
        med = float(np.median(times))
        legs.append({"shape": f"{n_seqs}x{n_sites}", "value": round(1.0 / med, 5), "unit": "alignments/s",
                     "median_s": round(med, 4), "times_s": [round(t, 4) for t in times]})
    head = legs[0]
    return {"value": head["value"], "unit": "alignments/s", "cores": cores, "kind": "port",
            "sample": f"median of {repeats} timed forwards (after 1 warm-up at the same shape) of one "
                      f"{head['shape']} alignment, torch CPU ops in the reference's op order "
                      f"({head['median_s']:.2f} s each), {torch.get_num_threads()} threads",
            "legs": legs}


class PowerSampler:
    """rocm-smi in a side thread (a separate process per sample: nothing touches the GPU queue)."""

    def __init__(self, period=0.25):
        self.samples, self._stop, self._period = [], threading.Event(), period
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        while not self._stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True,
                                     timeout=5).stdout
                pw = re.search(r"Power \(W\): ([0-9.]+)", out)
                ck = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
                if pw and ck:
                    self.samples.append((float(pw.group(1)), int(ck.group(1))))
            except Exception:  # noqa: BLE001 - evidence only, never fails the bench
                return
            self._stop.wait(self._period)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=6)

    def summary(self):
        busy = [s for s in self.samples if s[1] > 500]
        if not busy:
            return None
        pw = sorted(s[0] for s in busy)
        ck = sorted(s[1] for s in busy)
        return {"samples": len(busy), "median_w": pw[len(pw) // 2], "median_sclk_mhz": ck[len(ck) // 2],
                "cap_w": 1400, "max_sclk_mhz": 2400, "source": "rocm-smi during the timed regions"}


def run(args, rank, world, local_rank, group, make_engine, weights, out=sys.stdout):
    """The rank logic of the benchmark.  ``group``: TcpGroup (or None when world == 1 and no --force-dist);
    ``make_engine(device)`` returns an object with the Engine interface (tests pass a fake)."""
    from phyloformer_amd import dist as pfdist
    from phyloformer_amd.msa_sim import simulate_batch

    comm_note = None
    eng = make_engine(int(os.environ.get("PF_BENCH_DEVICE", local_rank)))
    comm = None
    if args.force_dist and world == 1:
        eng.set_option("force_rccl", 1)
        eng.comm_init(eng.unique_id(), 0, 1)
        comm = eng.comm_info()
    elif world > 1 and args.shard == "sites":
        # every rank must agree on whether the RCCL communicator came up: if it did not on ANY rank, all of
        # them tear theirs down and fall back to sharding whole alignments (no collective), so the scaling
        # run still measures something, and the line says so
        ok, why = 1, ""
        try:
            pfdist.init_engine_comm(eng, group)
            comm = eng.comm_info()
        except Exception as exc:  # noqa: BLE001
            ok, why = 0, f"{type(exc).__name__}: {exc}"
        reasons = group.allgather((ok, why))
        if not all(o for o, _ in reasons):
            eng.comm_destroy()              # pf_forward* must not see a half-built communicator
            why = next(wy for o, wy in reasons if not o)
            if rank == 0:
                print(f"bench: RCCL communicator unavailable ({why}); falling back to --shard alignments",
                      file=sys.stderr)
            args.shard = "alignments"
            comm, comm_note = None, "site-sharding unavailable (RCCL init failed), alignments sharded instead"

    N, L = args.n_seqs, args.n_sites
    P = N * (N - 1) // 2
    if args.shard == "sites":
        B = args.batch * world                      # global batch, every alignment split over ranks
        lo, hi = pfdist.site_range(L, world, rank)
        idx = simulate_batch(min(B, 8), N, L, seed=3)
        idx = idx[np.arange(B) % idx.shape[0]][:, :, lo:hi]
    else:
        B = args.batch
        lo, hi = 0, L
        idx = simulate_batch(min(B, 8), N, L, seed=3 + rank)
        idx = idx[np.arange(B) % idx.shape[0]]
    idx = np.ascontiguousarray(idx)
    d_idx = eng.malloc(idx.nbytes)
    d_out = eng.malloc(B * P * 4)
    eng.h2d(d_idx, idx)

    def step():
        if args.shard == "sites":
            eng.forward_sharded_device(d_idx, B, N, lo, hi, L, d_out)
        else:
            eng.forward_device(d_idx, B, N, L, d_out)

    def step_host():
        if args.shard == "sites":
            return eng.forward_sharded(idx, lo, hi, L)
        return eng.forward(idx)

    def barrier():
        eng.synchronize()
        if group is not None:
            group.barrier()

    def timed(fn, steps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        dt = time.perf_counter() - t0
        return group.allreduce_max(dt) if group is not None else dt

    def streams(two):
        # the default schedule runs a batch as two half-batches on two streams (single GPU: each half fills the
        # other's kernel tails; site-sharded: a half's all-reduce runs under the other half's kernels)
        eng.set_option("two_streams", 1 if two else 0)
        eng.set_option("overlap", 1 if two else 0)

    streams(not args.one_stream)
    for _ in range(args.warmup):
        step()
    sampler = PowerSampler() if (rank == 0 and not args.no_power) else None
    if sampler:
        sampler.__enter__()
    dt = timed(step, args.steps)
    # The roofline of the dominant kernel is taken in a second timed region of the same length with the batch
    # on ONE stream: there a k_main launch has the chip to itself and covers the whole batch, so its HIP-event
    # duration is the kernel's own.  (In the two-stream schedule a half-batch launch shares the CUs with the
    # other half's column statistics and its bracket measures that mix.)
    prof, dt_one = {}, None
    if not args.no_profile:
        streams(False)
        step()
        eng.set_option("profile", 2)   # HIP events around every k_main launch only (see header)
        eng.profile_reset()
        dt_one = timed(step, args.steps)
        prof["main"] = eng.profile_get("main")
        eng.set_option("profile", 0)
        streams(not args.one_stream)
    result = np.empty((B, P), np.float32)
    eng.d2h(result, d_out)
    assert np.isfinite(result).all() and (result > 0).all()
    # the same step with host buffers: H2D + forward + D2H + synchronisation per call
    step_host()
    dt_host = timed(step_host, args.steps)
    if sampler:
        sampler.__exit__(None, None, None)
    info = eng.device_info()

    total_alignments = (B if args.shard == "sites" else B * world) * args.steps
    value = total_alignments / dt
    if rank == 0:
        tokens_per_launch = B * P * (hi - lo)
        roof = None
        if prof.get("main", (0, 0))[0]:
            n_main, ms_main = prof["main"]
            avg_s = ms_main / n_main * 1e-3
            nb = weights.n_blocks
            flops = tokens_per_launch * ((nb - 1) * FLOPS_MAIN_MID + FLOPS_MAIN_LAST) / nb
            ach = flops / avg_s / 1e12
            roof = {"bound": "mfma", "kernel": "k_main", "achieved": round(ach, 2),
                    "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                    "traffic": pmc_traffic(tokens_per_launch),
                    "traffic_source": f"{PMC_FILE}: separate rocprofv3 --pmc passes of this build (tools/pmc.sh), "
                                      "scaled by tokens; not measured by this run",
                    "avg_launch_ms": round(avg_s * 1e3, 4), "launches": n_main,
                    "schedule": "second timed region of the same steps with the batch on one stream (a launch covers "
                                "the whole batch and has the chip to itself); `value` is the default two-stream schedule"
                                if not args.one_stream else "one stream (--one-stream)",
                    "note": "algorithmic flops (1 pass); the split-bf16 scheme issues 3 MFMA passes, "
                            "so frac tops out at 1/3"}
        line = {
            "metric": "alignments/sec, 60-leaf/500-site LG+GC-like MSAs, Phyloformer forward (pf.ckpt)",
            "value": round(value, 3), "unit": "alignments/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16x3-split MFMA, fp32 accumulate/residual", "data": "synthetic",
            "config": {"workload": f"configs[2]: {N}-leaf/{L}-site LG+GC-like MSAs, pf.ckpt",
                       "global_batch": B if args.shard == "sites" else B * world,
                       "n_seqs": N, "n_sites": L, "parallelism": f"{args.shard}-sharded x{world}",
                       "device": info["name"].strip()},
            "value_one_stream": round(total_alignments / dt_one, 3) if dt_one else None,
            "value_host_buffers": round(total_alignments / dt_host, 3),
            "value_host_buffers_note": "same steps through pf_forward[_sharded] with host buffers: H2D of the "
                                       "indices + D2H of the distances + one synchronisation per call "
                                       "(PCIe-inclusive, SURVEY.md 8d); `value` has the indices resident in HBM",
            "kernel_ms": {k: round(v[1], 3) for k, v in prof.items()} if prof else None,
            "roofline": roof,
            "power": sampler.summary() if sampler else None,
            "cpu_baseline": None,
        }
        if comm:
            line["config"]["rccl"] = comm
        if comm_note:
            line["config"]["note"] = comm_note
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(weights)
            line["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), file=out, flush=True)
    eng.free(d_idx)
    eng.free(d_out)
    eng.close()
    if group is not None:
        group.barrier()
    return value


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    from phyloformer_amd.engine import Engine
    from phyloformer_amd.rendezvous import TcpGroup
    from phyloformer_amd.weights import load_weights

    group = TcpGroup(rank, world) if (world > 1 or args.force_dist) else None
    w = load_weights(args.ckpt)
    try:
        run(args, rank, world, local_rank, group, lambda device: Engine(w, device=device), w)
    finally:
        if group is not None:
            group.close()


if __name__ == "__main__":
    main()
