#!/usr/bin/env python3
"""Headline benchmark: alignments/s of the Phyloformer forward on synthetic LG+GC-like MSAs.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A *step* is one pass of the hot path (``pf_forward_device`` /
``pf_forward_sharded_device``) over one batch of synthetic alignments whose residue
indices are already resident in HBM.  Workload: BASELINE.json configs[2], the
headline 60-leaf / 500-site shape, ``pf.ckpt`` weights.

N > 1 (default ``--shard sites``): the global batch is ``batch x N`` alignments and
every alignment is *site-sharded* over the N ranks (rank r holds 500/N sites of
every pair); row-attention statistics are all-reduced once per block and the site
sums once at the end with RCCL (7 collectives per step).  Per-GPU work is fixed as
N grows → "weak".  ``--shard alignments`` shards whole alignments instead (no collective).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
``roofline`` (dominant kernel ``k_main``: algorithmic flops / HIP-event time, vs the
dense bf16 MFMA peak) and ``cpu_baseline`` (the torch op-order port of the reference
timed on this host's cores; N = 1, rank 0 only).
"""
import argparse
import json
import os
import sys
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


def parse_args():
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
    ap.add_argument("--force-dist", action="store_true",
                    help="test aid: run the N>1 code path (gloo rendezvous, RCCL communicator, 7 all-reduces "
                         "per step) even with one rank")
    return ap.parse_args()


def pmc_traffic(tokens_per_launch):
    """HBM bytes per k_main launch from the committed PMC passes (profiles/pmc_k_main.json, written by
    tools/pmc.sh on the GPU box: separate --pmc runs for FETCH_SIZE and WRITE_SIZE, KiB units,
    FETCH_SIZE doubled for 16-byte-per-lane streaming reads as MI355X_MICROARCH.md prescribes),
    scaled by tokens if the profiled batch differed.  None if no PMC file is present."""
    path = os.path.join(REPO, "profiles", "pmc_k_main.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        p = json.load(fh)
    return round(p["hbm_bytes_per_token"] * tokens_per_launch)


def cpu_baseline(w, n_seqs, n_sites):
    """Reference-op-order torch port on the host cores: 1 warm-up (small) + 1 timed forward."""
    import torch
    from oracle import pf_oracle_torch
    from phyloformer_amd.msa_sim import simulate_batch
    # 32 threads: the fastest of 16/32/64/256 on the 2 x 64-core GPU host (9.9 s vs 59.6 s with all
    # 256 hardware threads, where the OpenMP pool oversubscribes) - tests/dev/cpu_threads.py
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    pf_oracle_torch.forward(w.tensors, simulate_batch(1, 20, 100, seed=9)[0])   # thread-pool warm-up
    idx = simulate_batch(1, n_seqs, n_sites, seed=3)[0]
    t0 = time.perf_counter()
    pf_oracle_torch.forward(w.tensors, idx)
    dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 5), "unit": "alignments/s", "cores": cores, "kind": "port",
            "sample": f"1 forward of one {n_seqs}x{n_sites} alignment, torch CPU ops in the "
                      f"reference's op order ({dt:.1f} s), {torch.get_num_threads()} threads"}


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
    from phyloformer_amd.msa_sim import simulate_batch
    from phyloformer_amd.weights import load_weights
    from phyloformer_amd import dist as pfdist

    comm_note = None
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    w = load_weights(args.ckpt)
    # PF_BENCH_DEVICE: test aid - put every rank on one device (a 1-GPU box can then exercise the N > 1
    # launch path; RCCL refuses two ranks on one GPU, which also exercises the fallback below)
    eng = Engine(w, device=int(os.environ.get("PF_BENCH_DEVICE", local_rank)))
    if args.force_dist and world == 1:
        eng.set_option("force_rccl", 1)
        uid = pfdist.broadcast_bytes(eng.unique_id(), 128, src=0)
        eng.comm_init(uid, 0, 1)
    elif world > 1 and args.shard == "sites":
        # every rank must agree on whether the RCCL communicator came up: if it did not (librccl missing,
        # init error) on any rank, all of them fall back to sharding whole alignments (no collective) so
        # the scaling run still measures something, and the line says so
        import torch
        ok, why = 1, ""
        try:
            pfdist.init_engine_comm(eng)
        except Exception as exc:  # noqa: BLE001
            ok, why = 0, f"{type(exc).__name__}: {exc}"
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if rank == 0:
                print(f"bench: RCCL communicator unavailable ({why or 'failed on another rank'}); "
                      "falling back to --shard alignments", file=sys.stderr)
            args.shard = "alignments"
            comm_note = "site-sharding unavailable (RCCL init failed), alignments sharded instead"

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

    def barrier():
        eng.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    if not args.no_profile:
        eng.set_option("profile", 2)   # HIP events around every k_main launch only (see header)
        eng.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    prof = {}
    if not args.no_profile:
        prof["main"] = eng.profile_get("main")
        eng.set_option("profile", 0)
    out = np.empty((B, P), np.float32)
    eng.d2h(out, d_out)
    assert np.isfinite(out).all() and (out > 0).all()
    info = eng.device_info()

    total_alignments = (B if args.shard == "sites" else B * world) * args.steps
    value = total_alignments / dt
    if rank == 0:
        tokens_per_launch = B * P * (hi - lo)
        roof = None
        if prof.get("main", (0, 0))[0]:
            n_main, ms_main = prof["main"]
            avg_s = ms_main / n_main * 1e-3
            nb = w.n_blocks
            flops = tokens_per_launch * ((nb - 1) * FLOPS_MAIN_MID + FLOPS_MAIN_LAST) / nb
            ach = flops / avg_s / 1e12
            roof = {"bound": "mfma", "kernel": "k_main", "achieved": round(ach, 2),
                    "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                    "traffic": pmc_traffic(tokens_per_launch),
                    "avg_launch_ms": round(avg_s * 1e3, 4), "launches": n_main,
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
            "kernel_ms": {k: round(v[1], 3) for k, v in prof.items()} if prof else None,
            "roofline": roof,
            "cpu_baseline": None,
        }
        if comm_note:
            line["config"]["note"] = comm_note
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w, N, L)
            line["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    eng.free(d_idx)
    eng.free(d_out)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
