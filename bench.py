#!/usr/bin/env python3
"""Headline benchmark: alignments/s of the Phyloformer forward on synthetic LG+GC-like MSAs.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 *without* a launcher environment (``WORLD_SIZE`` unset) makes this process
a launcher: before any HIP call and without importing the engine it picks a free ``MASTER_PORT``, starts N
fresh children of itself with ``RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / PF_RUN_ID`` set,
relays rank 0's JSON line and exits non-zero if a child fails or the watchdog (``--launch-timeout``) fires.
Under ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` the environment is already
there and every process is a rank.

A *step* is one pass of the hot path (``pf_forward_device`` / ``pf_forward_sharded_device``) over one batch of
synthetic alignments whose residue indices are already resident in HBM.  Default workload: BASELINE.json
configs[2], the headline 60-leaf / 500-site shape, ``pf.ckpt`` weights (``--n-seqs / --n-sites`` select
another; the line's ``metric`` and ``config.workload`` are derived from the shape that ran).

N > 1 (default ``--shard sites``): the global batch is ``batch x N`` alignments and every alignment is
*site-sharded* over the N ranks (rank r holds L/N sites of every pair); row-attention statistics are
all-reduced once per block and the site sums once at the end with RCCL (n_blocks + 1 = 7 collectives per
forward, issued per half-batch on two streams / two communicators: 14 per step).  Per-GPU work is fixed as N
grows -> "weak".  ``--shard alignments`` shards whole alignments instead (no collective).  The ranks meet
through ``phyloformer_amd.rendezvous.TcpGroup`` (standard library, no torch import in a GPU rank).

Rank 0 prints ONE JSON line (contract in the task statement).  ``value`` is the rate with the indices resident
in HBM when the timed region starts (the task statement: the PCIe-inclusive rate "is never `value`");
``value_pcie_inclusive`` is the same step through ``pf_forward`` with host buffers (H2D of the indices, D2H of
the distances, one synchronisation per call) - the rate SURVEY.md 8d words its metric on.  Extra objects:
``roofline`` (dominant kernel ``k_main``: algorithmic flops / HIP-event time vs the dense bf16 MFMA peak),
``configs`` (N = 1: the other BASELINE configurations, a few hundred ms each), ``cpu_baseline`` (torch
op-order port of the reference on this host's cores; N = 1, rank 0 only) and ``power`` (socket power / clock
/ energy read in-process from librocm_smi64 during the timed regions).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time
import uuid

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16
HBM_PEAK_TBS = 8.0
# algorithmic flops per token per k_main launch (MAC = 2 flops):
#   FFN 64->256->64 = 65,536; column out_proj 64x64 = 8,192; next block's row v/q/k
#   projection 72x64 = 9,216; row mix apply (4 heads + bias) x 64 = 640.  The last block
#   swaps the row projection for the 64->1 head (128).
FLOPS_MAIN_MID = 65536 + 8192 + 9216 + 640
FLOPS_MAIN_LAST = 65536 + 8192 + 640 + 128
FLOPS_ALG_PER_TOKEN = 602240           # whole forward, SURVEY.md §8d
BYTES_ALG_PER_TOKEN = 3328             # whole forward, SURVEY.md §8d
PMC_FILE = os.path.join("profiles", "pmc_k_main.json")

# BASELINE.json configs by shape
WORKLOADS = {(20, 200): "configs[1]", (60, 500): "configs[2]", (60, 2000): "configs[3]", (200, 500): "configs[4]"}


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
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configurations (N = 1 leg)")
    ap.add_argument("--no-profile", action="store_true", help="skip HIP-event bracketing of kernels")
    ap.add_argument("--no-power", action="store_true", help="skip the power / clock samples")
    ap.add_argument("--one-stream", action="store_true",
                    help="run every pass with the batch on one stream (engine options two_streams / overlap = 0): "
                         "launches are serial, so a rocprofv3 --stats summary of this command agrees with roofline.avg_launch_ms")
    ap.add_argument("--force-dist", action="store_true",
                    help="test aid: run the N>1 code path (rendezvous, RCCL communicators, 14 all-reduces "
                         "per step) even with one rank")
    ap.add_argument("--reserve-cus", type=int, default=8,
                    help="CUs the persistent kernels leave to the RCCL kernels while collectives run")
    ap.add_argument("--launch-timeout", type=float, default=300.0,
                    help="self-launch (--gpus N > 1 without a launcher): seconds before the ranks are killed")
    return ap.parse_args(argv)


def workload_label(n_seqs, n_sites, ckpt):
    """(metric, config.workload) of the shape that actually runs."""
    name = os.path.basename(ckpt)
    tag = WORKLOADS.get((n_seqs, n_sites))
    shape = f"{n_seqs}-leaf/{n_sites}-site LG+GC-like MSAs"
    metric = f"alignments/sec, {shape}, Phyloformer forward ({name})"
    return metric, (f"{tag}: " if tag else "not a BASELINE shape: ") + f"{shape}, {name}"


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
    host_cores = os.cpu_count() or 1
    threads = min(host_cores, 32)
    torch.set_num_threads(threads)
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
    return {"value": head["value"], "unit": "alignments/s", "cores": threads, "threads": threads,
            "host_cores": host_cores, "kind": "port",
            "sample": f"median of {repeats} timed forwards (after 1 warm-up at the same shape) of one "
                      f"{head['shape']} alignment, torch CPU ops in the reference's op order "
                      f"({head['median_s']:.2f} s each), {torch.get_num_threads()} threads of a host with "
                      f"{host_cores} hardware threads",
            "legs": legs}


class PowerSampler:
    """Socket power, shader clock and the energy counter, read in-process (phyloformer_amd/smi.py: ctypes on
    librocm_smi64, i.e. sysfs reads; no subprocess, nothing touches the GPU queue).  Evidence only: any failure
    turns the sampler off, never the bench."""

    def __init__(self, period=0.25):
        self.samples, self._stop, self._period = [], threading.Event(), period
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._smi, self.joules, self.seconds = None, None, None

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append((self._smi.power_w(), self._smi.sclk_mhz()))
            except Exception:  # noqa: BLE001
                return
            self._stop.wait(self._period)

    def __enter__(self):
        try:
            from phyloformer_amd.smi import Smi
            self._smi = Smi(int(os.environ.get("PF_SMI_DEVICE", "0")))
            self._j0, self._t0 = self._smi.energy_j(), time.perf_counter()
            self._thread.start()
        except Exception:  # noqa: BLE001
            self._smi = None
        return self

    def __exit__(self, *exc):
        if self._smi is None:
            return
        self._stop.set()
        self._thread.join(timeout=3)
        try:
            self.joules, self.seconds = self._smi.energy_j() - self._j0, time.perf_counter() - self._t0
            self._smi.close()
        except Exception:  # noqa: BLE001
            pass

    def summary(self):
        busy = [s for s in self.samples if s[1] > 500]
        if not busy:
            return None
        pw = sorted(s[0] for s in busy)
        ck = sorted(s[1] for s in busy)
        out = {"samples": len(busy), "median_w": round(pw[len(pw) // 2], 1), "median_sclk_mhz": round(ck[len(ck) // 2]),
               "cap_w": 1400, "max_sclk_mhz": 2400, "source": "librocm_smi64 in-process during the timed regions"}
        if self.joules is not None and self.seconds:
            out["avg_w_energy_counter"] = round(self.joules / self.seconds, 1)
        return out


def time_config(eng, n, l, B, gaps=False, budget_s=0.6):
    """One BASELINE configuration on an existing engine: device-resident rate, a few hundred ms of timed work."""
    from phyloformer_amd.msa_sim import simulate_batch
    base = simulate_batch(min(B, 4), n, l, seed=2, gaps=gaps)
    idx = np.ascontiguousarray(base[np.arange(B) % base.shape[0]])
    P = n * (n - 1) // 2
    d_idx, d_out = eng.malloc(idx.nbytes), eng.malloc(B * P * 4)
    eng.h2d(d_idx, idx)
    t0 = time.perf_counter()
    for _ in range(2):
        eng.forward_device(d_idx, B, n, l, d_out)
    eng.synchronize()
    per = (time.perf_counter() - t0) / 2
    reps = max(3, int(budget_s / max(per, 1e-5)))
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.forward_device(d_idx, B, n, l, d_out)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / reps
    eng.free(d_idx)
    eng.free(d_out)
    tok = B * P * l
    return {"n_seqs": n, "n_sites": l, "gapped": gaps, "batch": B, "timed_forwards": reps,
            "ms_per_step": round(dt * 1e3, 4), "alignments_per_s": round(B / dt, 2),
            "frac_mfma": round(FLOPS_ALG_PER_TOKEN * tok / dt / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
            "frac_hbm": round(BYTES_ALG_PER_TOKEN * tok / dt / 1e12 / HBM_PEAK_TBS, 4)}


def configs_leg(eng, make_engine_for, device):
    """Every other BASELINE configuration (N = 1, after the headline region; a few seconds in total):
    whole-forward roofline fractions of SURVEY.md 8d (602,240 flop and 3,328 B per token against 2.5 PFLOP/s
    and 8 TB/s; the split scheme issues three MFMA passes, so frac_mfma tops out at 1/3)."""
    out = {}
    out["configs[1] 20x200 batch 64, pf.ckpt"] = time_config(eng, 20, 200, 64)
    out["configs[1] 20x200 batch 1, pf.ckpt"] = time_config(eng, 20, 200, 1, budget_s=0.3)
    out["configs[2] 60x500 batch 1, pf.ckpt"] = time_config(eng, 60, 500, 1, budget_s=0.3)
    out["configs[3]-shape 60x2000 batch 4 on one GPU, pf.ckpt"] = time_config(eng, 60, 2000, 4)
    indel = os.path.join(REPO, "models", "pf_indel.ckpt")
    if os.path.exists(indel):
        e2 = make_engine_for(indel, device)
        try:
            out["configs[4] 200x500 gapped batch 2, pf_indel.ckpt"] = time_config(e2, 200, 500, 2, gaps=True)
        finally:
            e2.close()
    return out


def run(args, rank, world, local_rank, group, make_engine, weights, out=sys.stdout, make_engine_for=None):
    """The rank logic of the benchmark.  ``group``: TcpGroup (or None when world == 1 and no --force-dist);
    ``make_engine(device)`` returns an object with the Engine interface (tests pass a fake);
    ``make_engine_for(ckpt, device)`` builds an engine for another checkpoint (the ``configs`` leg)."""
    from phyloformer_amd import dist as pfdist
    from phyloformer_amd.msa_sim import simulate_batch

    comm_note = None
    device = int(os.environ.get("PF_BENCH_DEVICE", local_rank))
    try:
        eng = make_engine(device)
    except ValueError as exc:
        # more ranks than devices (e.g. `--gpus 2` tried on a 1-GPU box): share the devices round-robin instead of
        # dying - RCCL will refuse two ranks on one device, and the ranks then agree to shard whole alignments
        import re
        m = re.search(r"device \d+ out of range \(have (\d+)\)", str(exc))
        if not m or int(m.group(1)) < 1:
            raise
        device = local_rank % int(m.group(1))
        comm_note = f"{world} ranks on {m.group(1)} device(s): rank {rank} shares device {device}"
        print(f"bench: {comm_note}", file=sys.stderr)
        eng = make_engine(device)
    comm = None
    if world > 1 or args.force_dist:
        eng.set_option("reserve_cus", args.reserve_cus)
    if args.force_dist and world == 1:
        eng.set_option("force_rccl", 1)
        eng.comm_init(eng.unique_id(), 0, 1)
        comm = eng.comm_info()
    elif world > 1 and args.shard == "sites":
        # every rank must agree on whether the RCCL communicators came up: if they did not on ANY rank, all of
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
            comm = None
            comm_note = ((comm_note + "; ") if comm_note else "") + \
                "site-sharding unavailable (RCCL init failed), alignments sharded instead"

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
    coll0 = eng.collective_count()
    dt = timed(step, args.steps)
    collectives_per_step = (eng.collective_count() - coll0) / max(args.steps, 1)
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
        metric, workload = workload_label(N, L, args.ckpt)
        line = {
            "metric": metric,
            "value": round(value, 3), "unit": "alignments/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16x3-split MFMA, fp32 accumulate/residual", "data": "synthetic",
            "config": {"workload": workload,
                       "global_batch": B if args.shard == "sites" else B * world,
                       "n_seqs": N, "n_sites": L, "parallelism": f"{args.shard}-sharded x{world}",
                       "device": info["name"].strip(),
                       "n_ranks_in_comm": world if comm else 0,
                       "communicators": 2 if comm else 0,
                       "collectives_per_step": round(collectives_per_step, 2),
                       "reserve_cus": args.reserve_cus if comm else None,
                       "rccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS") if comm else None},
            "value_definition": "indices resident in HBM when the timed region starts (task statement, "
                                "Measurement: the PCIe-inclusive rate is never `value`); value_pcie_inclusive is "
                                "the rate SURVEY.md 8d words its metric on",
            "value_pcie_inclusive": round(total_alignments / dt_host, 3),
            "value_one_stream": round(total_alignments / dt_one, 3) if dt_one else None,
            "value_host_buffers": round(total_alignments / dt_host, 3),
            "value_host_buffers_note": "same steps through pf_forward[_sharded] with host buffers: H2D of the "
                                       "indices + D2H of the distances + one synchronisation per call "
                                       "(PCIe-inclusive, SURVEY.md 8d); `value` has the indices resident in HBM",
            "kernel_ms": {k: round(v[1], 3) for k, v in prof.items()} if prof else None,
            "roofline": roof,
            "power": sampler.summary() if sampler else None,
            "configs": None,
            "cpu_baseline": None,
        }
        if comm:
            line["config"]["rccl"] = comm
        if comm_note:
            line["config"]["note"] = comm_note
        if world == 1 and not args.no_configs and make_engine_for is not None:
            eng.free(d_idx)
            eng.free(d_out)
            d_idx = d_out = None
            streams(True)
            line["configs"] = configs_leg(eng, make_engine_for, device)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(weights)
            line["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), file=out, flush=True)
    if d_idx is not None:
        eng.free(d_idx)
        eng.free(d_out)
    eng.close()
    if group is not None:
        group.barrier()
    return value


# ---- self-launch: `python3 bench.py --gpus N` without a launcher -------------------------------------------
def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """Parent of N ranks.  Touches neither HIP nor the engine: it only starts fresh children of this script,
    one per GPU, and waits.  Rank 0's stdout (the JSON line) is relayed; every child's stderr is inherited.
    Returns the exit code: 0 only if every rank exited 0 within ``--launch-timeout`` seconds."""
    world = args.gpus
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()),
                "PF_RUN_ID": uuid.uuid4().hex, "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    deadline = time.monotonic() + args.launch_timeout
    rc, why = 0, ""
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while True:
        codes = [p.poll() for p in procs]
        if any(c not in (None, 0) for c in codes):
            # a rank that dies usually takes its peers down with it (they lose the rendezvous connection):
            # give them a moment, then name every rank that failed, not just the first one seen
            time.sleep(0.5)
            bad = [(r, p.poll()) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
            rc, why = 1, ", ".join(f"rank {r} exited with code {c}" for r, c in bad)
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            rc, why = 124, f"watchdog: ranks still running after {args.launch_timeout:.0f} s"
            break
        time.sleep(0.05)
    if rc:
        for p in procs:                 # exactly the processes started above, by PID
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 5
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        print(f"bench: {why}; all {world} ranks stopped", file=sys.stderr)
    reader.join(timeout=5)
    text = (out0[0] if out0 else b"").decode(errors="replace")
    if text:
        sys.stdout.write(text)
        sys.stdout.flush()
    if rc == 0 and not text.strip():
        print("bench: rank 0 printed nothing", file=sys.stderr)
        rc = 1
    return rc


def engine_factories(ckpt):
    """(weights, make_engine(device), make_engine_for(ckpt, device)).  PF_BENCH_ENGINE_FACTORY=module:function
    swaps in a stand-in for the CPU test of the self-launch path: ``function(device)`` returns an object with the
    Engine interface and ``module.bench_weights()`` one with ``n_blocks``."""
    hook = os.environ.get("PF_BENCH_ENGINE_FACTORY")
    if hook:
        import importlib
        mod_name, fn = hook.split(":")
        mod = importlib.import_module(mod_name)
        return mod.bench_weights(), getattr(mod, fn), None
    from phyloformer_amd.engine import Engine
    from phyloformer_amd.weights import load_weights
    w = load_weights(ckpt)
    return w, (lambda device: Engine(w, device=device)), (lambda c, device: Engine(load_weights(c), device=device))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args, argv)          # before anything touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    if world > 1 or args.force_dist:
        # The persistent kernels leave `reserve_cus` CUs to RCCL while collectives run; a collective that wants
        # more channels (= workgroups) than that would wait for a whole k_main launch of the other half-batch to
        # drain.  Unless the user says otherwise, RCCL is therefore told to use at most that many channels
        # (read by librccl when it is first loaded, i.e. before the engine resolves it).
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(max(1, args.reserve_cus)))

    # Native libraries write to file descriptor 1 as they please (RCCL prints a version banner there when a
    # communicator fails): the contract is ONE JSON line on stdout, so descriptor 1 is pointed at stderr for the
    # life of the rank and the line goes to a private copy of the original stdout.
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    from phyloformer_amd.rendezvous import TcpGroup
    w, make_engine, make_engine_for = engine_factories(args.ckpt)
    group = TcpGroup(rank, world) if (world > 1 or args.force_dist) else None
    try:
        run(args, rank, world, local_rank, group, make_engine, w, out=line_out, make_engine_for=make_engine_for)
    finally:
        if group is not None:
            group.close()
        line_out.flush()
    return 0


if __name__ == "__main__":
    sys.exit(main())
