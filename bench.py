#!/usr/bin/env python3
"""Headline benchmark: alignments/s of the Phyloformer forward on synthetic LG+GC-like MSAs.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 *without* a launcher environment (``WORLD_SIZE`` unset) makes this process
a launcher: before any HIP call and without importing the engine it picks a free ``MASTER_PORT`` and starts N
fresh children of itself with ``RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / PF_RUN_ID`` set.
Those - like the N processes of ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``, the
other way to start an N > 1 run - do not become ranks: each *supervises* one rank, a fresh child of its own
(``supervise``), so that a stalled collective ends as a line from the next rung of the ladder below, not as a hang.

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

Fallback ladder: the supervisors run rung 1 = sites on two streams / two communicators; if a
rank dies, no rank makes progress (``--stall-timeout``) or the rung exceeds ``--rung-timeout``, those ranks are
killed and FRESH ones are started for rung 2 = ``--one-stream`` (serial collectives on one stream), then rung 3 =
``--shard alignments`` (no collective).  A rank is never re-exec'ed.  The line records ``config.rung`` and why
earlier rungs were abandoned.  Exit codes: 0 ok; 3 = the parity bound failed (the line is still printed, no
further rung is tried); 4 = more ranks than GPUs without ``--allow-shared-devices``; 124 = every rung timed out.

Parity (BASELINE's metric is "alignments/sec + max-abs distance error"): after the timed regions and outside them
every rank runs the SAME entry point it timed on the reference outputs committed under ``tests/golden/`` (data
written by oracle/gen_golden.py from the reference's CPU forward; not the oracle): configs[2] 60x500 always,
for N > 1 also configs[3] 60x2000 with L / N sites per rank over real RCCL (and its rate, as
``configs["configs[3] 60x2000 site-sharded xN"]``); at N = 1 all four BASELINE shapes.  The line carries
``max_abs_err`` (worst case), ``max_abs_err_ok`` (<= 1e-4), ``ranks_bit_identical`` (CRC of every rank's result)
and the per-case ``parity`` object.

Rank 0 prints ONE JSON line (contract in the task statement).  ``value`` is the rate with the indices resident
in HBM when the timed region starts (the task statement: the PCIe-inclusive rate "is never `value`");
``value_pcie_inclusive`` is the same step through ``pf_forward`` with host buffers (H2D of the indices, D2H of
the distances, one synchronisation per call) - the rate SURVEY.md 8d words its metric on.  Extra objects:
``roofline`` (dominant kernel ``k_main``: algorithmic flops / HIP-event time vs the dense bf16 / fp16 MFMA peak),
``configs`` (N = 1: the other BASELINE configurations, a few hundred ms each), ``cpu_baseline`` (torch
op-order port of the reference on this host's cores; N = 1, rank 0 only) and ``power`` (socket power / clock
/ energy read in-process from librocm_smi64 during the timed regions).
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import threading
import time
import uuid
import zlib

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16 / fp16 (the same rate)
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

# Parity: the reference's own outputs (tests/golden/*.npz, written by oracle/gen_golden.py from the reference's
# CPU forward - data, not the oracle) for every BASELINE shape.  name -> (file, key prefix, checkpoint)
PARITY_BOUND = 1e-4        # north star: max-abs distance error vs the reference CPU forward
GOLDENS = {
    "configs[1] 20x200 x3": ("configs.npz", "c2", "pf.ckpt"),
    "configs[2] 60x500": ("configs.npz", "c3", "pf.ckpt"),
    "configs[3] 60x2000": ("configs_big.npz", "c4", "pf.ckpt"),
    "configs[4] 200x500 gapped": ("configs_big.npz", "c5", "pf_indel.ckpt"),
}
EXIT_PARITY, EXIT_DEVICES, EXIT_WATCHDOG = 3, 4, 124


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
    ap.add_argument("--rccl-max-nchannels", type=int, default=0,
                    help="opt-in: set NCCL_MAX_NCHANNELS for the ranks (0 = leave RCCL's default; never measured at N > 1)")
    ap.add_argument("--allow-shared-devices", action="store_true",
                    help="let several ranks share a GPU when there are more ranks than devices (the line is then "
                         "marked as not a scaling result); without it such a run exits with code 4")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity leg (max_abs_err vs the committed reference outputs)")
    ap.add_argument("--launch-timeout", type=float, default=400.0,
                    help="self-launch (--gpus N > 1 without a launcher): seconds for all rungs of the fallback ladder together")
    ap.add_argument("--rung-timeout", type=float, default=150.0, help="self-launch: seconds one rung of the ladder may take")
    ap.add_argument("--stall-timeout", type=float, default=75.0,
                    help="self-launch: a rung whose ranks report no progress for this long is abandoned")
    return ap.parse_args(argv)


def workload_label(n_seqs, n_sites, ckpt):
    """(metric, config.workload) of the shape that actually runs."""
    name = os.path.basename(ckpt)
    tag = WORKLOADS.get((n_seqs, n_sites))
    shape = f"{n_seqs}-leaf/{n_sites}-site LG+GC-like MSAs"
    metric = f"alignments/sec, {shape}, Phyloformer forward ({name})"
    return metric, (f"{tag}: " if tag else "not a BASELINE shape: ") + f"{shape}, {name}"


def pmc_traffic(tokens_per_launch, build=None):
    """(HBM bytes per k_main launch, where the figure comes from).  The bytes are those of the committed PMC passes
    (profiles/pmc_k_main.json, written by tools/pmc.sh on the GPU box: separate --pmc runs for FETCH_SIZE and
    WRITE_SIZE, KiB units, FETCH_SIZE doubled for 16-byte-per-lane streaming reads as MI355X_MICROARCH.md
    prescribes), scaled by tokens if the profiled batch differed - NOT measured by this run, so the file carries the
    ``kernel_hash`` of the library it was taken with (pf_build_info: k_main's source + the flags of its translation
    unit) and a library with another hash gets ``None``: stale counters are not reported as this build's."""
    path = os.path.join(REPO, PMC_FILE)
    if not os.path.exists(path):
        return None, f"{PMC_FILE} absent"
    with open(path) as fh:
        p = json.load(fh)
    have, want = p.get("kernel_hash"), (build or {}).get("kernel_hash")
    if want is None or have != want:
        return None, (f"{PMC_FILE} was taken with kernel_hash {have}, this library has {want}: stale counters are not "
                      "reported (re-run tools/pmc.sh)")
    return round(p["hbm_bytes_per_token"] * tokens_per_launch), (
        f"{PMC_FILE}: separate rocprofv3 --pmc passes of a library with this kernel_hash ({have}; tools/pmc.sh), "
        "scaled by tokens; not measured by this run")


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

    def __init__(self, period=0.25, pci=None):
        # pci = (domain, bus, device) of the engine's GPU: rocm_smi indexes in PCI order, HIP ordinals follow
        # HIP_VISIBLE_DEVICES - the sampler looks the device up by address ($PF_SMI_DEVICE forces an index)
        self._pci = pci
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
            forced = os.environ.get("PF_SMI_DEVICE")
            self._smi = Smi(int(forced)) if forced is not None else Smi(0, pci=self._pci)
            self._where = {"smi_index": self._smi.index, "pci": self._smi.bdf,
                           "matched_by": "PF_SMI_DEVICE" if forced is not None else ("pci address of the engine's device" if self._pci else "index 0")}
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
        out["device"] = getattr(self, "_where", None)
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


def configs_leg(eng, make_engine_for, device, parity=None):
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
            if parity is not None:
                c = parity_case(e2, "configs[4] 200x500 gapped", False, 0, 1, None)
                if c is not None:
                    parity["configs[4] 200x500 gapped"] = c
        finally:
            e2.close()
    return out


_LAST_TICK = [0.0]


def beat(stage, expect_s=None, budget_s=None):
    """Progress mark for the supervisors' watchdogs: one line per stage in $PF_BENCH_PROGRESS/rank<r>.
    ``expect_s``: how long the rank expects to stay silent after this mark (a timed region of K asynchronous steps
    ends in ONE synchronisation: K x the step time measured in the warm-up) - the stall watchdog allows twice that;
    ``budget_s``: the rank's estimate of the whole run, which stretches the per-rung and overall limits the same way
    (fixed 75 / 150 / 400 s limits declared a healthy run with many steps stalled)."""
    d = os.environ.get("PF_BENCH_PROGRESS")
    if not d:
        return
    _LAST_TICK[0] = time.monotonic()
    extra = (f" expect={expect_s:.1f}" if expect_s is not None else "") + (f" budget={budget_s:.1f}" if budget_s is not None else "")
    try:
        with open(os.path.join(d, "rank" + os.environ.get("RANK", "0")), "a") as fh:
            fh.write(f"{time.time():.3f} {stage.replace(' ', '_')}{extra}\n")
    except OSError:
        pass


def tick(stage, expect_s=None):
    """beat(), at most once a second: for loops."""
    if time.monotonic() - _LAST_TICK[0] >= 1.0:
        beat(stage, expect_s)


def load_golden(name):
    """(idx uint8[b][N][L], reference distances float32[b][P], checkpoint name) or None if the file is absent."""
    fname, key, ckpt = GOLDENS[name]
    path = os.path.join(REPO, "tests", "golden", fname)
    if not os.path.exists(path):
        return None
    with np.load(path) as g:
        return np.ascontiguousarray(g[key + "_idx"]), np.ascontiguousarray(g[key + "_dist"]), ckpt


def parity_case(eng, name, sharded, rank, world, group, copies=2):
    """One committed reference output through the entry point the bench timed: ``pf_forward_sharded_device`` with
    this rank's site range (real RCCL when the handle carries communicators) or ``pf_forward_device``.  The
    alignment is repeated to at least ``copies`` so that both half-batches - two streams, two communicators - are
    exercised; every copy is compared.  The ranks exchange (error, CRC32 of the result bytes) through the
    rendezvous: in a site-sharded run every rank holds the all-reduced result, and all must hold the same bits."""
    from phyloformer_amd import dist as pfdist
    got = load_golden(name)
    if got is None:
        return None
    idx, ref, _ = got
    reps = max(1, -(-copies // idx.shape[0]))
    idx, ref = np.ascontiguousarray(np.concatenate([idx] * reps)), np.concatenate([ref] * reps)
    B, N, L = idx.shape
    P = N * (N - 1) // 2
    lo, hi = pfdist.site_range(L, world, rank) if sharded else (0, L)
    local = np.ascontiguousarray(idx[:, :, lo:hi])
    d_idx, d_out = eng.malloc(max(local.nbytes, 1)), eng.malloc(B * P * 4)
    if local.nbytes:
        eng.h2d(d_idx, local)
    coll0 = eng.collective_count()
    if sharded:
        eng.forward_sharded_device(d_idx, B, N, lo, hi, L, d_out)
    else:
        eng.forward_device(d_idx, B, N, L, d_out)
    res = np.empty((B, P), np.float32)
    eng.d2h(res, d_out)
    ncoll = eng.collective_count() - coll0
    eng.free(d_idx)
    eng.free(d_out)
    finite = bool(np.isfinite(res).all())
    err = float(np.abs(res - ref).max()) if finite else 1e30
    every = group.allgather([err, zlib.crc32(res.tobytes())]) if group is not None else [[err, 0]]
    worst = max(e for e, _ in every)
    same = len({c for _, c in every}) == 1 if len(every) > 1 else None
    return {"max_abs_err": worst, "max_abs_ref": round(float(np.abs(ref).max()), 4), "alignments": B,
            "entry_point": "pf_forward_sharded_device" if sharded else "pf_forward_device",
            "sites_per_rank": hi - lo, "collectives": ncoll, "finite": finite, "ranks_bit_identical": same,
            "ok": bool(worst <= PARITY_BOUND and same is not False)}


def run(args, rank, world, local_rank, group, make_engine, weights, out=sys.stdout, make_engine_for=None):
    """The rank logic of the benchmark.  ``group``: TcpGroup (or None when world == 1 and no --force-dist);
    ``make_engine(device)`` returns an object with the Engine interface (tests pass a fake);
    ``make_engine_for(ckpt, device)`` builds an engine for another checkpoint (the ``configs`` leg)."""
    from phyloformer_amd import dist as pfdist
    from phyloformer_amd.msa_sim import simulate_batch

    comm_note = None
    beat("start")
    device = int(os.environ.get("PF_BENCH_DEVICE", local_rank))
    eng, have = None, None
    try:
        eng = make_engine(device)
    except ValueError as exc:
        import re
        m = re.search(r"device \d+ out of range \(have (\d+)\)", str(exc))
        if not m or int(m.group(1)) < 1 or world == 1:
            raise
        have = int(m.group(1))
    n_devices = world
    if world > 1 and group is not None:
        # One rank per GPU is the contract.  More ranks than devices (`--gpus 8` on a smaller box used to
        # emit a "weak scaling" number measured on shared GPUs) ends the run with its own exit code on EVERY rank -
        # unless --allow-shared-devices, and then the line says what it is: n_gpus = distinct devices, not ranks.
        # (identity = the PCI address when the engine has one: with a per-rank HIP_VISIBLE_DEVICES every rank's
        # ordinal is 0 for a different physical GPU - ADVICE r04)
        def ident(e, ordinal):
            try:
                return list(e.device_pci())
            except Exception:  # noqa: BLE001 - stand-in engines
                return ordinal
        seen = group.allgather(ident(eng, device) if eng is not None else -1)
        if any(d == -1 for d in seen) or len({json.dumps(d) for d in seen}) < world:
            if not args.allow_shared_devices:
                if eng is not None:
                    eng.close()
                if rank == 0:
                    print(f"bench: {world} ranks but not {world} distinct GPUs (devices asked for: {seen}, -1 = no such "
                          "device); one rank per GPU is the contract - pass --allow-shared-devices to share them "
                          "(the line is then marked as not a scaling result)", file=sys.stderr)
                raise SystemExit(EXIT_DEVICES)
            if eng is None:
                device = local_rank % have
                eng = make_engine(device)
            n_devices = len({json.dumps(d) for d in group.allgather(ident(eng, device))})
            comm_note = (f"{world} ranks share {n_devices} device(s) (--allow-shared-devices): NOT a scaling result; "
                         "RCCL refuses two ranks on one device, so whole alignments are sharded")
            if rank == 0:
                print(f"bench: {comm_note}", file=sys.stderr)
    if eng is None:
        raise SystemExit(f"bench: device {device} does not exist (have {have})")
    beat("engine")
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

    beat("communicators")
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

    est = {"step": None}          # seconds per step, measured in the warm-up (None: no warm-up)

    def timed(fn, steps, stage="timed region", scale=1.0):
        expect = est["step"] * steps * scale if est["step"] else None
        beat(stage + " begins", expect)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
            tick(stage, expect)      # (launches are asynchronous: the loop runs ahead of the GPU, the region ends in
        barrier()                    #  one synchronisation - hence the expectation published above)
        dt = time.perf_counter() - t0
        return group.allreduce_max(dt) if group is not None else dt

    def streams(two):
        # the default schedule runs a batch as two half-batches on two streams (single GPU: each half fills the
        # other's kernel tails; site-sharded: a half's all-reduce runs under the other half's kernels)
        eng.set_option("two_streams", 1 if two else 0)
        eng.set_option("overlap", 1 if two else 0)

    streams(not args.one_stream)
    t_w = time.perf_counter()
    if args.warmup:
        step()                       # first touch: workspaces, communicator channels
        eng.synchronize()
        beat("first step")
        if args.warmup > 1:
            t_w = time.perf_counter()
        for _ in range(args.warmup - 1):
            step()
            tick("warm-up")
        eng.synchronize()
    if args.warmup:
        est["step"] = (time.perf_counter() - t_w) / max(1, args.warmup - 1)
        # three regions of `steps` (two-stream, one-stream, host buffers) + parity / configs / baseline allowance
        beat("warm-up", budget_s=est["step"] * args.steps * 3.5 + 60.0)
    else:
        beat("warm-up")
    sampler = None
    if rank == 0 and not args.no_power:
        try:
            sampler = PowerSampler(pci=eng.device_pci() if hasattr(eng, "device_pci") else None)
        except Exception:  # noqa: BLE001 - evidence only
            sampler = PowerSampler()
    if sampler:
        sampler.__enter__()
    coll0 = eng.collective_count()
    dt = timed(step, args.steps)
    collectives_per_step = (eng.collective_count() - coll0) / max(args.steps, 1)
    beat("timed region")
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
        dt_one = timed(step, args.steps, "roofline region", 1.2)
        prof["main"] = eng.profile_get("main")
        eng.set_option("profile", 0)
        streams(not args.one_stream)
        beat("roofline region")
    result = np.empty((B, P), np.float32)
    eng.d2h(result, d_out)
    assert np.isfinite(result).all() and (result > 0).all()
    # the same step with host buffers: H2D + forward + D2H + synchronisation per call
    step_host()
    dt_host = timed(step_host, args.steps, "host-buffer region", 1.2)
    if sampler:
        sampler.__exit__(None, None, None)
    info = eng.device_info()
    beat("host-buffer region")

    total_alignments = (B if args.shard == "sites" else B * world) * args.steps
    value = total_alignments / dt
    multi = world > 1 or args.force_dist
    sharded = args.shard == "sites"          # the entry point the timed regions used
    how = f"{args.shard}-sharded x{world}"

    # ---- parity leg: the metric's second half, through the entry point that was timed (outside the timed regions)
    parity = None
    if not args.no_parity:
        parity = {}
        names = ["configs[2] 60x500", "configs[3] 60x2000"] if multi else \
                ["configs[1] 20x200 x3", "configs[2] 60x500", "configs[3] 60x2000"]
        for nm in names:
            if GOLDENS[nm][2] != os.path.basename(args.ckpt):
                continue                        # the goldens belong to their checkpoint
            c = parity_case(eng, nm, sharded, rank, world, group)
            if c is not None:
                parity[nm] = c
            beat("parity " + nm)

    # ---- configs[3] as BASELINE words it (60 x 2000, site-sharded over the N ranks), on the N > 1 line: the same
    # per-GPU token count as the headline batch (4 x the sites, a quarter of the alignments)
    extra_configs = None
    if multi and not args.no_configs:
        n3, l3 = 60, 2000
        p3 = n3 * (n3 - 1) // 2
        if sharded:
            b3 = max(2, (args.batch // 4) * world)
            lo3, hi3 = pfdist.site_range(l3, world, rank)
        else:
            b3 = max(1, args.batch // 4)
            lo3, hi3 = 0, l3
        base3 = simulate_batch(min(b3, 4), n3, l3, seed=2 + (0 if sharded else rank))
        idx3 = np.ascontiguousarray(base3[np.arange(b3) % base3.shape[0]][:, :, lo3:hi3])
        d3, o3 = eng.malloc(max(idx3.nbytes, 1)), eng.malloc(b3 * p3 * 4)
        eng.h2d(d3, idx3)

        def step3():
            if sharded:
                eng.forward_sharded_device(d3, b3, n3, lo3, hi3, l3, o3)
            else:
                eng.forward_device(d3, b3, n3, l3, o3)

        step3()
        steps3 = max(2, min(args.steps, 5))
        dt3 = timed(step3, steps3, "configs[3] region")
        eng.free(d3)
        eng.free(o3)
        tot3 = (b3 if sharded else b3 * world) * steps3
        tok3 = (b3 if sharded else b3 * world) * p3 * l3 / world        # tokens per GPU and step
        extra_configs = {f"configs[3] 60x2000 {how}": {
            "n_seqs": n3, "n_sites": l3, "global_batch": b3 if sharded else b3 * world,
            "sites_per_rank": hi3 - lo3, "timed_steps": steps3, "ms_per_step": round(dt3 / steps3 * 1e3, 4),
            "alignments_per_s": round(tot3 / dt3, 3),
            "frac_mfma_per_gpu": round(FLOPS_ALG_PER_TOKEN * tok3 / (dt3 / steps3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
            "max_abs_err": (parity or {}).get("configs[3] 60x2000", {}).get("max_abs_err")}}
        beat("configs[3]")
        if sharded:
            # The case site sharding exists for: ONE long alignment per step, latency and rate.
            # With a single alignment there is no second half-batch to hide the collectives under: the 7 all-reduces
            # of [P][72] floats (510 KB) sit on the critical path - every kernel behind them (k_rowfin -> k_colstats ->
            # k_main) needs the reduced statistics of ALL pairs, so a split of the pairs would overlap nothing but the
            # 5 us k_rowfin.  Reported as measured: ms per alignment, and beside it the same
            # alignment unsharded on this rank's GPU for the ratio.
            one = np.ascontiguousarray(simulate_batch(1, n3, l3, seed=5)[:, :, lo3:hi3])
            d1, o1 = eng.malloc(max(one.nbytes, 1)), eng.malloc(p3 * 4)
            eng.h2d(d1, one)

            def step1():
                eng.forward_sharded_device(d1, 1, n3, lo3, hi3, l3, o1)

            step1()
            c0 = eng.collective_count()
            steps1 = max(3, min(args.steps, 10))
            dt1 = timed(step1, steps1, "strong-scaling region")
            ncoll1 = (eng.collective_count() - c0) / steps1
            res1 = np.empty((1, p3), np.float32)
            eng.d2h(res1, o1)
            crcs = group.allgather(zlib.crc32(res1.tobytes())) if group is not None else [0]
            entry = {"n_seqs": n3, "n_sites": l3, "global_batch": 1, "sites_per_rank": hi3 - lo3, "timed_steps": steps1,
                     "ms_per_alignment": round(dt1 / steps1 * 1e3, 4), "alignments_per_s": round(steps1 / dt1, 3),
                     "collectives_per_alignment": round(ncoll1, 2), "scaling": "strong",
                     "ranks_bit_identical": (len(set(crcs)) == 1) if world > 1 else None, "finite": bool(np.isfinite(res1).all())}
            if world == 1:
                # single rank (--force-dist): the same alignment through the unsharded entry point - the same bits,
                # and the cost of the seven (single-rank) collectives and their k_rowsum launches
                full = np.ascontiguousarray(simulate_batch(1, n3, l3, seed=5))
                eng.h2d(d1, full)
                ou = eng.malloc(p3 * 4)
                eng.forward_device(d1, 1, n3, l3, ou)
                dtu = timed(lambda: eng.forward_device(d1, 1, n3, l3, ou), steps1, "strong-scaling reference")
                resu = np.empty((1, p3), np.float32)
                eng.d2h(resu, ou)
                eng.free(ou)
                entry["bit_identical_to_pf_forward"] = bool(np.array_equal(resu, res1))
                entry["ms_per_alignment_unsharded"] = round(dtu / steps1 * 1e3, 4)
            eng.free(d1)
            eng.free(o1)
            extra_configs[f"60x2000 x1 sites-sharded x{world}"] = entry
            beat("strong scaling")
    if rank == 0:
        tokens_per_launch = B * P * (hi - lo)
        roof = None
        try:
            build = eng.build_info()
        except Exception:  # noqa: BLE001 - stand-in engines have no native library
            build = None
        if prof.get("main", (0, 0))[0]:
            n_main, ms_main = prof["main"]
            avg_s = ms_main / n_main * 1e-3
            nb = weights.n_blocks
            flops = tokens_per_launch * ((nb - 1) * FLOPS_MAIN_MID + FLOPS_MAIN_LAST) / nb
            ach = flops / avg_s / 1e12
            traffic, traffic_source = pmc_traffic(tokens_per_launch, build)
            roof = {"bound": "mfma", "kernel": "k_main", "achieved": round(ach, 2),
                    "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                    "traffic": traffic, "traffic_source": traffic_source,
                    "avg_launch_ms": round(avg_s * 1e3, 4), "launches": n_main,
                    "schedule": "second timed region of the same steps with the batch on one stream (a launch covers "
                                "the whole batch and has the chip to itself); `value` is the default two-stream schedule"
                                if not args.one_stream else "one stream (--one-stream)",
                    "note": "algorithmic flops (1 pass); the split-fp16 scheme issues 3 MFMA passes, "
                            "so frac tops out at 1/3"}
        metric, workload = workload_label(N, L, args.ckpt)
        line = {
            "metric": metric,
            "value": round(value, 3), "unit": "alignments/s", "n_gpus": n_devices, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16x3-split MFMA (two fp16 limbs per operand, 3 passes), fp32 accumulate/residual", "data": "synthetic",
            "config": {"workload": workload,
                       "global_batch": B if args.shard == "sites" else B * world,
                       "n_seqs": N, "n_sites": L, "parallelism": how,
                       "device": info["name"].strip(),
                       "n_ranks_in_comm": world if comm else 0,
                       "communicators": 2 if comm else 0,
                       "collectives_per_step": round(collectives_per_step, 2),
                       "reserve_cus": args.reserve_cus if comm else None,
                       "rccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS") if comm else None,
                       "rung": rung_info(),
                       # what the timed library was built from (pf_build_info): compiler, the scheduling strategy that
                       # REALLY compiled the kernels (sched_fallback: true = hipcc could not use the intended one),
                       # hashes of the sources
                       "build": build},
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
            "configs": extra_configs,
            "cpu_baseline": None,
        }
        if comm:
            line["config"]["rccl"] = comm
        if comm_note:
            line["config"]["note"] = comm_note
        if n_devices != world:
            line["n_ranks"] = world
            line["ranks_per_device"] = round(world / n_devices, 2)
            line["scaling_result"] = False
        if world == 1 and not multi and not args.no_configs and make_engine_for is not None:
            eng.free(d_idx)
            eng.free(d_out)
            d_idx = d_out = None
            streams(True)
            line["configs"] = configs_leg(eng, make_engine_for, device, parity)
        if parity is not None:
            errs = [c["max_abs_err"] for c in parity.values()]
            line["max_abs_err"] = max(errs) if errs else None
            line["max_abs_err_ok"] = bool(errs) and all(c["ok"] for c in parity.values())
            # (one rank: there is nothing to compare - null, not a vacuous true)
            line["ranks_bit_identical"] = all(c["ranks_bit_identical"] for c in parity.values()) if world > 1 else None
            line["parity"] = {"bound": PARITY_BOUND, "cases": parity,
                              "reference": "outputs of the reference's CPU forward committed under tests/golden/ "
                                           "(oracle/gen_golden.py; data, not the oracle)",
                              "how": "after the timed regions, through the entry point that was timed; every rank "
                                     "compares, errors and result CRCs are exchanged over the rendezvous"}
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
    beat("done")
    # every rank knows every case (the allgather is symmetric): all of them report a failed bound the same way
    parity_ok = parity is None or (len(parity) > 0 and all(c["ok"] for c in parity.values())) or \
        (len(parity) == 0 and os.path.basename(args.ckpt) != "pf.ckpt")
    if not parity_ok and rank == 0:
        print(f"bench: PARITY FAILED: {json.dumps(parity)}", file=sys.stderr)
    return value, parity_ok


# ---- self-launch: `python3 bench.py --gpus N` without a launcher -------------------------------------------
def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


# The fallback ladder of a self-launched N > 1 run: (name, extra arguments for the ranks).
RUNGS = [("sites, two streams / two communicators", []),
         ("sites, one stream (serial collectives)", ["--one-stream"]),
         ("whole alignments per rank, no collective", ["--shard", "alignments"])]


def ladder(args):
    if args.shard == "alignments":
        return RUNGS[2:]
    return RUNGS[1:] if args.one_stream else RUNGS


def rung_info():
    """What the launcher told this rank about the ladder (config.rung of the line); None outside a self-launch."""
    if "PF_BENCH_RUNG" not in os.environ:
        return None
    try:
        return {"index": int(os.environ["PF_BENCH_RUNG"]), "name": os.environ.get("PF_BENCH_RUNG_NAME", ""),
                "abandoned": json.loads(os.environ.get("PF_BENCH_RUNG_HISTORY", "[]"))}
    except ValueError:
        return None


def self_launch(args, argv):
    """`python3 bench.py --gpus N` without a launcher: this process touches neither HIP nor the engine.  It starts N
    fresh children of this script with the environment an external launcher would give them (``RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT / PF_RUN_ID``) - and those, like the processes of
    ``torch.distributed.run``, become the *supervisors* of the ranks (``supervise``): ONE implementation of the
    fallback ladder serves both forms.  Supervisor 0 inherits this process's stdout and prints the JSON line; every
    child's stderr is inherited.  Returns the supervisors' common exit code (they agree by construction); if they
    are still there ``--launch-timeout`` + 45 s after the start, their process groups are killed and the code is 124."""
    world = args.gpus
    env = dict(os.environ)
    # HSA_ENABLE_IPC_MODE_LEGACY=0: the GPU pool's host driver only supports dmabuf IPC; with the legacy mode RCCL's
    # intra-node transport setup fails with "hipIpcGetMemHandle: invalid argument" (task statement, Environment).
    # It is already exported on the boxes; the launcher pins it for its children in case a wrapper dropped it.
    env.update({"WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()),
                "PF_RUN_ID": uuid.uuid4().hex, "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    for k in ("PF_BENCH_RUNG", "PF_BENCH_RUNG_NAME", "PF_BENCH_RUNG_HISTORY", "PF_BENCH_PROGRESS"):
        env.pop(k, None)
    # supervisor 0 stretches the limits when the ranks publish a longer expected duration (many steps): it writes
    # the new overall limit (seconds since the start) here, and this process follows it
    fd, limit_file = tempfile.mkstemp(prefix="pf_bench_limit_")
    os.close(fd)
    env["PF_BENCH_LIMIT_FILE"] = limit_file
    t_start = time.monotonic()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv),
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), start_new_session=True,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    limit = args.launch_timeout

    def current_limit():
        try:
            with open(limit_file) as fh:
                return max(args.launch_timeout, float(fh.read().strip() or 0))
        except (OSError, ValueError):
            return args.launch_timeout
    def exited(p):
        # Has the supervisor ended?  Asked WITHOUT reaping it (WNOWAIT): the zombie keeps its pid - and, as the session
        # leader it is, its process-group id - reserved until p.wait() below, so the killpg that follows cannot reach a
        # stranger's recycled group (poll() reaped, and a reaped pid may be handed out again).
        if p.returncode is not None:
            return True
        try:
            return os.waitid(os.P_PID, p.pid, os.WEXITED | os.WNOHANG | os.WNOWAIT) is not None
        except ChildProcessError:
            return True
    while not all(exited(p) for p in procs):
        limit = current_limit()
        if time.monotonic() - t_start > limit + 45.0:
            break
        time.sleep(0.05)
    try:
        os.unlink(limit_file)
    except OSError:
        pass
    late = [p for p in procs if not exited(p)]
    # Every process group started above is signalled, the late ones and those whose supervisor has already left: a
    # supervisor that died on an exception may have left its rank behind, parked in a collective and holding its GPU
    # (ADVICE r04).  The groups are this launcher's own (start_new_session) and none of their leaders has been reaped
    # yet, so the ids still name them.
    for p in procs:
        try:
            os.killpg(p.pid, 15)
        except OSError:
            pass
    for p in procs:
        if p not in late:
            p.wait()
    for p in late:
        try:
            p.wait(timeout=5)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, 9)
            except OSError:
                pass
            p.wait()
    if late:
        print(f"bench: watchdog: {len(late)} supervisor(s) still running {limit + 45:.0f} s after the start; "
              "killed", file=sys.stderr)
        return EXIT_WATCHDOG
    codes = [p.returncode for p in procs]
    if len(set(codes)) > 1:
        print(f"bench: supervisors disagree on the exit code: {codes}", file=sys.stderr)
    return next((c for c in codes if c), 0)


def supervise(args, argv, rank, world):
    """The processes a launcher started (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, or
    `self_launch`) do not become ranks - a stalled collective would then hang until someone else's timeout, with no
    number.  Each becomes the SUPERVISOR of one rank instead: it touches neither HIP nor the engine, starts the real
    rank as a fresh child of this script and watches it; the supervisors meet in a rendezvous of their own and walk
    the fallback ladder in lockstep (every 0.25 s they exchange their child's exit code and the age of its last
    progress mark; the verdict is a function of the exchanged values, so all of them reach it together): a dead or
    stalled rank anywhere makes every supervisor kill its child and start a fresh one for the next rung, under a new
    rendezvous key.  Supervisor 0 relays its child's JSON line.  Exit code: 0, or 3 / 4 from the ranks (parity bound,
    too few GPUs: final), 124 if every rung was abandoned by a watchdog, else 1."""
    import signal
    from phyloformer_amd.rendezvous import TcpGroup, default_key

    class Stopped(Exception):
        pass

    def on_signal(signum, _frame):      # torchrun and self_launch stop workers with SIGTERM: unwind through the finally
        raise Stopped(signum)           # blocks below, so that the rank this supervisor started never outlives it
    for sig in (signal.SIGTERM, signal.SIGINT):
        try:
            signal.signal(sig, on_signal)
        except (ValueError, OSError):   # not the main thread (tests)
            pass

    def stop_child(child):
        if child is not None and child.poll() is None:            # exactly the process started below
            child.terminate()
            try:
                child.wait(timeout=5)
            except subprocess.TimeoutExpired:
                child.kill()
                child.wait()

    def last_mark(path):
        """(stage, expect_s, budget_s) of the rank's newest progress line."""
        stage, expect, budget = "-", 0.0, 0.0
        try:
            with open(path) as fh:
                lines = fh.read().strip().splitlines()
        except OSError:
            return stage, expect, budget
        for ln in lines:
            for tok in ln.split()[2:]:
                if tok.startswith("budget="):
                    budget = float(tok[7:])
        if lines:
            parts = lines[-1].split()
            stage = parts[1] if len(parts) > 1 else "-"
            for tok in parts[2:]:
                if tok.startswith("expect="):
                    expect = float(tok[7:])
        return stage, expect, budget
    base_run = os.environ.get("PF_RUN_ID") or os.environ.get("TORCHELASTIC_RUN_ID", "none")
    sup = TcpGroup(rank, world, key=default_key() + "_supervisors", timeout=max(30.0, args.rung_timeout))
    t_launch = time.monotonic()
    history, rc, text = [], 1, ""
    rungs = ladder(args)
    first = len(RUNGS) - len(rungs) + 1
    child = None
    launch_limit = args.launch_timeout
    try:
        for index, (name, extra) in enumerate(rungs, start=first):
            left = launch_limit - (time.monotonic() - t_launch)
            if any(sup.allgather(index > first and left < 15)):
                if rank == 0:
                    print("bench: no time left for another rung", file=sys.stderr)
                break
            progress = tempfile.mkdtemp(prefix="pf_bench_progress_")
            env = dict(os.environ, PF_RUN_ID=f"{base_run}_rung{index}", PF_BENCH_RUNG=str(index), PF_BENCH_RUNG_NAME=name,
                       PF_BENCH_RUNG_HISTORY=json.dumps(history), PF_BENCH_PROGRESS=progress,
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv) + extra, env=env,
                                     stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL)
            out0 = []
            reader = None
            if rank == 0:
                reader = threading.Thread(target=lambda: out0.append(child.stdout.read()), daemon=True)
                reader.start()
            t0, last, size = time.monotonic(), time.monotonic(), 0
            mark = os.path.join(progress, f"rank{rank}")
            rc, why = 0, ""
            while True:
                try:
                    sz = os.path.getsize(mark)
                except OSError:
                    sz = 0
                now = time.monotonic()
                if sz != size:
                    size, last = sz, now
                stage, expect, budget = last_mark(mark)
                seen = sup.allgather([child.poll(), round(now - last, 2), round(now - t0, 2),
                                      round(now - t_launch, 2), stage, expect, budget])
                codes = [c for c, *_ in seen]
                where = "; ".join(f"rank {r}: {st}" for r, (_c, _a, _b, _d, st, *_x) in enumerate(seen))
                # limits stretch with what the ranks expect (functions of the exchanged values: the same everywhere)
                stall_limit = max(args.stall_timeout, 2.0 * max(e for *_x, e, _b in seen) + 30.0)
                rung_limit = max(args.rung_timeout, 2.0 * max(b for *_x, b in seen) + 60.0)
                new_launch = launch_limit if rung_limit <= args.rung_timeout else \
                    max(launch_limit, seen[0][3] - seen[0][2] + rung_limit + 30.0)     # only a published budget stretches
                if new_launch > launch_limit:
                    launch_limit = new_launch
                    if rank == 0 and os.environ.get("PF_BENCH_LIMIT_FILE"):
                        try:
                            with open(os.environ["PF_BENCH_LIMIT_FILE"], "w") as fh:
                                fh.write(f"{launch_limit:.1f}")
                        except OSError:
                            pass
                if any(c not in (None, 0) for c in codes):
                    bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
                    rc = bad[0][1] if all(c == bad[0][1] for _, c in bad) and bad[0][1] in (EXIT_PARITY, EXIT_DEVICES) else 1
                    why = ", ".join(f"rank {r} exited with code {c}" for r, c in bad)
                    if rc == 1:
                        time.sleep(0.5)
                    break
                if all(c == 0 for c in codes):
                    break
                if seen[0][2] > rung_limit or seen[0][3] > launch_limit:
                    rc, why = EXIT_WATCHDOG, f"watchdog: ranks still running after {seen[0][2]:.0f} s ({where})"
                    break
                if min(age for _c, age, *_ in seen) > stall_limit:
                    rc, why = EXIT_WATCHDOG, f"watchdog: no progress from any rank for {stall_limit:.0f} s ({where})"
                    break
                time.sleep(0.25)
            stop_child(child)
            if reader is not None:
                reader.join(timeout=5)
            shutil.rmtree(progress, ignore_errors=True)
            text = (out0[0] if out0 else b"").decode(errors="replace")
            if rc and rank == 0:
                print(f"bench: rung {index} ({name}): {why}; all {world} ranks stopped", file=sys.stderr)
            if rc == 0 and not any(sup.allgather(rank == 0 and not text.strip())):
                break
            if rc == 0:
                rc, why = 1, "rank 0 printed nothing"
            if rc in (EXIT_PARITY, EXIT_DEVICES):
                break
            history.append({"rung": index, "name": name, "why": why})
            text = ""
    except Stopped as exc:
        print(f"bench: supervisor {rank}: signal {exc.args[0]}; stopping its rank", file=sys.stderr)
        rc, text = 128 + int(exc.args[0]), ""
    finally:
        # whatever way this function is left - a peer supervisor gone (allgather raises), a signal, a bug - the rank
        # it started does not outlive it (an orphan would sit in an RCCL collective holding its GPU)
        stop_child(child)
        sup.close()
    if rank == 0 and text:
        sys.stdout.write(text)
        sys.stdout.flush()
    if rc and history and rc not in (EXIT_PARITY, EXIT_DEVICES):
        rc = EXIT_WATCHDOG if all("watchdog" in h["why"] for h in history) else 1
    return rc


def engine_factories(ckpt):
    """(weights, make_engine(device), make_engine_for(ckpt, device)).  PF_BENCH_ENGINE_FACTORY=module:function
    swaps in a stand-in for the CPU test of the self-launch path: ``function(device)`` returns an object with the
    Engine interface and ``module.bench_weights()`` one with ``n_blocks``."""
    hook = os.environ.get("PF_BENCH_ENGINE_FACTORY")
    if hook:
        import importlib
        print(f"bench: PF_BENCH_ENGINE_FACTORY={hook} REPLACES the GPU engine (test hook; the line's config.device names "
              "the stand-in)", file=sys.stderr)
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
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) > 1 and "PF_BENCH_RUNG" not in os.environ \
            and not os.environ.get("PF_BENCH_NO_SUPERVISOR"):
        # started by an external launcher (torch.distributed.run): supervise a fresh child per rung (see supervise)
        return supervise(args, argv, int(os.environ.get("RANK", "0")), int(os.environ["WORLD_SIZE"]))
    beat("process up")          # (the launcher's stall watchdog counts from the last mark of ANY rank)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    if (world > 1 or args.force_dist) and args.rccl_max_nchannels > 0:
        # Opt-in only: a channel cap has never been measured at N > 1, and RCCL's channels are
        # independent workgroups - those that find no free CU next to the persistent kernels wait for the end of
        # a k_main launch (<= 4 ms), they do not deadlock.  Read by librccl when it is first loaded.
        os.environ["NCCL_MAX_NCHANNELS"] = str(args.rccl_max_nchannels)

    # Native libraries write to file descriptor 1 as they please (RCCL prints a version banner there when a
    # communicator fails): the contract is ONE JSON line on stdout, so descriptor 1 is pointed at stderr for the
    # life of the rank and the line goes to a private copy of the original stdout.
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    from phyloformer_amd.rendezvous import TcpGroup
    w, make_engine, make_engine_for = engine_factories(args.ckpt)
    beat("weights loaded")
    group = TcpGroup(rank, world) if (world > 1 or args.force_dist) else None
    beat("rendezvous")
    ok = True
    try:
        _value, ok = run(args, rank, world, local_rank, group, make_engine, w, out=line_out, make_engine_for=make_engine_for)
    finally:
        if group is not None:
            group.close()
        line_out.flush()
    return 0 if ok else EXIT_PARITY


if __name__ == "__main__":
    sys.exit(main())
