#!/usr/bin/env python3
"""Infer evolutionary distances with Phyloformer on an MI355X — drop-in for the reference CLI.

Same command line and output files as /root/reference/infer_alns.py:42-123:

    python infer_alns.py WEIGHTS ALNDIR [-o OUTDIR] [-t]

Every entry of ``ALNDIR`` must end in ``.fa``/``.fasta`` (case-insensitive,
infer_alns.py:36-38,100-103) or the run aborts with ``ValueError``; for each
alignment ``OUTDIR/<stem>.phy`` receives the PHYLIP distance matrix
(``%.10f``) and, with ``-t``, ``OUTDIR/<stem>.nj.nwk`` a neighbour-joining tree.

Additive flags (not in the reference): ``--device`` / ``--devices 0,1,...`` (one
process per GPU; ``--shard files`` - the default - deals the files to the GPUs, ``--shard sites``
spreads every alignment over them: each rank holds a block of sites, RCCL all-reduces inside
``pf_forward_sharded``, rank 0 writes the outputs), ``--batch`` (same-shape alignments per launch;
default: fill a token budget per shape), ``--io-threads``, ``--gpu-streams``, ``--precise``, ``--python-io``,
``--bench`` (print a JSON timing line).  Scheduling lives in
``phyloformer_amd/scheduler.py``: files are bucketed by shape, parsed ahead of the
GPU and written behind it.  A directory entry without a FASTA extension, or a file that does
not parse, has the reference's side effects (infer_alns.py:97-117): every entry in front of it
(in ``glob`` order, as there) gets its output, nothing behind it does, then the reference's
exception is raised; the multi-GPU modes, which have no counterpart, refuse such a directory up front.
The forward pass runs in ``libphyloformer_amd.so``; there is no CPU fallback.
"""
import argparse
import json
import os
import sys
import time
from glob import glob
from pathlib import Path


def has_fasta_ext(alnpath):
    """Checks if a path ends in .fa or .fasta"""
    return alnpath.lower().endswith(".fa") or alnpath.lower().endswith(".fasta")


def build_parser():
    parser = argparse.ArgumentParser(description="Infer evolutionnary distances with PhyloFormer")
    parser.add_argument("weights", help="Path to model weights to use")
    parser.add_argument("alndir", help="Path to directory containing alignments to infer")
    parser.add_argument("--outdir", "-o", default=None, required=False,
                        help="Path to directory where inferred distance matrices will be written")
    parser.add_argument("--trees", "-t", action="store_true",
                        help="Output NJ trees as well as matrices")
    parser.add_argument("--device", type=int, default=0, help="HIP device ordinal (default 0)")
    parser.add_argument("--devices", default=None,
                        help="comma-separated HIP device ordinals: shard the files over these GPUs, "
                             "one process per GPU (alignment-level data parallelism, no collective)")
    parser.add_argument("--shard", choices=["files", "sites"], default="files",
                        help="with --devices: 'files' (default) gives every GPU its share of the files, no collective; "
                             "'sites' spreads every alignment over the GPUs (each holds L / n sites of every pair, the "
                             "row-attention statistics and the final site sums are all-reduced over RCCL): for "
                             "alignments too long or too few to fill the GPUs one file each")
    parser.add_argument("--batch", type=int, default=0,
                        help="same-shape alignments per launch; 0 (default) = fill a token budget per shape, "
                             "1 = one alignment per launch as in the reference")
    parser.add_argument("--io-threads", type=int, default=4, help="FASTA loader / PHYLIP writer threads")
    parser.add_argument("--gpu-streams", type=int, default=2,
                        help="engines (HIP streams, one host thread each) per GPU; 2 hides the host-side gaps "
                             "of a synchronous forward, 1 = one launch sequence at a time")
    parser.add_argument("--precise", choices=["auto", "always", "never"], default="auto",
                        help="float64 kernels: auto = for alignments of fewer than 32 sites or 8,192 pair-site tokens (where the fp32 reference itself is "
                             "ill-conditioned) and, after the fact, for any alignment with a predicted distance above 8 substitutions per "
                             "site (never an alignment; an absolute 1e-4 there is fp32's own rounding level); always = every alignment (3-9 x slower); never = the split-fp16 MFMA kernels "
                             "on every shape")
    parser.add_argument("--python-io", action="store_true",
                        help="use the pure-Python FASTA parser and PHYLIP writer instead of the native ones")
    parser.add_argument("--worker", default=None, help=argparse.SUPPRESS)   # "r/W": share r of W of the files
    parser.add_argument("--bench", action="store_true", help="print a JSON timing summary to stderr")
    return parser


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)

    from phyloformer_amd import scheduler

    in_dir = os.path.abspath(args.alndir)
    out_dir = os.path.abspath(args.outdir)  # TypeError if omitted, as in the reference (:90)
    os.makedirs(out_dir, exist_ok=True)

    if args.devices and not args.worker:
        devices = [int(d) for d in args.devices.split(",") if d.strip() != ""]
        if len(devices) > 1:
            t0 = time.perf_counter()
            child_argv = [a for a in argv if a != "--bench"]
            for flag in ("--devices", "--device"):
                while flag in child_argv:
                    k = child_argv.index(flag)
                    del child_argv[k:k + 2]
            child_argv = [a for a in child_argv if not a.startswith("--devices=") and not a.startswith("--device=")]
            for flag in ("--shard",):
                while flag in child_argv:
                    k = child_argv.index(flag)
                    del child_argv[k:k + 2]
            child_argv = [a for a in child_argv if not a.startswith("--shard=")]
            rc, reports = scheduler.run_multi_device(os.path.abspath(__file__), child_argv, devices, shard=args.shard)
            if args.bench:
                wall = time.perf_counter() - t0
                sites = args.shard == "sites" and all(r.get("site_sharded_over") for r in reports)
                # site-sharded ranks all work on every alignment: the job's count and rate are rank 0's
                n = (reports[0]["alignments"] if reports else 0) if sites else sum(r["alignments"] for r in reports)
                rate = (reports[0]["alignments_per_s"] or 0) if (sites and reports) else sum(r["alignments_per_s"] or 0 for r in reports)
                print(json.dumps({"alignments": n, "devices": devices, "shard": args.shard if sites or args.shard == "files" else "files (fallback)",
                                  "wall_s_incl_startup": round(wall, 4), "alignments_per_s": round(rate, 3),
                                  "workers": reports}), file=sys.stderr)
            return rc
        args.device = devices[0]

    from phyloformer_amd.model import Phyloformer

    try:
        from tqdm import tqdm
    except Exception:  # pragma: no cover
        tqdm = None

    paths = glob(f"{in_dir}/*")
    rank, world = 0, 1
    if args.worker:
        rank, world = (int(v) for v in args.worker.split("/"))
    elif int(os.environ.get("WORLD_SIZE", "1")) > 1 and "RANK" in os.environ:
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])   # launched by torchrun

    if args.shard == "sites" and (world > 1 or os.environ.get("PF_CLI_FORCE_RCCL")):
        return run_site_sharded(args, paths, rank, world, out_dir, tqdm)

    t0 = time.perf_counter()
    model = Phyloformer.from_checkpoint(args.weights, device=args.device, engine_factory=scheduler.cli_engine)
    model.eval()
    load_s = time.perf_counter() - t0

    if world > 1:
        for p in paths:
            if not scheduler.has_fasta_ext(p):
                raise ValueError("Input files must be fasta files (.fa or .fasta). Got " f"{p}")
        paths = scheduler.slice_paths(paths, rank, world)

    bar = tqdm(total=len(paths)) if (tqdm is not None and world == 1) else None
    engines = [model.engine]
    if args.batch != 1:
        engines += [scheduler.cli_engine(model.weights, args.device) for _ in range(max(1, args.gpu_streams) - 1)]
    for e in engines:
        e.set_option("precise", {"auto": -1, "always": 1, "never": 0}[args.precise])
    if len(engines) > 1:
        # several engines already keep several streams busy; each splitting its batches over two more only adds
        # contention (tools/cli_bench.py, same box: 505 against 498 alignments/s)
        for e in engines:
            e.set_option("two_streams", 0)
    runner = scheduler.DirectoryRunner(engines, out_dir, trees=args.trees, batch=args.batch,
                                       io_threads=args.io_threads, native_io=not args.python_io,
                                       progress=bar.update if bar is not None else None)
    try:
        stats = runner.run(paths)
    finally:
        if bar is not None:
            bar.close()
    if args.bench:
        rep = scheduler.summarize(stats, load_s)
        rep["device"] = args.device
        print(json.dumps(rep), file=sys.stderr)
    for e in engines[1:]:
        e.close()
    model.close()
    return 0


def run_site_sharded(args, paths, rank, world, out_dir, tqdm):
    """One rank of ``--devices ... --shard sites`` (started by scheduler.run_multi_device, which set the rendezvous
    environment).  The ranks first agree that every one of them has its RCCL communicators; if not, all of them
    destroy theirs and fall back to ``--shard files`` (each its share of the files, no collective) - a run that
    still writes every output.  ``PF_CLI_FORCE_RCCL=1`` makes a single process take this path with a real
    single-rank communicator (the GPU test of a 1-GPU box)."""
    from phyloformer_amd import dist as pfdist
    from phyloformer_amd import scheduler
    from phyloformer_amd.weights import load_weights

    t0 = time.perf_counter()
    weights = load_weights(args.weights)
    engine = scheduler.cli_engine(weights, args.device)
    engine.set_option("precise", {"auto": -1, "always": 1, "never": 0}[args.precise])
    load_s = time.perf_counter() - t0
    group = None
    try:
        if world > 1:
            from phyloformer_amd.rendezvous import TcpGroup
            group = TcpGroup(rank, world)
            ok, why = 1, ""
            try:
                pfdist.init_engine_comm(engine, group)
            except Exception as exc:  # noqa: BLE001 - every rank learns about it below
                ok, why = 0, f"rank {rank}: {type(exc).__name__}: {exc}"
            seen = group.allgather([ok, why])
            if not all(o for o, _ in seen):
                engine.comm_destroy()
                if rank == 0:
                    print("infer_alns: site-sharding unavailable (" + "; ".join(w for o, w in seen if not o) +
                          "); sharding the files over the GPUs instead", file=sys.stderr)
                for p in paths:
                    if not scheduler.has_fasta_ext(p):
                        raise ValueError("Input files must be fasta files (.fa or .fasta). Got " f"{p}")
                runner = scheduler.DirectoryRunner([engine], out_dir, trees=args.trees, batch=args.batch,
                                                   io_threads=args.io_threads, native_io=not args.python_io)
                stats = runner.run(scheduler.slice_paths(paths, rank, world))
                group.barrier()
                if args.bench:
                    rep = scheduler.summarize(stats, load_s)
                    rep["device"] = args.device
                    print(json.dumps(rep), file=sys.stderr)
                return 0
        else:
            engine.set_option("force_rccl", 1)
            engine.comm_init(engine.unique_id(), 0, 1)
        bar = tqdm(total=len(paths)) if (tqdm is not None and rank == 0 and not args.worker) else None
        runner = scheduler.SiteShardedRunner(engine, group, rank, world, out_dir, trees=args.trees, batch=args.batch,
                                             io_threads=args.io_threads, native_io=not args.python_io,
                                             progress=bar.update if bar is not None else None,
                                             heartbeat_s=(float(os.environ.get("PF_CLI_HEARTBEAT", "5")) if args.worker else None))
        try:
            stats = runner.run(paths)
        finally:
            if bar is not None:
                bar.close()
        if args.bench:
            rep = scheduler.summarize(stats, load_s)
            rep.update({"device": args.device, "site_sharded_over": world, "rank": rank,
                        "collectives": engine.collective_count() if hasattr(engine, "collective_count") else None})
            print(json.dumps(rep), file=sys.stderr)
        return 0
    finally:
        if group is not None:
            group.close()
        engine.close()


if __name__ == "__main__":
    sys.exit(main())
