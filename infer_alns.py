#!/usr/bin/env python3
"""Infer evolutionary distances with Phyloformer on an MI355X — drop-in for the reference CLI.

Same command line and output files as /root/reference/infer_alns.py:42-123:

    python infer_alns.py WEIGHTS ALNDIR [-o OUTDIR] [-t]

Every entry of ``ALNDIR`` must end in ``.fa``/``.fasta`` (case-insensitive,
infer_alns.py:36-38,100-103) or the run aborts with ``ValueError``; for each
alignment ``OUTDIR/<stem>.phy`` receives the PHYLIP distance matrix
(``%.10f``) and, with ``-t``, ``OUTDIR/<stem>.nj.nwk`` a neighbour-joining tree.

Additive flags (not in the reference): ``--device`` / ``--devices 0,1,...`` (one
process per GPU, files sharded), ``--batch`` (same-shape alignments per launch;
default: fill a token budget per shape), ``--io-threads``, ``--gpu-streams``, ``--python-io``,
``--bench`` (print a JSON timing line).  Scheduling lives in
``phyloformer_amd/scheduler.py``: files are bucketed by shape, parsed ahead of the
GPU and written behind it.  Unlike the reference, a non-FASTA entry aborts the run
before the first forward instead of when the loop reaches it.
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
    parser.add_argument("--batch", type=int, default=0,
                        help="same-shape alignments per launch; 0 (default) = fill a token budget per shape, "
                             "1 = one alignment per launch as in the reference")
    parser.add_argument("--io-threads", type=int, default=4, help="FASTA loader / PHYLIP writer threads")
    parser.add_argument("--gpu-streams", type=int, default=2,
                        help="engines (HIP streams, one host thread each) per GPU; 2 hides the host-side gaps "
                             "of a synchronous forward, 1 = one launch sequence at a time")
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
            rc, reports = scheduler.run_multi_device(os.path.abspath(__file__), child_argv, devices)
            if args.bench:
                wall = time.perf_counter() - t0
                n = sum(r["alignments"] for r in reports)
                print(json.dumps({"alignments": n, "devices": devices, "wall_s_incl_startup": round(wall, 4),
                                  "alignments_per_s": round(sum(r["alignments_per_s"] or 0 for r in reports), 3),
                                  "workers": reports}), file=sys.stderr)
            return rc
        args.device = devices[0]

    from phyloformer_amd.model import Phyloformer

    try:
        from tqdm import tqdm
    except Exception:  # pragma: no cover
        tqdm = None

    t0 = time.perf_counter()
    model = Phyloformer.from_checkpoint(args.weights, device=args.device)
    model.eval()
    load_s = time.perf_counter() - t0

    paths = glob(f"{in_dir}/*")
    rank, world = 0, 1
    if args.worker:
        rank, world = (int(v) for v in args.worker.split("/"))
    elif int(os.environ.get("WORLD_SIZE", "1")) > 1 and "RANK" in os.environ:
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])   # launched by torchrun
    if world > 1:
        for p in paths:
            if not scheduler.has_fasta_ext(p):
                raise ValueError("Input files must be fasta files (.fa or .fasta). Got " f"{p}")
        paths = scheduler.slice_paths(paths, rank, world)

    bar = tqdm(total=len(paths)) if (tqdm is not None and world == 1) else None
    engines = [model.engine]
    if args.batch != 1:
        from phyloformer_amd.engine import Engine
        engines += [Engine(model.weights, device=args.device) for _ in range(max(1, args.gpu_streams) - 1)]
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


if __name__ == "__main__":
    sys.exit(main())
