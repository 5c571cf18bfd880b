#!/usr/bin/env python3
"""Infer evolutionary distances with Phyloformer on an MI355X — drop-in for the reference CLI.

Same command line and output files as /root/reference/infer_alns.py:42-123:

    python infer_alns.py WEIGHTS ALNDIR [-o OUTDIR] [-t]

Every entry of ``ALNDIR`` must end in ``.fa``/``.fasta`` (case-insensitive,
infer_alns.py:36-38,100-103) or the run aborts with ``ValueError``; for each
alignment ``OUTDIR/<stem>.phy`` receives the PHYLIP distance matrix
(``%.10f``) and, with ``-t``, ``OUTDIR/<stem>.nj.nwk`` a neighbour-joining tree.

Additive flags (not in the reference): ``--device``, ``--batch`` (group
same-shape alignments into one launch), ``--bench`` (print a JSON timing line).
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
    parser.add_argument("--batch", type=int, default=1,
                        help="max same-shape alignments per launch (default 1 = reference order)")
    parser.add_argument("--bench", action="store_true", help="print a JSON timing summary to stderr")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)

    from phyloformer_amd.fasta import load_alignment
    from phyloformer_amd.model import Phyloformer
    from phyloformer_amd.phylip import vec_to_phylip

    try:
        from tqdm import tqdm
    except Exception:  # pragma: no cover
        def tqdm(x, **_k):
            return x

    model = Phyloformer.from_checkpoint(args.weights, device=args.device)
    model.eval()

    in_dir = os.path.abspath(args.alndir)
    out_dir = os.path.abspath(args.outdir)  # TypeError if omitted, as in the reference (:90)
    os.makedirs(out_dir, exist_ok=True)

    paths = glob(f"{in_dir}/*")
    t_io = t_fwd = 0.0
    n_done = 0

    def flush(group):
        nonlocal t_fwd, t_io, n_done
        if not group:
            return
        import numpy as np
        t0 = time.perf_counter()
        preds = model.engine.forward(np.stack([g[1] for g in group]))
        t_fwd += time.perf_counter() - t0
        t0 = time.perf_counter()
        for (alnpath, _idx, ids), pred in zip(group, preds):
            stem = Path(alnpath).stem
            dm, phylip = vec_to_phylip(pred, ids)
            with open(os.path.join(out_dir, f"{stem}.phy"), "w") as outfile:
                outfile.write(phylip)
            if args.trees:
                from phyloformer_amd.nj import neighbor_joining
                with open(os.path.join(out_dir, f"{stem}.nj.nwk"), "w") as outfile:
                    outfile.write(neighbor_joining(dm.astype("float64"), ids))
        t_io += time.perf_counter() - t0
        n_done += len(group)
        group.clear()

    group = []
    for alnpath in tqdm(paths):
        if not has_fasta_ext(alnpath):
            raise ValueError("Input files must be fasta files (.fa or .fasta). Got " f"{alnpath}")
        t0 = time.perf_counter()
        idx, ids = load_alignment(alnpath)
        t_io += time.perf_counter() - t0
        if group and (group[0][1].shape != idx.shape or len(group) >= args.batch):
            flush(group)
        group.append((alnpath, idx, ids))
    flush(group)

    if args.bench:
        print(json.dumps({"alignments": n_done, "forward_s": round(t_fwd, 6), "io_s": round(t_io, 6),
                          "alignments_per_s": round(n_done / t_fwd, 3) if t_fwd > 0 else None}),
              file=sys.stderr)
    model.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
