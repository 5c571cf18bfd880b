#!/usr/bin/env python3
"""NJ goldens from a reference-held neighbour-joining: FastME ``-m N`` (no NNI / SPR afterwards).

Build container only.  The reference's ``--trees`` path calls ``skbio.tree.nj`` (infer_alns.py:62-64,
120-123); scikit-bio is not installed here, but the reference checkout ships FastME
(``bin/bin_linux/fastme``), whose ``-m N`` is the same Saitou & Nei algorithm.  For each of the 20 test MSAs
the reference's own pf.ckpt distances (tests/golden/e2e_testdata.npz) are written as a PHYLIP matrix, FastME
builds the NJ tree, and the Newick text is stored in ``tests/golden/nj_fastme.json``.

    python oracle/gen_golden_nj.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
FASTME = "/root/reference/bin/bin_linux/fastme"


def main():
    from phyloformer_amd import fasta
    from phyloformer_amd.phylip import vec_to_phylip
    gold = np.load(os.path.join(REPO, "tests", "golden", "e2e_testdata.npz"))
    msas = os.path.join(REPO, "data", "testdata", "msas")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name in sorted(os.listdir(msas)):
            stem = name[:-3]
            _idx, ids = fasta.load_alignment(os.path.join(msas, name))
            _dm, text = vec_to_phylip(gold[f"pf/{stem}"], ids)
            src, dst = os.path.join(tmp, stem + ".phy"), os.path.join(tmp, stem + ".nwk")
            with open(src, "w") as fh:
                fh.write(text)
            # -m N: neighbour joining; -n / -s absent: no topology search afterwards
            subprocess.run([FASTME, "-i", src, "-o", dst, "-m", "N"], check=True, capture_output=True, cwd=tmp)
            with open(dst) as fh:
                out[stem] = fh.read().strip()
            print(stem, len(out[stem]), "chars")
    with open(os.path.join(REPO, "tests", "golden", "nj_fastme.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
