"""End-to-end tree check (README.md:83-99 of the reference).
Distances -> tree -> comparison with data/testdata/trees.  In the build container FastME from the
reference checkout can be used (--fastme PATH); everywhere else the build's own NJ.
    python tools/tree_check.py [--fastme /root/reference/bin/bin_linux/fastme] [--matrices DIR]
Without --matrices the reference's own distances (tests/golden/e2e_testdata.npz, pf.ckpt) are used."""
import argparse, json, os, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from phyloformer_amd import fasta, phylip, treecmp
from phyloformer_amd.nj import neighbor_joining

ap = argparse.ArgumentParser()
ap.add_argument("--fastme", default=None)
ap.add_argument("--matrices", default=None, help="directory of .phy files (output of infer_alns.py)")
ap.add_argument("--model", default="pf")
a = ap.parse_args()
gold = np.load(os.path.join(REPO, "tests/golden/e2e_testdata.npz"))
rows = []
for name in sorted(os.listdir(os.path.join(REPO, "data/testdata/msas"))):
    stem = name[:-3]
    _idx, ids = fasta.load_alignment(os.path.join(REPO, "data/testdata/msas", name))
    if a.matrices:
        lines = open(os.path.join(a.matrices, stem + ".phy")).read().splitlines()
        dm = np.array([[float(v) for v in l.split(" ")[1:]] for l in lines[1:]])
        text = "\n".join(lines) + "\n"
    else:
        dm, text = phylip.vec_to_phylip(gold[f"{a.model}/{stem}"], ids)
    if a.fastme:
        with tempfile.TemporaryDirectory() as t:
            open(os.path.join(t, "m.phy"), "w").write(text)
            subprocess.run([a.fastme, "-i", os.path.join(t, "m.phy"), "-o", os.path.join(t, "t.nwk"), "--nni", "--spr"],
                           check=True, capture_output=True)
            nwk = open(os.path.join(t, "t.nwk")).read()
    else:
        nwk = neighbor_joining(dm.astype(np.float64), ids)
    true = treecmp.parse_newick(open(os.path.join(REPO, "data/testdata/trees", stem + ".nwk")).read())
    est = treecmp.parse_newick(nwk)
    rf, nrf = treecmp.robinson_foulds(true, est)
    rows.append((stem, rf, nrf, treecmp.branch_score(true, est)))
print(json.dumps({"method": "fastme --nni --spr" if a.fastme else "own NJ", "n": len(rows),
                  "mean_rf": round(float(np.mean([r[1] for r in rows])), 4),
                  "mean_normalised_rf": round(float(np.mean([r[2] for r in rows])), 4),
                  "mean_branch_score": round(float(np.mean([r[3] for r in rows])), 4)}))
