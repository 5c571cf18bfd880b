"""-m gpu: the drop-in CLI (infer_alns.py) end to end against the reference CLI's behaviour."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(repo, args, **kw):
    return subprocess.run([sys.executable, os.path.join(repo, "infer_alns.py"), *args],
                          capture_output=True, text=True, cwd=repo, **kw)


def _read_phy(path):
    lines = open(path).read().splitlines()
    n = int(lines[0])
    ids = [l.split(" ")[0] for l in lines[1:1 + n]]
    dm = np.array([[float(v) for v in l.split(" ")[1:]] for l in lines[1:1 + n]])
    return ids, dm


def test_cli_matches_reference_phylip(repo, tmp_path):
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    for stem in ("0_20_tips", "1_30_tips"):
        shutil.copy(os.path.join(repo, "data/testdata/msas", f"{stem}.fa"), ind / f"{stem}.fa")
    r = _run(repo, [os.path.join(repo, "models/pf_base.ckpt"), str(ind), "-o", str(outd), "-t", "--bench"])
    assert r.returncode == 0, r.stderr
    ids, dm = _read_phy(outd / "0_20_tips.phy")
    rids, rdm = _read_phy(os.path.join(repo, "tests/golden/0_20_tips.pf_base.phy"))   # reference CLI output
    assert ids == rids and dm.shape == (20, 20)
    assert np.abs(dm - rdm).max() <= 1e-4
    text = open(outd / "0_20_tips.phy").read()
    assert text.splitlines()[0] == "20" and len(text.splitlines()[1].split(" ")[1].split(".")[1]) == 10
    nwk = open(outd / "1_30_tips.nj.nwk").read()
    assert nwk.endswith(";\n") and nwk.count(",") == 29     # 30 leaves


def test_cli_error_behaviour(repo, tmp_path):
    ind = tmp_path / "in"
    ind.mkdir()
    (ind / "notes.txt").write_text("x")
    r = _run(repo, [os.path.join(repo, "models/pf.ckpt"), str(ind), "-o", str(tmp_path / "o")])
    assert r.returncode != 0 and "Input files must be fasta files" in r.stderr     # infer_alns.py:100-103
    r = _run(repo, [os.path.join(repo, "models/pf.ckpt"), str(ind)])
    assert r.returncode != 0 and "TypeError" in r.stderr                            # -o omitted, :53,90
