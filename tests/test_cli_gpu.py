"""-m gpu: the drop-in CLI (infer_alns.py) end to end against the reference CLI's behaviour."""
import json
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


@pytest.mark.parametrize("scenario", ["bad_extension", "bad_residue", "too_many_seqs"])
def test_cli_bad_entry_leaves_what_the_reference_leaves(repo, tmp_path, scenario):
    """The whole CLI on the GPU against the fixture the REAL reference CLI produced (oracle/gen_golden_cli_errors.py):
    the same file names; the entries `glob` lists in front of the offender get their .phy, nothing behind it does, and
    the process dies with the reference's exception (infer_alns.py:97-117)."""
    import json
    from glob import glob
    from phyloformer_amd.msa_sim import simulate_batch, to_fasta
    g = json.load(open(os.path.join(repo, "tests/golden/cli_bad_entry.json")))[scenario]
    ind = tmp_path / "in"
    ind.mkdir()
    alns = iter(simulate_batch(8, 5, 12, seed=77))
    for name in g["listing_order"]:
        if name == g["offender"]:
            (ind / name).write_bytes({"bad_extension": b"not an alignment\n", "bad_residue": b">s0\nARNDB\n>s1\nARNDC\n",
                                      "too_many_seqs": "".join(f">t{k}\nAR{'N' if k % 2 else 'D'}\n" for k in range(201)).encode()}[scenario])
        else:
            (ind / name).write_text(to_fasta(next(alns)))
    order = [os.path.basename(p) for p in glob(f"{ind}/*")]              # this process's listing order (the CLI's own)
    want = sorted(os.path.splitext(n)[0] + ".phy" for n in order[:order.index(g["offender"])])
    r = _run(repo, [os.path.join(repo, "models/pf_base.ckpt"), str(ind), "-o", str(tmp_path / "o")])
    assert r.returncode != 0
    assert sorted(os.listdir(tmp_path / "o")) == want
    last = [ln for ln in r.stderr.strip().splitlines() if ln.strip()][-1]
    assert last.split(":")[0] == g["exception"] and last == g["last_line"].replace("<in>", str(ind))


def test_cli_bucketed_batches_equal_serial_order(repo, tmp_path):
    """All 20 reference test MSAs (4 shapes x 5): the default scheduler (shape buckets, native I/O)
    writes the same files as one-alignment-per-launch with the pure-Python parser/writer, and both
    match the reference's distances (golden e2e outputs) to the parity bar."""
    ind = os.path.join(repo, "data/testdata/msas")
    a, b = tmp_path / "auto", tmp_path / "serial"
    r = _run(repo, [os.path.join(repo, "models/pf.ckpt"), ind, "-o", str(a), "--bench"])
    assert r.returncode == 0, r.stderr
    rep = json.loads([l for l in r.stderr.splitlines() if l.startswith("{")][-1])
    assert rep["alignments"] == 20 and rep["launches"] == 4 and len(rep["shapes"]) == 4
    r = _run(repo, [os.path.join(repo, "models/pf.ckpt"), ind, "-o", str(b), "--batch", "1", "--python-io", "-t"])
    assert r.returncode == 0, r.stderr
    gold = np.load(os.path.join(repo, "tests/golden/e2e_testdata.npz"))
    names = sorted(os.listdir(a))
    assert names == sorted(n for n in os.listdir(b) if n.endswith(".phy")) and len(names) == 20
    for name in names:
        ids_a, dm_a = _read_phy(a / name)
        ids_b, dm_b = _read_phy(b / name)
        assert ids_a == ids_b
        # results do not depend on batching (tests/test_gpu_parity.py::test_batch_invariance...): the two
        # runs print the same digits
        assert np.array_equal(dm_a, dm_b)
        n = len(ids_a)
        ref = gold["pf/" + name[:-4]]
        assert np.abs(dm_a[np.triu_indices(n, 1)] - ref).max() <= 1e-4

    # end-to-end tree check: NJ trees from the GPU distances have the topology of
    # NJ trees from the reference's distances, and sit as close to the true trees (tests/test_treecmp.py)
    from phyloformer_amd import treecmp
    from phyloformer_amd.nj import neighbor_joining
    from phyloformer_amd.phylip import vec_to_matrix
    rf_vs_ref, nrf_vs_true = 0, []
    for name in names:
        stem = name[:-4]
        ids, _dm = _read_phy(b / name)
        mine = treecmp.parse_newick(open(b / f"{stem}.nj.nwk").read())
        ref = treecmp.parse_newick(neighbor_joining(vec_to_matrix(gold["pf/" + stem], len(ids)).astype(np.float64), ids))
        true = treecmp.parse_newick(open(os.path.join(repo, "data/testdata/trees", stem + ".nwk")).read())
        rf_vs_ref += treecmp.robinson_foulds(mine, ref)[0]
        nrf_vs_true.append(treecmp.robinson_foulds(mine, true)[1])
    # NJ has near-ties on this set: uniform noise of 1e-6 on the reference's own distances — its fp32
    # rounding level — already moves 6-8 splits (of 1,280) in one or two of the 20 trees, so equality of
    # topologies is not a meaningful bar; <= 1 % of the splits and the same distance to the true trees is.
    print("RF(mine, ref) summed over 20 trees:", rf_vs_ref, " mean nRF vs true:", np.mean(nrf_vs_true))
    assert rf_vs_ref <= 12
    assert abs(np.mean(nrf_vs_true) - 0.1857) <= 0.01


def test_cli_file_sharding_over_worker_processes(repo, tmp_path):
    """--devices: one worker process per listed device, each on its share of the files (here the same
    GPU twice, which exercises the process fan-out, --worker slicing and report aggregation)."""
    ind = os.path.join(repo, "data/testdata/msas")
    out = tmp_path / "out"
    r = _run(repo, [os.path.join(repo, "models/pf.ckpt"), ind, "-o", str(out), "--devices", "0,0", "--bench"])
    assert r.returncode == 0, r.stderr
    rep = json.loads([l for l in r.stderr.splitlines() if l.startswith("{")][-1])
    assert rep["alignments"] == 20 and len(rep["workers"]) == 2
    assert sorted(w["alignments"] for w in rep["workers"]) == [10, 10]
    assert len([n for n in os.listdir(out) if n.endswith(".phy")]) == 20
    gold = np.load(os.path.join(repo, "tests/golden/e2e_testdata.npz"))
    for name in sorted(os.listdir(out)):
        ids, dm = _read_phy(out / name)
        assert np.abs(dm[np.triu_indices(len(ids), 1)] - gold["pf/" + name[:-4]]).max() <= 1e-4


def test_cli_site_sharded_with_a_real_communicator(repo, tmp_path):
    """`--shard sites` on the GPU.  One box has one GPU, so the rank is alone
    (PF_CLI_FORCE_RCCL=1: a real single-rank RCCL communicator pair) - but it runs the path the ranks of
    `--devices 0,..,7 --shard sites` run: SiteShardedRunner, pf_forward_sharded, 14 collectives per batched launch
    on two streams.  Byte-identical files to the plain run; the two-rank plumbing itself is tests/test_scheduler.py."""
    ind = os.path.join(repo, "data/testdata/msas")
    a, b = tmp_path / "sites", tmp_path / "plain"
    r = _run(repo, [os.path.join(repo, "models/pf.ckpt"), ind, "-o", str(a), "--shard", "sites", "--bench", "-t"],
             env=dict(os.environ, PF_CLI_FORCE_RCCL="1"))
    assert r.returncode == 0, r.stderr
    rep = json.loads([l for l in r.stderr.splitlines() if l.startswith("{")][-1])
    assert rep["alignments"] == 20 and rep["launches"] == 4 and rep["site_sharded_over"] == 1
    assert rep["collectives"] == 4 * 14                     # four shape buckets of five alignments: two halves each
    r = _run(repo, [os.path.join(repo, "models/pf.ckpt"), ind, "-o", str(b), "-t"])
    assert r.returncode == 0, r.stderr
    names = sorted(os.listdir(b))
    assert names == sorted(os.listdir(a)) and len(names) == 40
    for n in names:
        assert (a / n).read_bytes() == (b / n).read_bytes(), n
    # two ranks on ONE GPU: RCCL refuses, the ranks agree and fall back to sharding the files - everything is written
    c = tmp_path / "two"
    r = _run(repo, [os.path.join(repo, "models/pf.ckpt"), ind, "-o", str(c), "--devices", "0,0", "--shard", "sites", "--bench"])
    assert r.returncode == 0, r.stderr
    rep = json.loads([l for l in r.stderr.splitlines() if l.startswith("{") and '"workers"' in l][-1])
    if rep["shard"] == "sites":          # a box where RCCL takes two ranks on one device: the real thing ran
        assert rep["alignments"] == 20 and all(w["site_sharded_over"] == 2 for w in rep["workers"])
    else:
        assert "site-sharding unavailable" in r.stderr and rep["alignments"] == 20
    for n in (x for x in names if x.endswith(".phy")):
        ids, dm = _read_phy(c / n)
        _ids, dm_b = _read_phy(b / n)
        assert np.abs(dm - dm_b).max() <= 2e-5
