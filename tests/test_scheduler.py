"""CPU tests of the CLI scheduler (phyloformer_amd/scheduler.py) with a stand-in engine:
shape bucketing, batch sizes, output files, error propagation, file sharding across workers."""
import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from phyloformer_amd import fasta, scheduler
from phyloformer_amd.phylip import vec_to_phylip


class FakeEngine:
    """forward(uint8 [B, N, L]) -> float32 [B, P]: a cheap deterministic function of each alignment."""

    def __init__(self):
        self.calls = []

    def forward(self, idx):
        idx = np.asarray(idx)
        assert idx.ndim == 3 and idx.dtype == np.uint8
        self.calls.append(idx.shape)
        B, N, L = idx.shape
        i, j = np.triu_indices(N, 1)
        return (idx[:, i, :] != idx[:, j, :]).mean(axis=2).astype(np.float32)


def _write_fasta(path, idx, prefix="s"):
    with open(path, "wb") as fh:
        for k, row in enumerate(idx):
            fh.write(b">%s%d\n" % (prefix.encode(), k) + bytes(fasta.ALPHABET[c] for c in row) + b"\n")


def _make_dir(tmp_path, shapes, seed=0):
    rng = np.random.default_rng(seed)
    d = tmp_path / "in"
    d.mkdir()
    truth = {}
    for k, (n, l) in enumerate(shapes):
        idx = rng.integers(0, 22, (n, l)).astype(np.uint8)
        _write_fasta(d / f"aln{k:03d}.fa", idx)
        truth[f"aln{k:03d}"] = idx
    return d, truth


@pytest.mark.parametrize("native_io", [True, False])
@pytest.mark.parametrize("batch", [0, 1, 3])
def test_directory_runner_buckets_by_shape_and_writes_everything(tmp_path, batch, native_io):
    shapes = [(4, 30), (6, 20), (4, 30), (5, 11), (6, 20), (4, 30), (4, 30), (6, 20), (4, 31)]
    d, truth = _make_dir(tmp_path, shapes)
    out = tmp_path / "out"
    out.mkdir()
    eng = FakeEngine()
    paths = sorted(str(p) for p in d.iterdir())
    stats = scheduler.DirectoryRunner(eng, str(out), batch=batch, io_threads=3, native_io=native_io).run(paths)
    assert stats["alignments"] == len(shapes)
    assert stats["shapes"] == {"4x30": 4, "6x20": 3, "5x11": 1, "4x31": 1}
    assert all(len({s[1:]}) == 1 for s in eng.calls)                      # one shape per launch
    if batch == 1:
        assert len(eng.calls) == len(shapes)
    elif batch == 3:
        assert sorted(s[0] for s in eng.calls) == [1, 1, 1, 3, 3]         # 4x30: 3+1, 6x20: 3, singles
    else:
        assert len(eng.calls) == 4                                        # auto: one launch per shape
    ref = FakeEngine()
    for stem, idx in truth.items():
        ids = [f"s{k}" for k in range(idx.shape[0])]
        _dm, text = vec_to_phylip(ref.forward(idx[None])[0], ids)
        assert (out / f"{stem}.phy").read_text() == text


def test_directory_runner_trees_and_errors(tmp_path):
    d, _truth = _make_dir(tmp_path, [(5, 12), (5, 12)])
    out = tmp_path / "out"
    out.mkdir()
    paths = sorted(str(p) for p in d.iterdir())
    scheduler.DirectoryRunner(FakeEngine(), str(out), trees=True).run(paths)
    nwk = (out / "aln000.nj.nwk").read_text()
    assert nwk.endswith(";\n") and nwk.count(",") == 4
    # a non-FASTA entry aborts with the reference's message (infer_alns.py:100-103)
    (d / "notes.txt").write_text("x")
    with pytest.raises(ValueError, match="Input files must be fasta files"):
        scheduler.DirectoryRunner(FakeEngine(), str(out)).run(sorted(str(p) for p in d.iterdir()))
    (d / "notes.txt").unlink()
    # parser errors surface with the reference's exception types (data.py:26)
    (d / "bad.fa").write_bytes(b">a\nARNDB\n>b\nARNDC\n")
    with pytest.raises(KeyError):
        scheduler.DirectoryRunner(FakeEngine(), str(out)).run(sorted(str(p) for p in d.iterdir()))


@pytest.mark.parametrize("native_io", [True, False])
@pytest.mark.parametrize("scenario", ["bad_extension", "bad_residue", "too_many_seqs", "single_sequence"])
def test_bad_entry_side_effects_are_the_references(tmp_path, golden, scenario, native_io):
    """VERDICT r04 / next 7.  The reference handles one directory entry after the other (infer_alns.py:97-117): when it
    meets an entry without a FASTA extension (ValueError, :100-103), a file that does not parse (KeyError from
    load_alignment, data.py:26), an alignment of more than 200 sequences (ValueError from adaptable_seq2pair inside the
    forward, model.py:24-28) or of a single one (RuntimeError from attention.py:193: no pair to view - round 6), every
    entry listed BEFORE it has its .phy and nothing after it does.  The fixture
    (oracle/gen_golden_cli_errors.py) holds what the real CLI left behind; this build's runner, batching and
    prefetching notwithstanding, must leave the same set for the same processing order, and raise the same."""
    g = json.load(open(os.path.join(REPO, "tests", "golden", "cli_bad_entry.json")))[scenario]
    order, offender = g["listing_order"], g["offender"]
    k = order.index(offender)
    assert g["outputs"] == sorted(os.path.splitext(n)[0] + ".phy" for n in order[:k])   # the reference's rule
    d = tmp_path / "in"
    d.mkdir()
    rng = np.random.default_rng(3)
    for name in order:
        if name == offender:
            (d / name).write_bytes({"bad_extension": b"not an alignment\n", "bad_residue": b">s0\nARNDB\n>s1\nARNDC\n",
                                    "too_many_seqs": "".join(f">t{k}\nAR{'N' if k % 2 else 'D'}\n" for k in range(201)).encode(),
                                    "single_sequence": b">s0\nARNDCQEGHILK\n"}[scenario])
        else:
            _write_fasta(d / name, rng.integers(0, 20, (5, 12)).astype(np.uint8))
    out = tmp_path / "out"
    out.mkdir()
    exc = {"ValueError": ValueError, "KeyError": KeyError, "RuntimeError": RuntimeError}[g["exception"]]
    with pytest.raises(exc) as info:
        scheduler.DirectoryRunner([FakeEngine(), FakeEngine()], str(out), io_threads=3, native_io=native_io,
                                  batch=2).run([str(d / n) for n in order])
    assert sorted(os.listdir(out)) == g["outputs"]
    want = g["last_line"].split(": ", 1)[1].replace("<in>", str(d))
    assert (str(info.value.args[0]) if exc is KeyError else str(info.value)) == want


def test_native_bulk_io_equals_python_io(tmp_path):
    """The native pipeline (FILES_PER_LOAD files per pf_fasta_batch_load call, pf_fasta_batch_gather into the launch
    buffer, pf_phylip_write_batch with the ids the batch object holds) against the per-file Python mirrors of the
    reference: byte-identical outputs over several load chunks, CRLF files, multi-line records, ids with blanks."""
    rng = np.random.default_rng(9)
    d = tmp_path / "in"
    d.mkdir()
    old = scheduler.FILES_PER_LOAD
    scheduler.FILES_PER_LOAD = 7                      # 40 files -> 6 native load calls
    try:
        for k in range(40):
            n, l = [(4, 30), (6, 20), (5, 33)][k % 3]
            idx = rng.integers(0, 22, (n, l)).astype(np.uint8)
            with open(d / f"x{k:02d}.fa", "wb") as fh:
                eol = b"\r\n" if k % 5 == 0 else b"\n"
                for r, row in enumerate(idx):
                    seq = bytes(fasta.ALPHABET[c] for c in row)
                    fh.write(b">seq %d of %d\t " % (r, k) + eol + seq[:l // 2] + eol + b"  " + seq[l // 2:] + b" " + eol)
        paths = sorted(str(p) for p in d.iterdir())
        outs = {}
        for native in (True, False):
            out = tmp_path / f"out{int(native)}"
            out.mkdir()
            stats = scheduler.DirectoryRunner([FakeEngine(), FakeEngine()], str(out), io_threads=3, native_io=native).run(paths)
            assert stats["alignments"] == 40
            outs[native] = {f: (out / f).read_bytes() for f in sorted(os.listdir(out))}
        assert len(outs[True]) == 40 and outs[True] == outs[False]
        assert outs[True]["x00.phy"].startswith(b"4\nseq 0 of 0 0.0000000000 ")
        # --trees stays on the native pipeline (round 6: pf_phylip_write_batch joins and writes <stem>.nj.nwk as well):
        # the same bytes as the per-file Python path (phylip.vec_to_phylip + nj.neighbor_joining)
        trees = {}
        for native in (True, False):
            out = tmp_path / f"tree{int(native)}"
            out.mkdir()
            r = scheduler.DirectoryRunner([FakeEngine(), FakeEngine()], str(out), io_threads=3, native_io=native, trees=True)
            fed = []
            r._feed_native = (lambda *a, _f=r._feed_native: fed.append(1) or _f(*a))
            assert r.run(paths)["alignments"] == 40 and bool(fed) == native
            trees[native] = {f: (out / f).read_bytes() for f in sorted(os.listdir(out))}
        assert len(trees[True]) == 80 and trees[True] == trees[False]
        assert all(trees[True][f] == outs[True][f] for f in outs[True])
        assert trees[True]["x00.nj.nwk"].startswith(b"(") and trees[True]["x00.nj.nwk"].endswith(b");\n")
    finally:
        scheduler.FILES_PER_LOAD = old


def test_auto_batch_and_slicing(tmp_path):
    assert scheduler.auto_batch(60, 500) == 16
    assert scheduler.auto_batch(200, 500) == 1
    assert scheduler.auto_batch(20, 200) == 372
    assert scheduler.auto_batch(2, 1, max_batch=100) == 100
    d, _ = _make_dir(tmp_path, [(3, 5 + (k % 4)) for k in range(11)])
    paths = [str(p) for p in d.iterdir()]
    parts = [scheduler.slice_paths(paths, r, 3) for r in range(3)]
    assert sorted(sum(parts, [])) == sorted(paths)
    assert max(map(len, parts)) - min(map(len, parts)) <= 1
    assert scheduler.slice_paths(paths, 0, 1) == paths


def test_two_gpu_worker_threads_and_worker_errors(tmp_path):
    shapes = [(4, 30)] * 9 + [(6, 20)] * 5 + [(5, 11)]
    d, truth = _make_dir(tmp_path, shapes)
    out = tmp_path / "out"
    out.mkdir()
    engs = [FakeEngine(), FakeEngine()]
    paths = sorted(str(p) for p in d.iterdir())
    stats = scheduler.DirectoryRunner(engs, str(out), batch=2, io_threads=2).run(paths)
    assert stats["alignments"] == len(shapes) and stats["launches"] == 5 + 3 + 1
    assert len(engs[0].calls) + len(engs[1].calls) == 9
    ref = FakeEngine()
    for stem, idx in truth.items():
        _dm, text = vec_to_phylip(ref.forward(idx[None])[0], [f"s{k}" for k in range(idx.shape[0])])
        assert (out / f"{stem}.phy").read_text() == text

    class Boom(FakeEngine):
        def forward(self, idx):
            raise RuntimeError("device lost")
    with pytest.raises(RuntimeError, match="device lost"):
        scheduler.DirectoryRunner([Boom(), Boom()], str(out), batch=2).run(paths)


# ---- --shard sites: the CLI's multi-rank plumbing at world 2, with real numerics (oracle engine) ----------------
def _cli(argv, env_extra, timeout=300):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra)
    env["PYTHONPATH"] = os.pathsep.join([os.path.dirname(os.path.abspath(__file__)), REPO, env.get("PYTHONPATH", "")])
    return subprocess.run([sys.executable, os.path.join(REPO, "infer_alns.py")] + argv, env=env, capture_output=True,
                          text=True, timeout=timeout)


@pytest.fixture(scope="module")
def small_dir(tmp_path_factory):
    from phyloformer_amd.msa_sim import simulate_batch
    d = tmp_path_factory.mktemp("alns")
    for k, a in enumerate(simulate_batch(3, 6, 41, seed=11)):
        _write_fasta(d / f"a{k}.fa", a)
    for k, a in enumerate(simulate_batch(2, 5, 33, seed=12, gaps=True)):
        _write_fasta(d / f"b{k}.fasta", a)
    _write_fasta(d / "c0.fa", simulate_batch(1, 4, 1, seed=13)[0])      # one site: rank 1 of 2 holds none of it
    return d


def test_cli_site_sharded_world2_writes_what_one_process_writes(small_dir, tmp_path):
    """`infer_alns.py --devices 0,1 --shard sites` - the north star's split behind the north
    star's surface.  Two ranks (fresh children of the CLI, TcpGroup rendezvous) each parse every file, keep their
    block of sites, exchange the row statistics of every block and the final site sums; rank 0 alone writes.  With
    the oracle engine the numerics are real: the .phy and .nwk files are byte-identical to a one-process run that
    evaluates the same two-shard sums (output contract: /root/reference/infer_alns.py:105-123)."""
    env = {"PF_CLI_ENGINE_FACTORY": "helpers.oracle_engine:make", "TMPDIR": str(tmp_path)}
    ckpt = os.path.join(REPO, "models", "pf_base.ckpt")
    r = _cli([ckpt, str(small_dir), "-o", str(tmp_path / "sites"), "-t", "--devices", "0,1", "--shard", "sites", "--batch", "2",
              "--bench"], env)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith("{") and '"workers"' in ln][-1])
    assert rep["shard"] == "sites" and rep["alignments"] == 6 and len(rep["workers"]) == 2
    assert all(w["site_sharded_over"] == 2 and w["alignments"] == 6 and w["launches"] == 4 for w in rep["workers"])
    assert all(w["collectives"] == 7 * 6 for w in rep["workers"])            # n_blocks + 1 per alignment (oracle engine)
    one = _cli([ckpt, str(small_dir), "-o", str(tmp_path / "one"), "-t", "--batch", "1"], dict(env, PF_ORACLE_SHARDS="2"))
    assert one.returncode == 0, one.stderr[-3000:]
    names = sorted(os.listdir(tmp_path / "one"))
    assert names == sorted(os.listdir(tmp_path / "sites")) and len(names) == 12
    for n in names:
        assert (tmp_path / "sites" / n).read_bytes() == (tmp_path / "one" / n).read_bytes(), n
    # and the sharded sums are the plain forward's to fp32 noise
    plain = _cli([ckpt, str(small_dir), "-o", str(tmp_path / "plain")], env)
    assert plain.returncode == 0, plain.stderr[-3000:]
    for n in names:
        if n.endswith(".phy"):
            a = np.array([[float(v) for v in ln.split(" ")[1:]] for ln in (tmp_path / "sites" / n).read_text().splitlines()[1:]])
            b = np.array([[float(v) for v in ln.split(" ")[1:]] for ln in (tmp_path / "plain" / n).read_text().splitlines()[1:]])
            assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(b).max())


def test_cli_site_sharded_falls_back_to_files_when_a_communicator_fails(small_dir, tmp_path):
    """A rank whose RCCL communicator does not come up is named; ALL ranks drop theirs and shard the files instead:
    every output is still written, by whichever rank got the file."""
    env = {"PF_CLI_ENGINE_FACTORY": "helpers.oracle_engine:make", "PF_FAKE_FAIL_COMM_ON": "1", "TMPDIR": str(tmp_path)}
    ckpt = os.path.join(REPO, "models", "pf_base.ckpt")
    r = _cli([ckpt, str(small_dir), "-o", str(tmp_path / "fb"), "--devices", "0,1", "--shard", "sites", "--bench"], env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "site-sharding unavailable (" in r.stderr and "rank 1: RuntimeError" in r.stderr and "sharding the files" in r.stderr
    rep = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith("{") and '"workers"' in ln][-1])
    assert rep["shard"] == "files (fallback)" and rep["alignments"] == 6 and sorted(w["alignments"] for w in rep["workers"]) == [3, 3]
    plain = _cli([ckpt, str(small_dir), "-o", str(tmp_path / "plain")], {k: v for k, v in env.items() if k != "PF_FAKE_FAIL_COMM_ON"})
    assert plain.returncode == 0
    for n in sorted(os.listdir(tmp_path / "plain")):
        assert (tmp_path / "fb" / n).read_bytes() == (tmp_path / "plain" / n).read_bytes(), n


def test_cli_site_sharded_dead_rank_ends_the_run_instead_of_hanging(small_dir, tmp_path):
    """in sites mode the ranks depend on each other.  Rank 1 dies after its first launch (injected
    os._exit); rank 0 is then parked in a collective that can never complete (the stand-in's own timeout is 600 s).
    The launcher polls all children, terminates the survivors and returns non-zero within seconds."""
    import time
    env = {"PF_CLI_ENGINE_FACTORY": "helpers.oracle_engine:make", "PF_FAKE_DIE_RANK": "1", "TMPDIR": str(tmp_path)}
    ckpt = os.path.join(REPO, "models", "pf_base.ckpt")
    t0 = time.time()
    r = _cli([ckpt, str(small_dir), "-o", str(tmp_path / "dead"), "--devices", "0,1", "--shard", "sites", "--batch", "1"], env,
             timeout=200)
    assert r.returncode == 7, r.returncode        # the code of the rank that failed on its own, not a terminated peer's -15
    assert "rank 1 exited with code 7; terminating the other site-sharded ranks" in r.stderr
    assert time.time() - t0 < 120


def test_cli_site_sharded_slow_healthy_run_survives_the_stall_watchdog(small_dir, tmp_path):
    """site-sharded workers print nothing until their report, so a healthy run longer than
    PF_CLI_STALL_TIMEOUT used to be terminated with partial outputs.  Workers now send a heartbeat line while they make
    progress; the launcher counts it as an event and keeps it out of its own stderr.  Six launches of 1 s each against a
    3 s stall limit."""
    env = {"PF_CLI_ENGINE_FACTORY": "helpers.oracle_engine:make", "PF_FAKE_SLOW_S": "1.0", "PF_CLI_STALL_TIMEOUT": "3",
           "PF_CLI_HEARTBEAT": "0.5", "TMPDIR": str(tmp_path)}
    ckpt = os.path.join(REPO, "models", "pf_base.ckpt")
    r = _cli([ckpt, str(small_dir), "-o", str(tmp_path / "slow"), "--devices", "0,1", "--shard", "sites", "--batch", "1"], env,
             timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(os.listdir(tmp_path / "slow")) == 6
    assert scheduler.HEARTBEAT not in r.stderr and "terminating" not in r.stderr


def test_cli_site_sharded_stalled_rank_is_reported_as_a_stall(small_dir, tmp_path):
    """A rank that stops (no line, no exit) does end the run after PF_CLI_STALL_TIMEOUT, and the exit code says why:
    124, not the -15 / 241 of the peers the launcher itself terminated."""
    env = {"PF_CLI_ENGINE_FACTORY": "helpers.oracle_engine:make", "PF_FAKE_HANG_RANK": "1", "PF_CLI_STALL_TIMEOUT": "4",
           "PF_CLI_HEARTBEAT": "0.5", "TMPDIR": str(tmp_path)}
    ckpt = os.path.join(REPO, "models", "pf_base.ckpt")
    r = _cli([ckpt, str(small_dir), "-o", str(tmp_path / "hang"), "--devices", "0,1", "--shard", "sites", "--batch", "1"], env,
             timeout=300)
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])
    assert "no rank made progress for 4 s; terminating the other site-sharded ranks" in r.stderr


def test_stale_partial_buckets_are_launched_not_hoarded(tmp_path):
    """The native pipeline keeps a load call's parsed files alive while any of them waits in a partial shape bucket; a rare
    shape must not pin a long directory in memory: a bucket that has waited STALE_LOADS load calls is launched as it is.
    Same outputs as the Python I/O path (an alignment's result does not depend on its batch)."""
    rng = np.random.default_rng(4)
    d = tmp_path / "in"
    d.mkdir()
    for k in range(60):
        n, l = (7, 9) if k == 3 else (4, 30)               # one rare shape early in the listing
        _write_fasta(d / f"y{k:02d}.fa", rng.integers(0, 22, (n, l)).astype(np.uint8))
    paths = sorted(str(p) for p in d.iterdir())
    old = (scheduler.FILES_PER_LOAD, scheduler.STALE_LOADS)
    scheduler.FILES_PER_LOAD, scheduler.STALE_LOADS = 5, 3   # 12 load calls; the rare bucket goes stale after 3
    try:
        eng = FakeEngine()
        out = tmp_path / "o"
        out.mkdir()
        scheduler.DirectoryRunner(eng, str(out), batch=1000, io_threads=2).run(paths)
        rare = [k for k, s_ in enumerate(eng.calls) if s_[1:] == (7, 9)]
        assert rare and rare[0] < len(eng.calls) - 1, "the rare shape was launched before the end of the directory"
        assert sum(s_[0] for s_ in eng.calls) == 60 and len(os.listdir(out)) == 60
        out2 = tmp_path / "o2"
        out2.mkdir()
        scheduler.DirectoryRunner(FakeEngine(), str(out2), batch=1000, io_threads=2, native_io=False).run(paths)
        for f in os.listdir(out):
            assert (out / f).read_bytes() == (out2 / f).read_bytes()
    finally:
        scheduler.FILES_PER_LOAD, scheduler.STALE_LOADS = old
