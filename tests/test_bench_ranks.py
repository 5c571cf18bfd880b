"""bench.py's rank logic (site slicing, agreed fallback, JSON assembly) at world size 2 over the TCP
rendezvous with a fake engine - the N > 1 launch path the driver runs on an 8-GPU node, minus the GPU."""
import io
import json
import multiprocessing as mp
import os
import sys
import uuid

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


from helpers.fake_engine import FakeEngine, FakeWeights as _W  # noqa: E402


def _worker(rank, world, key, tmp, fail_comm_on, q):
    sys.path.insert(0, REPO)
    import bench
    from phyloformer_amd.rendezvous import TcpGroup
    args = bench.parse_args(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "2", "--n-seqs", "6",
                             "--n-sites", "45", "--no-power"])
    log, out = [], io.StringIO()
    with TcpGroup(rank, world, key=key, directory=tmp, timeout=30) as g:
        _value, ok = bench.run(args, rank, world, rank, g, lambda dev: FakeEngine(rank, log, fail_comm_on), _W(), out=out)
        assert ok
    q.put((rank, log, out.getvalue()))


@pytest.mark.parametrize("fail_comm_on", [None, 1])
def test_bench_rank_logic_world2(fail_comm_on, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = "pf_bench_" + uuid.uuid4().hex
    procs = [ctx.Process(target=_worker, args=(r, 2, key, str(tmp_path), fail_comm_on, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r, (log, text)) for r, log, text in (q.get(timeout=120) for _ in range(2)))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    line = json.loads(res[0][1])
    assert res[1][1] == "", "only rank 0 prints"
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["value_host_buffers"] > 0 and line["cpu_baseline"] is None
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - line["config"]["global_batch"]) < 1e-2 * line["config"]["global_batch"]
    if fail_comm_on is None:
        # sites 0..22 on rank 0, 23..44 on rank 1 (ceil split), global batch = batch x world
        for rank, (lo, hi) in enumerate([(0, 23), (23, 45)]):
            steps = [e for e in res[rank][0] if e[0] == "sharded" and e[2] == 6]
            # warm-up + 3 timed steps on two streams, then 1 + 3 with the batch on one stream (roofline region)
            assert len(steps) == 8 and all(e == ("sharded", 4, 6, lo, hi, 45) for e in steps)
            # parity leg: the committed 60 x 500 and 60 x 2000 reference alignments, twice each (two half-batches),
            # through the same entry point with this rank's site range; then configs[3] timed: warm-up + 3 steps
            gold = [e for e in res[rank][0] if e[0] == "sharded" and e[2] == 60]
            assert gold[0] == ("sharded", 2, 60, 250 * rank, 250 * (rank + 1), 500)
            assert gold[1] == ("sharded", 2, 60, 1000 * rank, 1000 * (rank + 1), 2000)
            assert gold[2:6] == [("sharded", 2, 60, 1000 * rank, 1000 * (rank + 1), 2000)] * 4
            # the strong-scaling case: ONE 60 x 2000 alignment per step, warm-up + 3 timed steps
            assert gold[6:] == [("sharded", 1, 60, 1000 * rank, 1000 * (rank + 1), 2000)] * 4
            opts = [e[1:] for e in res[rank][0] if e[0] == "opt" and e[1] in ("two_streams", "overlap")]
            assert opts == [("two_streams", 1), ("overlap", 1), ("two_streams", 0), ("overlap", 0),
                            ("two_streams", 1), ("overlap", 1)]
            assert ("h2d", (4, 6, hi - lo)) in res[rank][0]
        assert line["config"]["parallelism"] == "sites-sharded x2" and line["config"]["global_batch"] == 4
        assert line["config"]["rccl"]["library"].endswith("librccl.so.1")
        c3 = line["configs"]["configs[3] 60x2000 sites-sharded x2"]
        assert c3["sites_per_rank"] == 1000 and c3["global_batch"] == 2 and c3["alignments_per_s"] > 0 and c3["max_abs_err"] == 0.0
        one = line["configs"]["60x2000 x1 sites-sharded x2"]
        assert one["global_batch"] == 1 and one["scaling"] == "strong" and one["sites_per_rank"] == 1000
        assert one["ms_per_alignment"] > 0 and one["ranks_bit_identical"] is True and one["timed_steps"] == 3
        cases = line["parity"]["cases"]
        assert set(cases) == {"configs[2] 60x500", "configs[3] 60x2000"}
        assert all(c["entry_point"] == "pf_forward_sharded_device" and c["collectives"] == 14 and c["ok"] for c in cases.values())
        assert line["roofline"]["launches"] == 6 * 8 and line["roofline"]["traffic_source"].startswith("profiles/")
        assert line["value_one_stream"] > 0 and "one stream" in line["roofline"]["schedule"]
    else:
        # rank 1's communicator failed: BOTH ranks destroy theirs and shard whole alignments instead
        for rank in (0, 1):
            log = res[rank][0]
            assert ("comm_destroy",) in log
            assert not [e for e in log if e[0] == "sharded"]
            assert len([e for e in log if e == ("plain", 2, 6, 45)]) == 8
            # the parity leg follows the fallback: whole goldens through pf_forward_device on every rank
            assert ("plain", 2, 60, 500) in log and ("plain", 2, 60, 2000) in log
        assert line["config"]["parallelism"] == "alignments-sharded x2" and line["config"]["global_batch"] == 4
        assert "RCCL init failed" in line["config"]["note"]
        assert all(c["entry_point"] == "pf_forward_device" and c["collectives"] == 0 for c in line["parity"]["cases"].values())
        assert "configs[3] 60x2000 alignments-sharded x2" in line["configs"]
    assert line["max_abs_err"] == 0.0 and line["max_abs_err_ok"] is True and line["ranks_bit_identical"] is True
    assert line["parity"]["bound"] == 1e-4 and line["n_gpus"] == 2 and "scaling_result" not in line


def _run_bench(argv, env_extra, timeout=120):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra)
    env["PYTHONPATH"] = os.pathsep.join([os.path.dirname(os.path.abspath(__file__)), REPO, env.get("PYTHONPATH", "")])
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, env=env, capture_output=True,
                          text=True, timeout=timeout)


@pytest.mark.parametrize("fail_comm_on", ["", "1"])
def test_bench_self_launch_world2(fail_comm_on, tmp_path):
    """`python3 bench.py --gpus 2` with no launcher environment - the form the driver uses - starts its own two
    ranks (fresh children, rendezvous over 127.0.0.1), relays rank 0's one JSON line and exits 0.  Fake engine:
    this is the launch / rank logic, not the GPU."""
    res = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2", "--n-seqs", "6", "--n-sites", "45",
                      "--no-power", "--launch-timeout", "60"],
                     {"PF_BENCH_ENGINE_FACTORY": "helpers.fake_engine:make_for_bench", "PF_FAKE_LOG_DIR": str(tmp_path),
                      "PF_FAKE_FAIL_COMM_ON": fail_comm_on, "TMPDIR": str(tmp_path)})
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["config"]["global_batch"] == 4
    assert line["metric"].startswith("alignments/sec, 6-leaf/45-site") and line["config"]["workload"].startswith("not a BASELINE shape")
    logs = [json.load(open(tmp_path / f"rank{r}.log")) for r in range(2)]
    if fail_comm_on == "":
        assert line["config"]["parallelism"] == "sites-sharded x2" and line["config"]["n_ranks_in_comm"] == 2
        assert line["config"]["collectives_per_step"] == 14 and line["config"]["communicators"] == 2
        assert line["config"]["reserve_cus"] == 8 and line["config"]["rccl"]["version"] == 22707
        assert line["config"]["rung"] == {"index": 1, "name": "sites, two streams / two communicators", "abandoned": []}
        assert line["config"]["rccl_max_nchannels"] is None          # RCCL's default unless asked for (--rccl-max-nchannels)
        assert line["max_abs_err_ok"] is True and line["ranks_bit_identical"] is True
        for rank, (lo, hi) in enumerate([(0, 23), (23, 45)]):
            assert ["sharded", 4, 6, lo, hi, 45] in logs[rank] and ["comm_init", rank, 2] in logs[rank]
    else:
        assert line["config"]["parallelism"] == "alignments-sharded x2" and "RCCL init failed" in line["config"]["note"]
        assert line["config"]["n_ranks_in_comm"] == 0 and line["config"]["collectives_per_step"] == 0
        assert "falling back" in res.stderr


def test_bench_self_launch_reports_a_dead_rank(tmp_path):
    """A rank that dies takes the launch down with a non-zero exit code instead of leaving the others waiting."""
    res = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1", "--n-seqs", "4", "--n-sites", "33",
                      "--no-power", "--launch-timeout", "60"],
                     {"PF_BENCH_ENGINE_FACTORY": "helpers.fake_engine:make_dying_on_rank1", "TMPDIR": str(tmp_path)})
    assert res.returncode != 0 and res.stdout.strip() == ""
    assert "rank 1 exited with code 7" in res.stderr


def test_bench_self_launch_watchdog(tmp_path):
    res = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1", "--n-seqs", "4", "--n-sites", "33",
                      "--no-power", "--launch-timeout", "3"],
                     {"PF_BENCH_ENGINE_FACTORY": "helpers.fake_engine:make_hanging", "TMPDIR": str(tmp_path)})
    assert res.returncode == 124 and "watchdog" in res.stderr


def test_slow_but_progressing_ranks_are_not_abandoned(tmp_path):
    """the limits of the ladder were fixed (75 / 150 / 400 s) whatever --steps is, and a timed region of K
    asynchronous steps ends in ONE synchronisation - one long silence.  The ranks now publish, from the step time
    measured in the warm-up, how long the next region and the whole run should take; the supervisors stretch the
    stall / rung / overall limits to twice that.  Here a step takes 0.15 s and a region of 40 steps 6 s against
    --stall-timeout 2 / --rung-timeout 10 / --launch-timeout 20: the run must finish on rung 1, not be declared
    stalled."""
    res = _run_bench(["--gpus", "2", "--steps", "40", "--warmup", "3", "--batch", "2", "--n-seqs", "6", "--n-sites", "45",
                      "--no-power", "--no-parity", "--no-configs", "--stall-timeout", "2", "--rung-timeout", "10",
                      "--launch-timeout", "20"],
                     {"PF_BENCH_ENGINE_FACTORY": "helpers.fake_engine:make_slow", "TMPDIR": str(tmp_path)}, timeout=200)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][0])
    assert line["config"]["rung"]["index"] == 1 and line["config"]["rung"]["abandoned"] == []
    assert line["steps"] == 40 and line["ms_per_step"] > 100


def test_a_stopped_supervisor_takes_its_rank_with_it(tmp_path):
    """torchrun stops its workers with SIGTERM; a supervisor that just died used to leave the rank it had
    started behind - parked in a collective, holding its GPU.  Two supervisors in the external-launcher form, ranks
    that hang; SIGTERM to the supervisors: their children are gone within seconds."""
    import signal
    import subprocess
    import time
    import psutil
    sys.path.insert(0, REPO)
    import bench
    port = str(bench.free_port())
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update({"PF_BENCH_ENGINE_FACTORY": "helpers.fake_engine:make_hanging", "TMPDIR": str(tmp_path), "WORLD_SIZE": "2",
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "PF_RUN_ID": uuid.uuid4().hex,
                "PYTHONPATH": os.pathsep.join([os.path.dirname(os.path.abspath(__file__)), REPO])})
    sups = [subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                              "--no-power"], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.DEVNULL,
                             stderr=subprocess.PIPE, text=True) for r in range(2)]
    try:
        kids, t0 = [], time.time()
        while len(kids) < 2 and time.time() - t0 < 60:
            kids = [c for p in sups for c in psutil.Process(p.pid).children()]
            time.sleep(0.2)
        assert len(kids) == 2, "both supervisors start a rank"
        for p in sups:
            p.send_signal(signal.SIGTERM)
        for p in sups:
            assert p.wait(timeout=30) == 128 + signal.SIGTERM
        gone, alive = psutil.wait_procs(kids, timeout=15)
        assert not alive, alive
    finally:
        for p in sups:
            if p.poll() is None:
                p.kill()


def test_workload_label_follows_the_shape():
    sys.path.insert(0, REPO)
    import bench
    m, w = bench.workload_label(60, 2000, "models/pf.ckpt")
    assert "60-leaf/2000-site" in m and w.startswith("configs[3]")
    assert bench.workload_label(60, 500, "pf.ckpt")[1].startswith("configs[2]")
    assert bench.workload_label(200, 500, "pf_indel.ckpt")[1].startswith("configs[4]")
    assert bench.workload_label(20, 200, "pf.ckpt")[1].startswith("configs[1]")


def test_power_sampler_is_evidence_only():
    """No GPU (or no librocm_smi64): the sampler turns itself off, the bench goes on; nothing is forked."""
    sys.path.insert(0, REPO)
    import bench
    with bench.PowerSampler(period=0.01) as s:
        pass
    assert s.summary() is None or "median_w" in s.summary()
    src = open(os.path.join(REPO, "bench.py")).read()
    assert '"rocm-smi"' not in src and "['rocm-smi" not in src, "power is read in-process, not through the rocm-smi script"


LADDER_ARGS = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--n-seqs", "6", "--n-sites", "45", "--no-power"]


def test_ladder_a_stalled_collective_falls_to_the_one_stream_rung(tmp_path):
    """a collective that never completes on rung 1 (two streams / two communicators) must
    not end the run without a number.  The launcher notices that no rank reports progress, kills exactly those
    children, starts FRESH ones with --one-stream and relays their line, which says which rung ran and why."""
    res = _run_bench(LADDER_ARGS + ["--stall-timeout", "4", "--rung-timeout", "40", "--launch-timeout", "100"],
                     {"PF_BENCH_ENGINE_FACTORY": "helpers.fake_engine:make_hanging_on_rung1", "PF_FAKE_LOG_DIR": str(tmp_path),
                      "TMPDIR": str(tmp_path)}, timeout=150)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip())
    rung = line["config"]["rung"]
    assert rung["index"] == 2 and "one stream" in rung["name"]
    assert len(rung["abandoned"]) == 1 and rung["abandoned"][0]["rung"] == 1
    assert "no progress" in rung["abandoned"][0]["why"] and "warm-up" not in rung["abandoned"][0]["why"]
    assert "rank 0: communicators" in rung["abandoned"][0]["why"]          # where the ranks were stuck
    assert line["config"]["parallelism"] == "sites-sharded x2" and line["max_abs_err_ok"] is True
    assert "one stream" in line["roofline"]["schedule"]
    logs = [json.load(open(tmp_path / f"rank{r}.log")) for r in range(2)]       # written by the rung that finished
    assert all(["opt", "two_streams", 1] not in lg for lg in logs)


def test_ladder_down_to_whole_alignments(tmp_path):
    """Rung 1 stalls, rung 2 loses a rank: rung 3 shards whole alignments (no collective) and still delivers."""
    res = _run_bench(LADDER_ARGS + ["--stall-timeout", "4", "--rung-timeout", "40", "--launch-timeout", "120"],
                     {"PF_BENCH_ENGINE_FACTORY": "helpers.fake_engine:make_failing_until_rung3", "PF_FAKE_LOG_DIR": str(tmp_path),
                      "TMPDIR": str(tmp_path)}, timeout=180)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip())
    rung = line["config"]["rung"]
    assert rung["index"] == 3 and [h["rung"] for h in rung["abandoned"]] == [1, 2]
    assert "rank 1 exited with code 9" in rung["abandoned"][1]["why"]
    assert line["config"]["parallelism"] == "alignments-sharded x2" and line["config"]["collectives_per_step"] == 0
    assert line["n_gpus"] == 2 and line["max_abs_err_ok"] is True


def test_a_failed_parity_bound_prints_the_line_and_exits_non_zero(tmp_path):
    """The metric is alignments/s AND max-abs error: a rank whose result is off by 1e-3 makes the run exit with
    code 3 - after the line, which carries the evidence - and no further rung is tried."""
    res = _run_bench(LADDER_ARGS + ["--launch-timeout", "100"],
                     {"PF_BENCH_ENGINE_FACTORY": "helpers.fake_engine:make_for_bench", "PF_FAKE_LOG_DIR": str(tmp_path),
                      "PF_FAKE_PARITY_ERROR": "1", "TMPDIR": str(tmp_path)})
    assert res.returncode == 3, res.stderr[-2000:]
    line = json.loads(res.stdout.strip())
    assert line["max_abs_err_ok"] is False and line["ranks_bit_identical"] is False
    assert abs(line["max_abs_err"] - 1e-3) < 1e-5 and line["config"]["rung"]["index"] == 1
    assert "PARITY FAILED" in res.stderr and "rung 2" not in res.stderr


def test_more_ranks_than_devices_is_refused_unless_allowed(tmp_path):
    """`--gpus 2` on one device used to publish a weak-scaling number measured on a shared GPU."""
    env = {"PF_BENCH_ENGINE_FACTORY": "helpers.fake_engine:make_for_bench", "PF_FAKE_LOG_DIR": str(tmp_path),
           "PF_BENCH_DEVICE": "0", "TMPDIR": str(tmp_path)}
    res = _run_bench(LADDER_ARGS + ["--launch-timeout", "100"], env)
    assert res.returncode == 4 and res.stdout.strip() == "" and "--allow-shared-devices" in res.stderr
    assert "rung 2" not in res.stderr                      # another schedule does not add a GPU: no further rung
    res = _run_bench(LADDER_ARGS + ["--launch-timeout", "100", "--allow-shared-devices"], env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip())
    assert line["n_gpus"] == 1 and line["n_ranks"] == 2 and line["ranks_per_device"] == 2 and line["scaling_result"] is False
    assert "NOT a scaling result" in line["config"]["note"]


def _torchrun(argv, env_extra, timeout=240):
    """bench.py the way the task statement says the driver starts an N > 1 run."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra)
    env["PYTHONPATH"] = os.pathsep.join([os.path.dirname(os.path.abspath(__file__)), REPO, env.get("PYTHONPATH", "")])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py")] + argv
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("factory", ["make_for_bench", "make_hanging_on_rung1"])
def test_under_torchrun_every_process_supervises_a_fresh_rank(factory, tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` (the task statement's N > 1 form): the
    processes torchrun starts become supervisors, the ranks are their fresh children, and the ladder works as in the
    self-launched form - a collective that never completes on rung 1 ends as a rung-2 line, not as a hang."""
    res = _torchrun(LADDER_ARGS + ["--stall-timeout", "4", "--rung-timeout", "40", "--launch-timeout", "100"],
                    {"PF_BENCH_ENGINE_FACTORY": f"helpers.fake_engine:{factory}", "PF_FAKE_LOG_DIR": str(tmp_path),
                     "TMPDIR": str(tmp_path)})
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "sites-sharded x2" and line["max_abs_err_ok"] is True
    if factory == "make_for_bench":
        assert line["config"]["rung"] == {"index": 1, "name": "sites, two streams / two communicators", "abandoned": []}
        assert line["config"]["collectives_per_step"] == 14
    else:
        rung = line["config"]["rung"]
        assert rung["index"] == 2 and rung["abandoned"][0]["rung"] == 1 and "no progress" in rung["abandoned"][0]["why"]
        assert "rung 1" in res.stderr
