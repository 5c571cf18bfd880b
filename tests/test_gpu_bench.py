"""-m gpu: bench.py as the driver invokes it, on the 1-GPU box - the self-launched N > 1 path down to the
agreed fallback, the site-sharded schedule with real (single-rank) RCCL communicators, and the library's
promise that distinct handles are independent (two engines, two host threads, first forwards concurrently)."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(argv, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, env=env, capture_output=True,
                          text=True, timeout=timeout)


GPUS2 = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--n-seqs", "20", "--n-sites", "200",
         "--no-cpu-baseline", "--no-power", "--launch-timeout", "300"]


@pytest.mark.parametrize("pin", [True, False])
def test_bench_gpus2_on_one_gpu_is_refused(pin):
    """one rank per GPU is the contract.  `python3 bench.py --gpus 2` on the 1-GPU box (both ranks
    pinned to device 0, or rank 1 asking for a device 1 that does not exist) ends with exit code 4 on every rank,
    no line, and no further rung of the ladder."""
    res = _bench(GPUS2, {"PF_BENCH_DEVICE": "0"} if pin else {})
    assert res.returncode == 4, res.stderr[-3000:]
    assert res.stdout.strip() == "" and "--allow-shared-devices" in res.stderr and "rung 2" not in res.stderr


def test_bench_gpus2_shared_device_falls_back_together():
    """With --allow-shared-devices the parent starts two ranks that share device 0.  RCCL refuses two ranks on
    one device; the ranks agree on that, destroy their communicators and shard whole alignments instead - one JSON
    line, exit code 0, marked as NOT a scaling result (n_gpus = distinct devices), parity checked on both ranks."""
    res = _bench(GPUS2 + ["--allow-shared-devices"])
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["n_ranks"] == 2 and line["scaling_result"] is False
    assert line["value"] > 0 and line["config"]["workload"].startswith("configs[1]")
    assert line["config"]["rung"]["index"] == 1 and line["config"]["rung"]["abandoned"] == []
    if line["config"]["parallelism"].startswith("alignments"):
        assert "RCCL init failed" in line["config"]["note"] and line["config"]["collectives_per_step"] == 0
    else:       # a box where RCCL accepts two ranks on one device: then the real thing ran
        assert line["config"]["n_ranks_in_comm"] == 2 and line["config"]["collectives_per_step"] == 14
    assert line["max_abs_err_ok"] is True and line["ranks_bit_identical"] is True and line["max_abs_err"] < 2e-5
    assert set(line["parity"]["cases"]) == {"configs[2] 60x500", "configs[3] 60x2000"}
    print("bench --gpus 2 on one GPU:", line["config"], line["max_abs_err"])


def test_bench_force_dist_runs_fourteen_collectives():
    """The N > 1 code path of bench.py (rendezvous, two RCCL communicators, two streams) with one rank."""
    res = _bench(["--force-dist", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-power",
                  "--batch", "4", "--n-seqs", "20", "--n-sites", "200"])
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    cfg = line["config"]
    assert cfg["collectives_per_step"] == 14 and cfg["communicators"] == 2 and cfg["n_ranks_in_comm"] == 1
    assert cfg["reserve_cus"] == 8 and cfg["rccl"]["version"] > 20000 and cfg["rccl"]["library"].endswith(".so.1")
    assert line["roofline"]["launches"] == 12 and line["value"] > 0
    # the parity leg went through the sharded entry point with real RCCL on both streams / communicators
    cases = line["parity"]["cases"]
    assert set(cases) == {"configs[2] 60x500", "configs[3] 60x2000"}
    for c in cases.values():
        assert c["entry_point"] == "pf_forward_sharded_device" and c["collectives"] == 14 and c["ok"] and c["max_abs_err"] < 2e-5
    assert line["max_abs_err_ok"] is True and line["ranks_bit_identical"] is None      # one rank: nothing to compare
    # the strong-scaling case (one 60 x 2000 alignment per step through the sharded entry point): 7 collectives, the
    # bits of the unsharded forward, and the build's identity on the line
    one = line["configs"]["60x2000 x1 sites-sharded x1"]
    assert one["collectives_per_alignment"] == 7 and one["bit_identical_to_pf_forward"] is True and one["scaling"] == "strong"
    assert cfg["build"]["sched_strategy"] == "iterative-ilp" and cfg["build"]["sched_fallback"] is False and len(cfg["build"]["kernel_hash"]) == 16
    c3 = line["configs"]["configs[3] 60x2000 sites-sharded x1"]
    assert c3["sites_per_rank"] == 2000 and c3["alignments_per_s"] > 0 and c3["max_abs_err"] == cases["configs[3] 60x2000"]["max_abs_err"]


def test_two_engines_first_forward_from_two_threads(weights, golden):
    """No process-global mutable state on the launch path: two handles created and driven through their FIRST
    forward by two host threads at once give the bits of a lone engine."""
    from phyloformer_amd.engine import Engine
    g = golden("configs.npz")
    a = g["c2_idx"]
    with Engine(weights("pf"), 0) as e:
        ref = e.forward(a)
    start = threading.Barrier(2)
    out, errs = [None, None], []

    def work(i):
        try:
            with Engine(weights("pf"), 0) as e:
                start.wait(timeout=60)
                out[i] = e.forward(a)              # first launch of every kernel on this handle
                for _ in range(3):
                    assert np.array_equal(e.forward(a), out[i])
        except Exception as exc:  # noqa: BLE001
            errs.append(repr(exc))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errs, errs
    assert np.array_equal(out[0], ref) and np.array_equal(out[1], ref)


def test_bench_default_line_carries_the_contract():
    """`python bench.py` as the driver runs it at N = 1 (shortened: 3 steps, no CPU baseline): ONE JSON line on
    stdout with the contract's keys, the roofline object of the dominant kernel and every other BASELINE
    configuration."""
    res = _bench(["--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout[:2000]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "configs", "value_pcie_inclusive"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "alignments/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "60-leaf/500-site" in d["metric"] and d["config"]["workload"].startswith("configs[2]") and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - d["config"]["global_batch"]) < 0.02 * d["config"]["global_batch"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and r["launches"] == 18
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.05 < r["frac"] < 0.34
    # traffic: the committed PMC figures IF they were taken with this library's kernels, else null with the reason
    assert (r["traffic"] or 0) > 0 or "stale" in r["traffic_source"] or "absent" in r["traffic_source"]
    assert d["config"]["build"]["kernel_hash"] in r["traffic_source"] or r["traffic"] is None
    assert len(d["configs"]) == 5 and all(v["alignments_per_s"] > 0 for v in d["configs"].values())
    assert d["value_pcie_inclusive"] <= d["value"] * 1.05
    # the metric's second half: max-abs error against the reference's outputs for every BASELINE shape
    cases = d["parity"]["cases"]
    assert set(cases) == {"configs[1] 20x200 x3", "configs[2] 60x500", "configs[3] 60x2000", "configs[4] 200x500 gapped"}
    assert all(c["ok"] and c["finite"] and c["max_abs_err"] < 2e-5 for c in cases.values()), cases
    assert d["max_abs_err"] == max(c["max_abs_err"] for c in cases.values()) and d["max_abs_err_ok"] is True
    assert d["power"] is None or d["power"]["device"]["matched_by"].startswith("pci")
    print("parity on the bench line:", {k: c["max_abs_err"] for k, c in cases.items()})


def test_bench_under_torchrun_two_ranks_share_the_gpu():
    """The task statement's N > 1 form - `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` - on
    the 1-GPU box: torchrun's processes become supervisors (no HIP), their fresh children are the ranks; RCCL refuses
    two ranks on one device, the ranks agree and shard whole alignments; supervisor 0 relays the one JSON line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py")] + GPUS2 + ["--allow-shared-devices"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_ranks"] == 2 and line["n_gpus"] == 1 and line["scaling_result"] is False and line["value"] > 0
    assert line["config"]["rung"]["index"] == 1 and line["max_abs_err_ok"] is True and line["ranks_bit_identical"] is True


def test_bench_force_dist_one_stream_is_rung_two():
    """Rung 2 of the ladder (`--one-stream`: one stream, one communicator in use, serial collectives) with real
    single-rank RCCL: 7 collectives per step and per parity case, the same distances."""
    res = _bench(["--force-dist", "--one-stream", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-power",
                  "--no-configs", "--batch", "4", "--n-seqs", "20", "--n-sites", "200"])
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["config"]["collectives_per_step"] == 7 and "one stream" in line["roofline"]["schedule"]
    cases = line["parity"]["cases"]
    assert all(c["collectives"] == 7 and c["ok"] and c["max_abs_err"] < 2e-5 for c in cases.values()), cases
    assert line["configs"] is None and line["max_abs_err_ok"] is True
