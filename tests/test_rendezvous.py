"""Torch-free rendezvous (phyloformer_amd/rendezvous.py) and the agreement logic built on it: real
processes, world sizes 2 and 3, no GPU."""
import multiprocessing as mp
import os
import sys
import uuid

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeEngine:
    """What dist.init_engine_comm needs from an Engine."""

    def __init__(self, rank, fail_uid=False, no_rccl=False):
        self.rank_, self.fail_uid, self.no_rccl, self.inited = rank, fail_uid, no_rccl, None

    def comm_info(self):
        if self.no_rccl:
            raise RuntimeError("librccl.so.1 is bound to another HIP runtime")
        return {"library": "/opt/rocm/lib/librccl.so.1", "version": 22707}

    def unique_id(self):
        if self.fail_uid:
            raise OSError("librccl.so.1: cannot open shared object file")
        return bytes((7 * i + 1) % 256 for i in range(256))

    def comm_init(self, uid, rank, world):
        self.inited = (uid, rank, world)


def _worker(rank, world, key, tmp, fail_uid, q):
    sys.path.insert(0, REPO)
    from phyloformer_amd import dist as pfdist
    from phyloformer_amd.rendezvous import TcpGroup
    try:
        with TcpGroup(rank, world, key=key, directory=tmp, timeout=30) as g:
            got = g.broadcast_bytes(bytes(range(128)) if rank == 0 else None, 128)
            ranks = g.allgather(rank)
            mx, mn = g.allreduce_max(float(rank) + 0.5), g.allreduce_min(float(rank) + 0.5)
            g.barrier()
            # fail_uid = "probe": the LAST rank cannot load librccl - found out before anyone blocks in comm_init
            eng = _FakeEngine(rank, fail_uid=fail_uid is True, no_rccl=fail_uid == "probe" and rank == world - 1)
            try:
                pfdist.init_engine_comm(eng, g)
                comm = ("ok", eng.inited)
            except RuntimeError as exc:
                comm = ("error", str(exc))
            flags = g.allgather(comm[0])
            q.put((rank, got, ranks, mx, mn, comm, flags))
    except Exception as exc:  # noqa: BLE001
        q.put((rank, "EXC", repr(exc)))


@pytest.mark.parametrize("world,fail_uid", [(2, False), (3, False), (2, True), (3, "probe")])
def test_tcp_group_collectives_and_comm_bootstrap(world, fail_uid, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = "pf_test_" + uuid.uuid4().hex
    procs = [ctx.Process(target=_worker, args=(r, world, key, str(tmp_path), fail_uid, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, got, ranks, mx, mn, comm, flags in res:
        assert got == bytes(range(128))
        assert ranks == list(range(world))
        assert mx == world - 0.5 and mn == 0.5
        if fail_uid:
            # rank 0 broadcast the sentinel instead of leaving its peers in a broadcast that never comes:
            # every rank raises the same error and they all agree
            assert comm[0] == "error" and ("unique id" in comm[1] if fail_uid is True else f"rank {world - 1}" in comm[1])
            assert flags == ["error"] * world
        else:
            uid, r, w = comm[1]
            assert comm[0] == "ok" and (r, w) == (rank, world)
            assert uid == bytes((7 * i + 1) % 256 for i in range(256))
    assert not [f for f in os.listdir(tmp_path) if f.startswith(key)], "rendezvous file left behind"


def test_tcp_group_single_rank_is_trivial():
    sys.path.insert(0, REPO)
    from phyloformer_amd.rendezvous import TcpGroup
    with TcpGroup(0, 1) as g:
        assert g.allgather("x") == ["x"] and g.allreduce_max(3.0) == 3.0
        assert g.broadcast_bytes(b"a" * 128, 128) == b"a" * 128
        g.barrier()


def _hub_worker(key, tmp, q):
    sys.path.insert(0, REPO)
    from phyloformer_amd.rendezvous import TcpGroup
    try:
        with TcpGroup(0, 2, key=key, directory=tmp, timeout=30) as g:
            q.put(("hub", g.allgather("zero")))
    except Exception as exc:  # noqa: BLE001
        q.put(("hub", "EXC " + repr(exc)))


def test_hub_survives_stray_and_hostile_clients(tmp_path):
    """rank 0 must not unpickle, must not block on a silent client and must not die on garbage.
    A silent client, a client with a wrong token and a client that sends a pickle are dropped; the real rank 1
    then joins and the collective completes."""
    import json
    import pickle
    import socket
    import stat
    import struct
    import time
    sys.path.insert(0, REPO)
    from phyloformer_amd.rendezvous import TcpGroup
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = "pf_test_" + uuid.uuid4().hex
    hub = ctx.Process(target=_hub_worker, args=(key, str(tmp_path), q))
    hub.start()
    path = os.path.join(str(tmp_path), key + ".json")
    for _ in range(400):
        if os.path.exists(path):
            break
        time.sleep(0.05)
    assert stat.S_IMODE(os.stat(path).st_mode) == 0o600, "the rendezvous file must not be world-readable"
    with open(path) as fh:
        info = json.load(fh)
    assert len(info["token"]) == 32
    silent = socket.create_connection(("127.0.0.1", info["port"]))          # says nothing
    wrong = socket.create_connection(("127.0.0.1", info["port"]))
    wrong.sendall(b"x" * 32 + struct.pack("<Q", 5) + b"hello")               # wrong token
    evil = socket.create_connection(("127.0.0.1", info["port"]))
    raw = pickle.dumps({"rank": 1, "world": 2})
    evil.sendall(info["token"].encode() + struct.pack("<Q", len(raw)) + raw)  # right token, pickle instead of JSON
    huge = socket.create_connection(("127.0.0.1", info["port"]))
    huge.sendall(info["token"].encode() + struct.pack("<Q", 1 << 40))        # absurd length
    t0 = time.monotonic()
    with TcpGroup(1, 2, key=key, directory=str(tmp_path), timeout=30) as g:
        assert g.allgather("one") == ["zero", "one"]
    assert time.monotonic() - t0 < 20, "a silent client must not hold the hub for its whole timeout"
    tag, got = q.get(timeout=30)
    assert got == ["zero", "one"], got
    hub.join(timeout=30)
    assert hub.exitcode == 0
    for s in (silent, wrong, evil, huge):
        s.close()


def test_messages_are_plain_values_only():
    sys.path.insert(0, REPO)
    from phyloformer_amd import rendezvous as rz
    assert rz._decode(rz._encode([1, 2.5, "a", None, True, b"\x00\xff", (3, b"x")])) == \
        [1, 2.5, "a", None, True, b"\x00\xff", [3, b"x"]]
    with pytest.raises(TypeError):
        rz._encode(object())
    assert "pickle" not in open(rz.__file__).read().split('"""', 2)[2], "no pickle on the wire"


def test_private_dir_is_0700_and_checked(tmp_path, monkeypatch):
    import stat
    import tempfile
    sys.path.insert(0, REPO)
    from phyloformer_amd import rendezvous as rz
    monkeypatch.setattr(tempfile, "tempdir", str(tmp_path))
    d = rz.private_dir()
    assert stat.S_IMODE(os.stat(d).st_mode) == 0o700
    os.chmod(d, 0o755)
    with pytest.raises(rz.RendezvousError):
        rz.private_dir()
