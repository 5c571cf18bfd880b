"""Torch-free rendezvous of the ranks of ONE node (the bench contract: one process per GPU).

The only things the ranks of a site-sharded run exchange on the host are the 128-byte RCCL unique id, a few
agreement flags and the barriers / max-reduction of the benchmark.  Importing ``torch.distributed`` for that
maps torch's bundled HIP runtime and librccl into a process whose hot path lives in
``libphyloformer_amd.so`` (linked against the system ROCm): two HIP runtimes in one address space.  This module
does the same job with the standard library only:

* rank 0 binds an ephemeral TCP port on 127.0.0.1 and publishes it in a small file whose name is derived
  from the launcher's environment (``MASTER_PORT`` and ``TORCHELASTIC_RUN_ID``, both exported by
  ``python -m torch.distributed.run``), written atomically (``os.replace``);
* the other ranks poll for the file, connect and identify themselves;
* every collective is a star through rank 0 (world <= 8: a few hundred bytes, microseconds).

``TcpGroup`` offers ``broadcast_bytes``, ``barrier``, ``allreduce_max / allreduce_min`` and ``gather`` - what
``bench.py`` and :func:`phyloformer_amd.dist.init_engine_comm` need.
"""
from __future__ import annotations

import json
import os
import pickle
import socket
import struct
import tempfile
import time
from typing import Any, List, Optional


class RendezvousError(RuntimeError):
    pass


def _send(sock: socket.socket, obj: Any) -> None:
    raw = pickle.dumps(obj, protocol=4)
    sock.sendall(struct.pack("<Q", len(raw)) + raw)


def _recv(sock: socket.socket) -> Any:
    def exact(n: int) -> bytes:
        chunks, got = [], 0
        while got < n:
            c = sock.recv(n - got)
            if not c:
                raise RendezvousError("peer closed the rendezvous connection")
            chunks.append(c)
            got += len(c)
        return b"".join(chunks)
    (n,) = struct.unpack("<Q", exact(8))
    return pickle.loads(exact(n))


def default_key() -> str:
    """A name the ranks of one launch share and other launches on the box do not."""
    return "pf_rdzv_{}_{}_{}".format(os.environ.get("MASTER_PORT", "0"),
                                     os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getuid())


class TcpGroup:
    """Process group of the ranks of one node; rank 0 is the hub."""

    def __init__(self, rank: int, world: int, key: Optional[str] = None, timeout: float = 120.0,
                 directory: Optional[str] = None):
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world of {world}")
        self.rank, self.world = rank, world
        self._peers: List[socket.socket] = []      # rank 0: sockets of ranks 1..world-1 (index rank-1)
        self._hub: Optional[socket.socket] = None  # other ranks: socket to rank 0
        self._path = os.path.join(directory or tempfile.gettempdir(), (key or default_key()) + ".json")
        self._server: Optional[socket.socket] = None
        if world == 1:
            return
        deadline = time.monotonic() + timeout
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(world)
            self._server = srv
            tmp = self._path + f".{os.getpid()}.tmp"
            with open(tmp, "w") as fh:
                json.dump({"port": srv.getsockname()[1], "pid": os.getpid(), "world": world}, fh)
            os.replace(tmp, self._path)
            slots: List[Optional[socket.socket]] = [None] * (world - 1)
            while any(s is None for s in slots):
                srv.settimeout(max(0.1, deadline - time.monotonic()))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    self.close()
                    raise RendezvousError(f"rank 0: only {sum(s is not None for s in slots)} of {world - 1} "
                                          f"peers connected within {timeout:.0f} s")
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                hello = _recv(conn)
                r = int(hello["rank"])
                if hello.get("world") != world or not 1 <= r < world or slots[r - 1] is not None:
                    conn.close()
                    continue            # a stale or foreign client: ignore it
                slots[r - 1] = conn
            self._peers = [s for s in slots if s is not None]
            for s in self._peers:
                _send(s, {"ok": True})
        else:
            while True:
                try:
                    with open(self._path) as fh:
                        info = json.load(fh)
                    if info.get("world") == world:
                        s = socket.create_connection(("127.0.0.1", int(info["port"])), timeout=2.0)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        s.settimeout(timeout)
                        _send(s, {"rank": rank, "world": world})
                        if _recv(s).get("ok"):
                            self._hub = s
                            break
                except (OSError, ValueError, RendezvousError):
                    pass                # file not there yet, stale file of an earlier launch, hub not listening yet
                if time.monotonic() > deadline:
                    raise RendezvousError(f"rank {rank}: no rendezvous with rank 0 within {timeout:.0f} s "
                                          f"(looked for {self._path})")
                time.sleep(0.05)

    # -- collectives (star through rank 0) ----------------------------------------------------------
    def gather(self, value: Any) -> Optional[List[Any]]:
        """Rank 0 receives ``[value of rank 0, ..., value of rank world-1]``; the others ``None``."""
        if self.world == 1:
            return [value]
        if self.rank == 0:
            return [value] + [_recv(s) for s in self._peers]
        _send(self._hub, value)
        return None

    def broadcast(self, value: Any = None) -> Any:
        """Everybody receives rank 0's ``value``."""
        if self.world == 1:
            return value
        if self.rank == 0:
            for s in self._peers:
                _send(s, value)
            return value
        return _recv(self._hub)

    def allgather(self, value: Any) -> List[Any]:
        return self.broadcast(self.gather(value))

    def broadcast_bytes(self, payload: Optional[bytes], nbytes: int) -> bytes:
        out = self.broadcast(bytes(payload) if self.rank == 0 else None)
        if len(out) != nbytes:
            raise RendezvousError(f"broadcast of {len(out)} bytes, expected {nbytes}")
        return out

    def allreduce_max(self, x: float) -> float:
        return max(self.allgather(x))

    def allreduce_min(self, x: float) -> float:
        return min(self.allgather(x))

    def barrier(self) -> None:
        self.allgather(None)

    def close(self) -> None:
        for s in self._peers:
            try:
                s.close()
            except OSError:
                pass
        self._peers = []
        if self._hub is not None:
            try:
                self._hub.close()
            except OSError:
                pass
            self._hub = None
        if self._server is not None:
            self._server.close()
            self._server = None
            try:
                os.unlink(self._path)
            except OSError:
                pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
