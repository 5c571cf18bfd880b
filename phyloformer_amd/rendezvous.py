"""Torch-free rendezvous of the ranks of ONE node (the bench contract: one process per GPU).

The only things the ranks of a site-sharded run exchange on the host are the 256-byte RCCL unique-id blob, a few
agreement flags and the barriers / max-reduction of the benchmark.  Importing ``torch.distributed`` for that
maps torch's bundled HIP runtime and librccl into a process whose hot path lives in
``libphyloformer_amd.so`` (linked against the system ROCm): two HIP runtimes in one address space.  This module
does the same job with the standard library only:

* rank 0 binds an ephemeral TCP port on 127.0.0.1 and publishes ``{port, token, world}`` in a file that only
  this user can read (mode 0600, inside a 0700 per-user directory), named after the launcher's environment
  (``MASTER_PORT`` plus ``PF_RUN_ID`` / ``TORCHELASTIC_RUN_ID``), written atomically (``os.replace``);
* the other ranks poll for the file, connect and open with the 32-byte token - rank 0 drops, within two
  seconds, any connection that does not;
* every collective is a star through rank 0 (world <= 8: a few hundred bytes, microseconds).

Wire format: an 8-byte little-endian length, then UTF-8 JSON.  Nothing that arrives from a socket is ever
unpickled or evaluated: the values the ranks exchange are ``None``, booleans, numbers, strings, ``bytes``
(sent as ``{"__bytes__": hex}``) and lists / tuples of those (tuples arrive as lists).

``TcpGroup`` offers ``broadcast_bytes``, ``barrier``, ``allreduce_max / allreduce_min`` and ``gather`` - what
``bench.py`` and :func:`phyloformer_amd.dist.init_engine_comm` need.
"""
from __future__ import annotations

import json
import os
import secrets
import socket
import stat
import struct
import tempfile
import time
from typing import Any, List, Optional

MAX_MESSAGE = 1 << 20        # bytes; the largest legitimate message is a list of eight short strings
TOKEN_BYTES = 32             # hex characters of the shared secret that opens a connection
HELLO_TIMEOUT = 2.0          # seconds rank 0 waits for a new connection's token + hello


class RendezvousError(RuntimeError):
    pass


def _encode(obj: Any) -> Any:
    if isinstance(obj, (bytes, bytearray)):
        return {"__bytes__": bytes(obj).hex()}
    if isinstance(obj, (list, tuple)):
        return [_encode(v) for v in obj]
    if isinstance(obj, dict):
        return {str(k): _encode(v) for k, v in obj.items()}
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    raise TypeError(f"rendezvous messages carry plain values only, got {type(obj).__name__}")


def _decode(obj: Any) -> Any:
    if isinstance(obj, dict):
        if set(obj) == {"__bytes__"}:
            return bytes.fromhex(obj["__bytes__"])
        return {k: _decode(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_decode(v) for v in obj]
    return obj


def _exact(sock: socket.socket, n: int) -> bytes:
    chunks, got = [], 0
    while got < n:
        c = sock.recv(n - got)
        if not c:
            raise RendezvousError("peer closed the rendezvous connection")
        chunks.append(c)
        got += len(c)
    return b"".join(chunks)


def _send(sock: socket.socket, obj: Any) -> None:
    raw = json.dumps(_encode(obj), separators=(",", ":")).encode()
    sock.sendall(struct.pack("<Q", len(raw)) + raw)


def _recv(sock: socket.socket) -> Any:
    (n,) = struct.unpack("<Q", _exact(sock, 8))
    if n > MAX_MESSAGE:
        raise RendezvousError(f"rendezvous message of {n} bytes refused (limit {MAX_MESSAGE})")
    try:
        return _decode(json.loads(_exact(sock, n).decode()))
    except (UnicodeDecodeError, ValueError) as exc:
        raise RendezvousError(f"malformed rendezvous message: {exc}") from None


def default_key() -> str:
    """A name the ranks of one launch share and other launches on the box do not."""
    run = os.environ.get("PF_RUN_ID") or os.environ.get("TORCHELASTIC_RUN_ID", "none")
    return "pf_rdzv_{}_{}".format(os.environ.get("MASTER_PORT", "0"), run)


def private_dir() -> str:
    """``$TMPDIR/pf_rdzv_<uid>``, created 0700 and refused if anyone else owns or can enter it."""
    d = os.path.join(tempfile.gettempdir(), f"pf_rdzv_{os.getuid()}")
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RendezvousError(f"{d} is not a private directory of this user (owner {st.st_uid}, "
                              f"mode {stat.S_IMODE(st.st_mode):o}); remove it or pass directory=")
    return d


class TcpGroup:
    """Process group of the ranks of one node; rank 0 is the hub.

    ``timeout`` bounds the rendezvous itself; ``op_timeout`` (default 30 min, ``None`` = none) bounds a
    single collective, i.e. the skew between the fastest and the slowest rank at a barrier."""

    def __init__(self, rank: int, world: int, key: Optional[str] = None, timeout: float = 120.0,
                 directory: Optional[str] = None, op_timeout: Optional[float] = 1800.0):
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world of {world}")
        self.rank, self.world = rank, world
        self._peers: List[socket.socket] = []      # rank 0: sockets of ranks 1..world-1 (index rank-1)
        self._hub: Optional[socket.socket] = None  # other ranks: socket to rank 0
        self._server: Optional[socket.socket] = None
        self._path = ""
        if world == 1:
            return
        self._path = os.path.join(directory or private_dir(), (key or default_key()) + ".json")
        deadline = time.monotonic() + timeout
        if rank == 0:
            self._serve(world, timeout, deadline, op_timeout)
        else:
            self._join(rank, world, timeout, deadline, op_timeout)

    # -- rank 0 ---------------------------------------------------------------------------------------
    def _serve(self, world, timeout, deadline, op_timeout):
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind(("127.0.0.1", 0))
        srv.listen(world + 8)
        self._server = srv
        token = secrets.token_hex(TOKEN_BYTES // 2)
        tmp = self._path + f".{os.getpid()}.tmp"
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
        with os.fdopen(fd, "w") as fh:
            json.dump({"port": srv.getsockname()[1], "pid": os.getpid(), "world": world, "token": token}, fh)
        os.replace(tmp, self._path)
        slots: List[Optional[socket.socket]] = [None] * (world - 1)
        while any(s is None for s in slots):
            left = deadline - time.monotonic()
            if left <= 0:
                n = sum(s is not None for s in slots)
                self.close()
                raise RendezvousError(f"rank 0: only {n} of {world - 1} peers connected within {timeout:.0f} s")
            srv.settimeout(min(left, 1.0))
            try:
                conn, _ = srv.accept()
            except socket.timeout:
                continue
            keep = False
            try:
                # the token comes first, raw and fixed-length: nothing is parsed for a client without it,
                # and a stray client that sends nothing is dropped after HELLO_TIMEOUT
                conn.settimeout(HELLO_TIMEOUT)
                if secrets.compare_digest(_exact(conn, TOKEN_BYTES), token.encode()):
                    hello = _recv(conn)
                    r = int(hello["rank"])
                    if hello.get("world") == world and 1 <= r < world and slots[r - 1] is None:
                        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        conn.settimeout(op_timeout)
                        slots[r - 1] = conn
                        keep = True
            except Exception:  # noqa: BLE001 - a stale, foreign or broken client never takes rank 0 down
                pass
            if not keep:
                try:
                    conn.close()
                except OSError:
                    pass
        self._peers = [s for s in slots if s is not None]
        for s in self._peers:
            _send(s, {"ok": True})

    # -- ranks 1 .. world-1 ---------------------------------------------------------------------------
    def _join(self, rank, world, timeout, deadline, op_timeout):
        while True:
            s = None
            try:
                with open(self._path) as fh:
                    info = json.load(fh)
                if info.get("world") == world and isinstance(info.get("token"), str):
                    s = socket.create_connection(("127.0.0.1", int(info["port"])), timeout=2.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    # the acknowledgement comes once ALL peers have connected: wait as long as the rendezvous may take
                    s.settimeout(max(1.0, deadline - time.monotonic()))
                    s.sendall(info["token"].encode()[:TOKEN_BYTES].ljust(TOKEN_BYTES, b"0"))
                    _send(s, {"rank": rank, "world": world})
                    ack = _recv(s)
                    if isinstance(ack, dict) and ack.get("ok") is True:
                        s.settimeout(op_timeout)
                        self._hub = s
                        return
            except Exception:  # noqa: BLE001 - file not there yet, stale file of an earlier launch, a foreign
                pass           # service on a stale port, hub not listening yet: close and try again
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
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
            return [_decode(_encode(value))] + [_recv(s) for s in self._peers]
        _send(self._hub, value)
        return None

    def broadcast(self, value: Any = None) -> Any:
        """Everybody receives rank 0's ``value``."""
        if self.world == 1:
            return value
        if self.rank == 0:
            for s in self._peers:
                _send(s, value)
            return _decode(_encode(value))
        return _recv(self._hub)

    def allgather(self, value: Any) -> List[Any]:
        return self.broadcast(self.gather(value))

    def broadcast_bytes(self, payload: Optional[bytes], nbytes: int) -> bytes:
        out = self.broadcast(bytes(payload) if self.rank == 0 else None)
        if not isinstance(out, bytes) or len(out) != nbytes:
            raise RendezvousError(f"broadcast of {len(out) if isinstance(out, bytes) else type(out).__name__} "
                                  f"bytes, expected {nbytes}")
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
