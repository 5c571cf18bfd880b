"""Multi-GPU plumbing: one process per GPU, RCCL over xGMI inside the native library.

Two ways to use several GPUs:

* **site sharding** (one alignment spread over ranks): rank ``r`` holds the
  sites ``site_range(L, world, r)`` of every pair.  Column attention, the FFN,
  LayerNorm and the head are site-local; row attention reduces over sites
  (/root/reference/phyloformer/attention.py:183-190 with ``dim=-2`` = sites,
  model.py:91), so the per-pair statistics ``[B][P][72]`` are all-reduced once
  per block and the ``[B][P]`` site sums once at the end — ``n_blocks + 1``
  collectives per forward, issued with ``ncclAllReduce`` on the engine's stream.
* **alignment sharding** (independent alignments per rank): no collective.

The host-side rendezvous (the RCCL unique ids - 256 bytes, one id per communicator / stream -, agreement flags, the
barriers of the benchmark) goes through :class:`phyloformer_amd.rendezvous.TcpGroup`
(standard library only), so a rank never maps torch's bundled HIP runtime next to
the one ``libphyloformer_amd.so`` is linked against.  A ``torch.distributed`` group
is still accepted (gloo tests).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import numpy as np


def site_range(L: int, world: int, rank: int) -> Tuple[int, int]:
    """Ceil-split of ``L`` sites; trailing ranks may be short or empty."""
    step = -(-L // world)
    return min(rank * step, L), min((rank + 1) * step, L)


def site_ranges(L: int, world: int) -> List[Tuple[int, int]]:
    return [site_range(L, world, r) for r in range(world)]


def alignment_range(B: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of a batch of ``B`` alignments for ``rank`` (sizes differ by at most 1)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_rank_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def broadcast_bytes(payload: Optional[bytes], nbytes: int, src: int = 0, group=None) -> bytes:
    """Broadcast a fixed-size byte string from rank ``src`` (= 0 for a TcpGroup).

    ``group``: a :class:`~phyloformer_amd.rendezvous.TcpGroup`, or ``None`` for the default
    ``torch.distributed`` group."""
    if group is not None:
        if src != 0:
            raise ValueError("TcpGroup broadcasts from rank 0")
        return group.broadcast_bytes(payload, nbytes)
    import torch
    import torch.distributed as dist
    buf = torch.zeros(nbytes, dtype=torch.uint8)
    if dist.get_rank() == src:
        buf[:] = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
    dev = None
    if dist.get_backend() == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        buf = buf.to(dev)
    dist.broadcast(buf, src=src)
    return bytes(buf.cpu().numpy().tobytes())


UNIQUE_ID_BYTES = 256    # include/phyloformer_amd.h: PF_UNIQUE_ID_BYTES (two ncclUniqueIds)
_NO_ID = bytes(UNIQUE_ID_BYTES)      # sentinel: rank 0 could not produce a unique id


def init_engine_comm(engine, group=None) -> Tuple[int, int]:
    """Create the engine's RCCL communicator across ``group`` (TcpGroup) or the torch.distributed world.

    Every rank takes part in the same host collectives whatever happens: if rank 0 cannot produce a
    unique id (librccl missing or bound to another HIP runtime) it broadcasts an all-zero sentinel and
    every rank raises the same ``RuntimeError`` - no rank is left waiting in a broadcast that never comes."""
    if group is not None:
        rank, world = group.rank, group.world
    else:
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size() == 1:
            engine.comm_init(None, 0, 1)
            return 0, 1
        rank, world = dist.get_rank(), dist.get_world_size()
    if world == 1:
        engine.comm_init(None, 0, 1)
        return 0, 1
    if group is not None and hasattr(engine, "comm_info"):
        # ncclCommInitRank blocks until every rank has entered it: make sure beforehand that every rank can
        # load a usable librccl, so that a rank that cannot is reported by all instead of hanging the others
        try:
            engine.comm_info()
            mine = ""
        except Exception as exc:  # noqa: BLE001 - reported to every rank below
            mine = f"rank {rank}: {type(exc).__name__}: {exc}"
        bad = [m for m in group.allgather(mine) if m]
        if bad:
            raise RuntimeError("RCCL cannot be loaded on every rank (" + "; ".join(bad) + ")")
    uid, why = None, ""
    if rank == 0:
        try:
            uid = engine.unique_id()
        except Exception as exc:  # noqa: BLE001 - reported to every rank below
            uid, why = _NO_ID, f"{type(exc).__name__}: {exc}"
    uid = broadcast_bytes(uid, UNIQUE_ID_BYTES, src=0, group=group)
    if uid == _NO_ID:
        raise RuntimeError("rank 0 could not create an RCCL unique id" + (f" ({why})" if why else ""))
    engine.comm_init(uid, rank, world)
    return rank, world


def shard_sites(idx: np.ndarray, world: int, rank: int) -> Tuple[np.ndarray, int, int]:
    """Slice ``uint8[..., N, L]`` to this rank's sites → ``(local idx, l_begin, l_end)``."""
    L = idx.shape[-1]
    lo, hi = site_range(L, world, rank)
    return np.ascontiguousarray(idx[..., lo:hi]), lo, hi
