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

``torch.distributed`` (gloo or nccl backend) is used only for the rendezvous:
broadcasting the RCCL unique id and the barriers of the benchmark.
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


def broadcast_bytes(payload: Optional[bytes], nbytes: int, src: int = 0) -> bytes:
    """Broadcast a fixed-size byte string over the default torch.distributed group."""
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


def init_engine_comm(engine) -> Tuple[int, int]:
    """Create the engine's RCCL communicator across the torch.distributed world."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        engine.comm_init(None, 0, 1)
        return 0, 1
    rank, world = dist.get_rank(), dist.get_world_size()
    uid = engine.unique_id() if rank == 0 else None
    uid = broadcast_bytes(uid, 128, src=0)
    engine.comm_init(uid, rank, world)
    return rank, world


def shard_sites(idx: np.ndarray, world: int, rank: int) -> Tuple[np.ndarray, int, int]:
    """Slice ``uint8[..., N, L]`` to this rank's sites → ``(local idx, l_begin, l_end)``."""
    L = idx.shape[-1]
    lo, hi = site_range(L, world, rank)
    return np.ascontiguousarray(idx[..., lo:hi]), lo, hi
