"""Batched, shape-bucketed, I/O-overlapped inference over a directory of alignments.

The reference loop (/root/reference/infer_alns.py:95-123) is strictly serial:
parse one FASTA, run one forward, format and write one PHYLIP file.  On an
MI355X the forward of a 60 x 500 alignment takes 2.3 ms — less than CPython
needs for the parsing and formatting around it — so the CLI would be host-bound
by a wide margin.  This module keeps the GPU fed:

* loader threads parse FASTA files ahead of the GPU (native parser, GIL released);
* alignments are *bucketed by shape* ``(N, L)``; a bucket is launched when it
  holds ``batch`` alignments (or a token budget's worth), whatever order the
  files arrived in, and all partial buckets are flushed at the end;
* writer threads format and write the PHYLIP (and NJ) files of batch ``k`` while
  the GPU runs batch ``k + 1``;
* two engines on two host threads (``--gpu-streams``) keep the GPU busy across the
  host-side gaps of the synchronous ``pf_forward``;
* ``run_multi_device`` shards the *files* over several GPUs, one process per
  GPU, no collective (alignment-level data parallelism, SURVEY.md §8e way 1);
* ``--shard sites`` (``SiteShardedRunner``) spreads every alignment over the GPUs instead
  (SURVEY.md §8e way 2): each rank parses the file, keeps its block of sites and calls
  ``pf_forward_sharded`` - n_blocks + 1 RCCL all-reduces per launch - and rank 0 writes the
  outputs.  The ranks meet through ``rendezvous.TcpGroup`` and walk the files in ONE
  deterministic order (collectives must be issued in the same order everywhere).

Every alignment is computed independently inside a launch and no launch parameter
that affects the order of a sum depends on the batch size: an alignment's distances
are bit-identical whatever batch it travels in (tests/test_gpu_parity.py,
tests/test_cli_gpu.py).
"""
from __future__ import annotations

import json
import os
import queue
import subprocess
import sys
import threading
import time
from collections import OrderedDict, deque
from concurrent.futures import Future, ThreadPoolExecutor
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# one launch should carry about this many (pair, site) tokens: 16 alignments of 60 x 500
# (3.6 GB of residual stream), the size bench.py measures the headline number at
TOKEN_BUDGET = 16 * 1770 * 500


# files per native load call (FastaBatch): enough to amortise a thread-pool start over many small files, few enough
# that the first launch does not wait for the whole directory
FILES_PER_LOAD = 256
# a partial shape bucket is launched after this many further load calls (bounds what the native pipeline keeps parsed)
STALE_LOADS = 16


def auto_batch(n_seqs: int, n_sites: int, max_batch: int = 4096, token_budget: int = TOKEN_BUDGET) -> int:
    """Alignments per launch for shape ``(N, L)``: fill the token budget, at least 1."""
    tokens = max(1, n_seqs * (n_seqs - 1) // 2 * n_sites)
    return int(min(max_batch, max(1, token_budget // tokens)))


HEARTBEAT = "#pf-heartbeat"      # stderr line prefix of a site-sharded worker's sign of life (swallowed by the launcher)
MAX_SEQS = 200     # SEQ2PAIR = seq2pair(200), /root/reference/phyloformer/model.py:39


def single_sequence_error(batch: int = 1) -> RuntimeError:
    """An alignment of ONE sequence parses (data.py:11-31) but has no pair: the reference's forward fails on the empty
    tensor in attention.py:193 with this RuntimeError (tests/golden/cli_bad_entry.json, from the real CLI)."""
    return RuntimeError(f"cannot reshape tensor of 0 elements into shape [{batch}, -1, 0, 64] because the unspecified "
                        "dimension size -1 can be any value and is ambiguous")


def too_many_seqs(n_seqs: int) -> Optional[Exception]:
    """The reference's forward refuses more than 200 sequences (ValueError, adaptable_seq2pair, model.py:24-28) and fails
    on a single one (RuntimeError, attention.py:193) when the loop reaches that file; the runner raises the same error at
    the same place in the order - after the files in front of it."""
    if n_seqs > MAX_SEQS:
        return ValueError(f"n_seqs must be smaller or equal to {MAX_SEQS} (or pre-compute a larger global_seq2pair)")
    if n_seqs == 1:
        return single_sequence_error()
    return None


def has_fasta_ext(alnpath: str) -> bool:
    """Checks if a path ends in .fa or .fasta (infer_alns.py:36-38)."""
    return alnpath.lower().endswith(".fa") or alnpath.lower().endswith(".fasta")


def slice_paths(paths: Sequence[str], rank: int, world: int) -> List[str]:
    """Deterministic share of ``paths`` for worker ``rank`` of ``world``: files are sorted by
    (size, name) and dealt round-robin, so every worker sees the same mix of shapes."""
    if world <= 1:
        return list(paths)

    def key(p):
        try:
            return (os.path.getsize(p), p)
        except OSError:
            return (0, p)
    return sorted(paths, key=key)[rank::world]


class DirectoryRunner:
    """Runs every alignment of a file list through the engine(s) and writes the outputs.

    ``engine`` may be one engine or a list: each engine gets its own host thread (one HIP stream each),
    so the host-side gaps of one synchronous ``pf_forward`` (index copy, result copy, Python between
    calls) are filled by the other's kernels — measured 486 -> 501 alignments/s at 60 x 500 with two."""

    def __init__(self, engine, out_dir: str, trees: bool = False, batch: int = 0,
                 io_threads: int = 4, native_io: bool = True, progress=None):
        self.engines = list(engine) if isinstance(engine, (list, tuple)) else [engine]
        self.out_dir = out_dir
        self.trees = trees
        self.batch = batch            # 0 = auto per shape
        self.io_threads = max(1, io_threads)
        self.native_io = native_io
        self.progress = progress
        self.stats = {"alignments": 0, "launches": 0, "forward_s": 0.0, "load_wait_s": 0.0,
                      "write_wait_s": 0.0, "shapes": {}, "gpu_streams": len(self.engines)}
        self._lock = threading.Lock()

    # -- stages -----------------------------------------------------------------------------
    def _load(self, path: str):
        if self.native_io:
            from .hostio import load_alignment
        else:
            from .fasta import load_alignment
        return load_alignment(path)

    def _write(self, path: str, pred: np.ndarray, ids: List[str]):
        stem = Path(path).stem
        if self.native_io:
            from .hostio import format_phylip, nj_newick
            with open(os.path.join(self.out_dir, f"{stem}.phy"), "wb") as fh:
                fh.write(format_phylip(pred, ids))
            if self.trees:                         # (infer_alns.py:120-123)
                with open(os.path.join(self.out_dir, f"{stem}.nj.nwk"), "wb") as fh:
                    fh.write(nj_newick(pred, ids))
            return
        from .phylip import vec_to_phylip
        dm, text = vec_to_phylip(pred, ids)
        with open(os.path.join(self.out_dir, f"{stem}.phy"), "w") as fh:
            fh.write(text)
        if self.trees:
            from .nj import neighbor_joining
            with open(os.path.join(self.out_dir, f"{stem}.nj.nwk"), "w") as fh:
                fh.write(neighbor_joining(dm.astype("float64"), ids))

    def _launch(self, engine, shape: Tuple[int, int], group: list, writers: ThreadPoolExecutor, pending: deque):
        native = group[0][2] is None              # entries of _feed_native: (path, (FastaBatch, file), None)
        t0 = time.perf_counter()
        if native:
            from .hostio import gather
            batch = gather([g[1] for g in group], shape[0], shape[1])
        else:
            batch = np.stack([g[1] for g in group])
        preds = engine.forward(batch)
        dt = time.perf_counter() - t0
        with self._lock:
            self.stats["forward_s"] += dt
            self.stats["launches"] += 1
            self.stats["alignments"] += len(group)
            key = f"{shape[0]}x{shape[1]}"
            self.stats["shapes"][key] = self.stats["shapes"].get(key, 0) + len(group)
            if native:
                pending.append(writers.submit(self._write_native, shape[0], group, preds))
            else:
                for (path, _idx, ids), pred in zip(group, preds):
                    pending.append(writers.submit(self._write, path, pred, ids))
            if self.progress is not None:
                self.progress(len(group))
            drain = []
            # bound the write queue so results do not pile up in memory
            while len(pending) > 8 * self.io_threads + (1 if native else len(group)):
                drain.append(pending.popleft())
        t0 = time.perf_counter()
        for f in drain:
            f.result()
        with self._lock:
            self.stats["write_wait_s"] += time.perf_counter() - t0

    def _write_native(self, n: int, group: list, preds: np.ndarray):
        """``<stem>.phy`` - and with ``--trees`` ``<stem>.nj.nwk`` - of a whole launch: formatted (the trees: joined) and
        written on native threads (infer_alns.py:105-123)."""
        from .hostio import write_phylip
        outs = [os.path.join(self.out_dir, f"{Path(g[0]).stem}.phy") for g in group]
        trees = [os.path.join(self.out_dir, f"{Path(g[0]).stem}.nj.nwk") for g in group] if self.trees else None
        # (at most 4 threads: creating files in ONE directory from 8 / 16 threads is a lock convoy on the directory -
        # 4,096 outputs took 0.98 / 1.19 s instead of 0.01 s, profiles/r05c_cli_bench.txt.  The neighbour joining of
        # --trees rides on the same threads: 64 us per 20-taxon tree, 18 ms at 200 taxa, against 90 us / 22 ms of GPU time
        # per alignment.  PF_WRITER_THREADS moves the cap for experiments, profiles/r06r_cli_bench_overlay_writer_threads.txt.)
        cap = int(os.environ.get("PF_WRITER_THREADS", "4"))
        write_phylip([g[1] for g in group], n, preds, outs, max(1, min(self.io_threads, cap)), trees)

    def _gpu_worker(self, engine, jobs: "queue.Queue", writers, pending, errors: list):
        while True:
            job = jobs.get()
            if job is None:
                return
            if errors:
                continue                      # drain the queue after a failure
            try:
                self._launch(engine, job[0], job[1], writers, pending)
            except BaseException as exc:      # noqa: BLE001 - re-raised in run()
                errors.append(exc)

    # -- driver -----------------------------------------------------------------------------
    def run(self, paths: Sequence[str]) -> dict:
        """Process ``paths`` in the given order.  Side effects on a bad entry are the reference's
        (infer_alns.py:97-117 handles one file after the other): every alignment in front of the first entry
        without a FASTA extension (``ValueError``, :100-103), or of the first file that does not parse
        (``KeyError`` / ``ValueError`` / ... from ``load_alignment``, :108), is computed and written; nothing
        behind it is; then the exception is raised."""
        paths = list(paths)
        deferred: Optional[BaseException] = None
        for k, p in enumerate(paths):
            if not has_fasta_ext(p):
                deferred = ValueError("Input files must be fasta files (.fa or .fasta). Got " f"{p}")
                paths = paths[:k]
                break
        t_start = time.perf_counter()
        pending: deque = deque()
        errors: list = []
        jobs: "queue.Queue" = queue.Queue(maxsize=2 * len(self.engines))
        with ThreadPoolExecutor(self.io_threads, thread_name_prefix="pf-load") as loaders, \
                ThreadPoolExecutor(self.io_threads, thread_name_prefix="pf-write") as writers:
            workers = [threading.Thread(target=self._gpu_worker, args=(e, jobs, writers, pending, errors),
                                        name=f"pf-gpu{k}", daemon=True) for k, e in enumerate(self.engines)]
            for w in workers:
                w.start()
            try:
                feed = self._feed_native if self.native_io else self._feed_python
                bad = feed(paths, loaders, jobs, errors)
                deferred = bad or deferred         # a file that fails to parse sits in front of a bad extension
            finally:
                for _ in workers:
                    jobs.put(None)
                for w in workers:
                    w.join()
            if errors:
                raise errors[0]
            t0 = time.perf_counter()
            while pending:
                pending.popleft().result()
            self.stats["write_wait_s"] += time.perf_counter() - t0
        self.stats["wall_s"] = time.perf_counter() - t_start
        if deferred is not None:
            raise deferred
        return self.stats

    def _feed_python(self, paths, loaders, jobs, errors) -> Optional[BaseException]:
        """One future per file (``--python-io``, ``-t``): parse ahead, bucket by shape, launch full buckets."""
        buckets: "OrderedDict[Tuple[int, int], list]" = OrderedDict()
        lookahead = max(64, 4 * (self.batch or 64))
        inflight: "deque[Tuple[str, Future]]" = deque()
        it = iter(paths)
        exhausted = False
        bad: Optional[BaseException] = None
        while not errors:
            while not exhausted and len(inflight) < lookahead:
                try:
                    p = next(it)
                except StopIteration:
                    exhausted = True
                    break
                inflight.append((p, loaders.submit(self._load, p)))
            if not inflight:
                break
            path, fut = inflight.popleft()
            t0 = time.perf_counter()
            try:
                idx, ids = fut.result()          # parser exceptions surface here, as in the reference
            except Exception as exc:             # noqa: BLE001 - raised by run() once the files in front are written
                bad = exc
                for _p, f in inflight:
                    f.cancel()
                break
            finally:
                self.stats["load_wait_s"] += time.perf_counter() - t0
            shape = (int(idx.shape[0]), int(idx.shape[1]))
            bad = too_many_seqs(shape[0])
            if bad is not None:
                for _p, f in inflight:
                    f.cancel()
                break
            group = buckets.setdefault(shape, [])
            group.append((path, idx, ids))
            if len(group) >= (self.batch or auto_batch(*shape)):
                jobs.put((shape, group))
                buckets[shape] = []
        for shape, group in sorted(buckets.items(), key=lambda kv: -len(kv[1])):
            if group and not errors:
                jobs.put((shape, group))
        return bad

    def _feed_native(self, paths, loaders, jobs, errors) -> Optional[BaseException]:
        """The fast path: ``FILES_PER_LOAD`` files per native call (read + parsed on ``io_threads`` std::threads,
        no GIL), two calls in flight ahead of the bucketing; a bucket entry is ``(path, (batch, file), None)``
        - residues and ids stay in the library until the launch gathers them / the writer formats them."""
        from .hostio import FastaBatch
        buckets: "OrderedDict[Tuple[int, int], list]" = OrderedDict()
        born: Dict[Tuple[int, int], int] = {}          # load call that opened each partial bucket
        chunks = [paths[k:k + FILES_PER_LOAD] for k in range(0, len(paths), FILES_PER_LOAD)]
        inflight: "deque[Future]" = deque()
        nxt = 0
        done = 0
        bad: Optional[BaseException] = None
        while not errors and bad is None:
            while nxt < len(chunks) and len(inflight) < 2:
                inflight.append(loaders.submit(FastaBatch, chunks[nxt], self.io_threads))
                nxt += 1
            if not inflight:
                break
            t0 = time.perf_counter()
            fb = inflight.popleft().result()
            self.stats["load_wait_s"] += time.perf_counter() - t0
            ok = (fb.status == 0) & (fb.l > 0) & (fb.n <= MAX_SEQS) & (fb.n != 1)
            stop = len(fb) if ok.all() else int(np.argmin(ok))
            if stop < len(fb):
                bad = fb.error(stop) or too_many_seqs(int(fb.n[stop]))
            ns, ls = fb.n.tolist(), fb.l.tolist()
            for i in range(stop):
                shape = (ns[i], ls[i])
                group = buckets.get(shape)
                if not group:
                    group = buckets[shape] = []
                    born[shape] = done
                group.append((fb.paths[i], (fb, i), None))
                if len(group) >= (self.batch or auto_batch(*shape)):
                    jobs.put((shape, group))
                    buckets[shape] = []
            done += 1
            # A partial bucket keeps its entries' batch objects - every parsed file of those load calls - alive: a rare
            # shape in a long directory would pin the whole directory in memory.  A bucket that has waited for
            # STALE_LOADS load calls is launched as it is (an alignment's bits do not depend on its batch).
            for shape in [sh for sh, g in buckets.items() if g and done - born[sh] >= STALE_LOADS]:
                jobs.put((shape, buckets[shape]))
                buckets[shape] = []
        for f in inflight:
            f.cancel()
        for shape, group in sorted(buckets.items(), key=lambda kv: -len(kv[1])):
            if group and not errors:
                jobs.put((shape, group))
        return bad


class SiteShardedRunner(DirectoryRunner):
    """``--shard sites``: every alignment is spread over the ``world`` ranks of one node.

    Rank ``r`` holds the sites ``dist.site_range(L, world, r)`` of every pair; the row-attention statistics
    (/root/reference/phyloformer/attention.py:183-190 reduced over sites, model.py:91) and the final site sums
    (model.py:185) are all-reduced inside ``pf_forward_sharded``, so every rank receives the full distance
    vector and rank 0 alone writes ``<stem>.phy`` / ``<stem>.nj.nwk`` (output contract: infer_alns.py:105-123).

    Collectives must be issued in the same order on every rank: all ranks sort the paths the same way, parse
    every file themselves (the parse is cheap beside a sharded forward of a long alignment), bucket by shape in
    that order on ONE host thread and flush the partial buckets in sorted shape order.  Before each launch the
    ranks compare (shape, batch, checksum of the file names) through the rendezvous: a directory that differs
    between ranks is an error on all of them, not a hang."""

    def __init__(self, engine, group, rank: int, world: int, out_dir: str, heartbeat_s: Optional[float] = None, **kw):
        super().__init__([engine], out_dir, **kw)
        self.group, self.rank, self.world = group, rank, world
        self.stats["site_sharded_over"] = world
        # A worker of `--devices ... --shard sites` prints nothing until its report (no progress bar): the launcher's
        # stall watchdog (run_multi_device) would take a long healthy run for a hang.  Every `heartbeat_s` seconds of
        # progress the rank says so on stderr; the launcher swallows these lines.
        self.heartbeat_s = heartbeat_s
        self._last_beat = time.monotonic()

    def _beat(self):
        if self.heartbeat_s is not None and time.monotonic() - self._last_beat >= self.heartbeat_s:
            self._last_beat = time.monotonic()
            print(f"{HEARTBEAT} rank {self.rank}: {self.stats['alignments']} alignments in {self.stats['launches']} launches",
                  file=sys.stderr, flush=True)

    def _launch_sharded(self, shape, group_items, writers, pending):
        import zlib
        from .dist import site_range
        engine = self.engines[0]
        N, L = shape
        lo, hi = site_range(L, self.world, self.rank)
        if self.group is not None:
            tag = [N, L, len(group_items), zlib.crc32("\n".join(os.path.basename(g[0]) for g in group_items).encode())]
            seen = self.group.allgather(tag)
            if any(list(t) != tag for t in seen):
                raise RuntimeError(f"site-sharded ranks disagree on the next launch: {seen} (do all ranks see the same files?)")
        local = np.ascontiguousarray(np.stack([g[1] for g in group_items])[:, :, lo:hi])
        t0 = time.perf_counter()
        preds = engine.forward_sharded(local, lo, hi, L)
        self.stats["forward_s"] += time.perf_counter() - t0
        self.stats["launches"] += 1
        self.stats["alignments"] += len(group_items)
        key = f"{N}x{L}"
        self.stats["shapes"][key] = self.stats["shapes"].get(key, 0) + len(group_items)
        if self.rank == 0:
            for (path, _idx, ids), pred in zip(group_items, preds):
                pending.append(writers.submit(self._write, path, pred, ids))
            while len(pending) > 8 * self.io_threads + len(group_items):
                pending.popleft().result()
        if self.progress is not None:
            self.progress(len(group_items))
        self._beat()

    def run(self, paths: Sequence[str]) -> dict:
        for p in paths:
            if not has_fasta_ext(p):
                raise ValueError("Input files must be fasta files (.fa or .fasta). Got " f"{p}")
        paths = sorted(paths)
        t_start = time.perf_counter()
        buckets: "OrderedDict[Tuple[int, int], list]" = OrderedDict()
        pending: deque = deque()
        lookahead = max(16, 2 * (self.batch or 16))
        with ThreadPoolExecutor(self.io_threads, thread_name_prefix="pf-load") as loaders, \
                ThreadPoolExecutor(self.io_threads, thread_name_prefix="pf-write") as writers:
            inflight: "deque[Tuple[str, Future]]" = deque()
            it = iter(paths)
            exhausted = False
            while True:
                while not exhausted and len(inflight) < lookahead:
                    try:
                        p = next(it)
                    except StopIteration:
                        exhausted = True
                        break
                    inflight.append((p, loaders.submit(self._load, p)))
                if not inflight:
                    break
                path, fut = inflight.popleft()
                t0 = time.perf_counter()
                idx, ids = fut.result()
                self.stats["load_wait_s"] += time.perf_counter() - t0
                self._beat()                 # (parsing a long run of files that fill no bucket is progress too)
                shape = (int(idx.shape[0]), int(idx.shape[1]))
                bucket = buckets.setdefault(shape, [])
                bucket.append((path, idx, ids))
                if len(bucket) >= (self.batch or auto_batch(shape[0], shape[1], token_budget=TOKEN_BUDGET * self.world)):
                    self._launch_sharded(shape, bucket, writers, pending)
                    buckets[shape] = []
            for shape in sorted(buckets):
                if buckets[shape]:
                    self._launch_sharded(shape, buckets[shape], writers, pending)
            t0 = time.perf_counter()
            while pending:
                pending.popleft().result()
            self.stats["write_wait_s"] += time.perf_counter() - t0
        if self.group is not None:
            self.group.barrier()
        self.stats["wall_s"] = time.perf_counter() - t_start
        return self.stats


def cli_engine(weights, device: int):
    """The engine the CLI drives.  ``PF_CLI_ENGINE_FACTORY=module:function`` swaps in a stand-in with the Engine
    interface (``function(weights, device)``): the CPU tests of the multi-rank plumbing, which have no GPU."""
    hook = os.environ.get("PF_CLI_ENGINE_FACTORY")
    if hook:
        import importlib
        # a test hook, never set by the product: say so on every use, so that a run through a stand-in cannot pass
        # for a GPU run (there is no CPU fallback: without the hook a missing library or GPU is an error)
        print(f"infer_alns: PF_CLI_ENGINE_FACTORY={hook} REPLACES the GPU engine (test hook)", file=sys.stderr)
        mod_name, fn = hook.split(":")
        return getattr(importlib.import_module(mod_name), fn)(weights, device)
    from .engine import Engine
    return Engine(weights, device=device)


def summarize(stats: dict, load_s: float = 0.0) -> dict:
    n, wall = stats["alignments"], stats.get("wall_s", 0.0)
    return {"alignments": n, "launches": stats["launches"], "shapes": stats["shapes"],
            "model_load_s": round(load_s, 4), "wall_s": round(wall, 6),
            "gpu_streams": stats.get("gpu_streams", 1),
            "forward_s": round(stats["forward_s"], 6),      # summed over the streams' host threads
            "load_wait_s": round(stats["load_wait_s"], 6), "write_wait_s": round(stats["write_wait_s"], 6),
            "alignments_per_s": round(n / wall, 3) if wall > 0 else None,
            "alignments_per_s_forward_only": round(n * stats.get("gpu_streams", 1) / stats["forward_s"], 3)
            if stats["forward_s"] > 0 else None}


def run_multi_device(script: str, argv: List[str], devices: Sequence[int], shard: str = "files") -> Tuple[int, List[dict]]:
    """One child process per GPU (``--worker r/W``): each on its own share of the files, or - ``shard="sites"`` -
    all of them on every file, each with its block of sites; the ranks then meet through the rendezvous the
    environment set here names (127.0.0.1, a free port, a run id).
    Children are started before any of them touches a GPU; the parent never does."""
    procs = []
    env = dict(os.environ)
    if shard == "sites":
        import socket
        import uuid
        s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        # (HSA_ENABLE_IPC_MODE_LEGACY=0: the pool's host driver only supports dmabuf IPC, which RCCL's intra-node
        # transports need; exported on the boxes already, pinned here for the children - DESIGN.md section 6)
        env.update({"WORLD_SIZE": str(len(devices)), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "PF_RUN_ID": uuid.uuid4().hex,
                    "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    for r, d in enumerate(devices):
        cmd = [sys.executable, script, *argv, "--device", str(d), "--worker", f"{r}/{len(devices)}", "--bench"]
        if shard == "sites":
            cmd += ["--shard", "sites"]
        procs.append(subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True,
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r)) if shard == "sites" else None))
    # One reader thread per child keeps its stderr drained (a rank blocked on a full 64 KB pipe while its peers wait
    # for it in a collective is a deadlock), the parent polls all children: in sites mode the ranks depend on each
    # other, so as soon as one exits non-zero - or no child has produced a line or exited for `stall_s` - the others,
    # parked in an all-reduce that will never complete, are terminated.
    lines: List[List[str]] = [[] for _ in procs]
    last_event = [time.monotonic()]

    def drain(k, pipe):
        for line in pipe:
            lines[k].append(line)
            last_event[0] = time.monotonic()
        pipe.close()
    readers = [threading.Thread(target=drain, args=(k, p.stderr), daemon=True) for k, p in enumerate(procs)]
    for t in readers:
        t.start()
    stall_s = float(os.environ.get("PF_CLI_STALL_TIMEOUT", "900"))
    failed, alive = None, set(range(len(procs)))
    while alive:
        for k in sorted(alive):
            code = procs[k].poll()
            if code is not None:
                alive.discard(k)
                last_event[0] = time.monotonic()
                if code != 0 and failed is None:
                    failed = (k, code)
        stalled = shard == "sites" and time.monotonic() - last_event[0] > stall_s
        if alive and shard == "sites" and (failed is not None or stalled):
            why = (f"rank {failed[0]} exited with code {failed[1]}" if failed is not None
                   else f"no rank made progress for {stall_s:.0f} s")
            print(f"infer_alns: {why}; terminating the other site-sharded ranks", file=sys.stderr)
            for k in alive:
                procs[k].terminate()
            deadline = time.monotonic() + 10
            for k in list(alive):
                try:
                    procs[k].wait(timeout=max(0.1, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    procs[k].kill()
                    procs[k].wait()
            if failed is None:
                failed = (-1, 124)
            alive.clear()
        if alive:
            time.sleep(0.05)
    for t in readers:
        t.join(timeout=5)
    # Exit code: the first rank that FAILED ON ITS OWN decides (or 124 for a stall) - not the -15 of a peer this
    # launcher terminated, which may have a lower rank number; a child ended by a signal counts as 128 + signal.
    def code_of(c):
        return 1 if c is None else (128 - c if c < 0 else c)
    rc, reports = (code_of(failed[1]) or 1) if failed is not None else 0, []
    for k, p in enumerate(procs):
        rc = rc or code_of(p.returncode)
        rep = None
        for line in "".join(lines[k]).splitlines():
            if line.startswith("{") and '"alignments"' in line:
                try:
                    rep = json.loads(line)
                    continue
                except ValueError:
                    pass
            if line.strip() and not line.startswith(HEARTBEAT):
                print(line, file=sys.stderr)
        if rep is not None:
            reports.append(rep)
    return rc, reports
