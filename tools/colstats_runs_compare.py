"""k_colstats by groups (colstats_fine 0) against by runs (1) and the batch-dependent default (-1): per-launch
times, whole-forward time with the indices resident in HBM, and bit-identity of the distances."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from phyloformer_amd.engine import Engine
from phyloformer_amd.weights import load_weights
from phyloformer_amd.msa_sim import simulate_batch
w = load_weights("models/pf.ckpt")
cases = [(1, 20, 200), (4, 20, 200), (1, 60, 500), (2, 60, 500), (4, 60, 500), (16, 60, 500), (1, 60, 2000), (1, 200, 500)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, n, l in cases:
    idx = np.ascontiguousarray(np.resize(simulate_batch(min(B, 4), n, l, seed=3), (B, n, l)))
    P = n * (n - 1) // 2
    ref = None
    for fine in (0, 1, -1):
        e = Engine(w, 0)
        e.set_option("colstats_fine", fine)
        out = e.forward(idx)
        if ref is None: ref = out
        same = np.array_equal(ref.view(np.uint32), out.view(np.uint32))
        d_idx = e.malloc(idx.nbytes); d_out = e.malloc(B * P * 4); e.h2d(d_idx, idx)
        for _ in range(2): e.forward_device(d_idx, B, n, l, d_out)
        e.synchronize()
        reps = 10 if B * P * l < 5e6 else 4
        t0 = time.perf_counter()
        for _ in range(reps): e.forward_device(d_idx, B, n, l, d_out)
        e.synchronize()
        dt = (time.perf_counter() - t0) / reps
        e.set_option("profile", 1); e.profile_reset()
        for _ in range(3): e.forward(idx)
        c, ms = e.profile_get("colstats"); c2, ms2 = e.profile_get("colfin")
        print(f"{n:3d}x{l:4d} batch {B:2d} fine {fine:2d}: forward {dt * 1e3:8.3f} ms ({B / dt:8.1f} aln/s)  colstats "
              f"{ms / c * 1e3:7.1f} us  colfin {ms2 / c2 * 1e3:6.1f} us  bits {'same' if same else 'DIFFER'}", flush=True)
        e.close()
