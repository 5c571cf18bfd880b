"""Throughput of the softmax MultiHeadAttention operator on one MI355X.
    python tools/mha_bench.py [--rows 1770] [--cols 500] [--iters 5]
Default shape = the row attention of one 60 x 500 alignment (R = 1770 pairs, C = 500 sites).
Algorithmic flops per token: 4 projections x 2*64*64 + QK^T and PV 2 * 2 * C * 64."""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from phyloformer_amd.attention import MultiHeadAttention

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1770)
ap.add_argument("--cols", type=int, default=500)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
rng = np.random.default_rng(0)
sd = {}
for p in ("q_proj", "k_proj", "v_proj", "out_proj"):
    sd[p + ".weight"] = (rng.standard_normal((64, 64)) / 8).astype(np.float32)
    sd[p + ".bias"] = (rng.standard_normal(64) * 0.1).astype(np.float32)
m = MultiHeadAttention(4, 64).load_state_dict(sd)
lib, h = m._lib, m._h
x = rng.standard_normal((1, a.rows, a.cols, 64)).astype(np.float32)
nbytes = x.nbytes
dx, dy = C.c_void_p(), C.c_void_p()
assert lib.pf_device_malloc(h, nbytes, C.byref(dx)) == 0 and lib.pf_device_malloc(h, nbytes, C.byref(dy)) == 0
assert lib.pf_memcpy_h2d(h, dx, x.ctypes.data, nbytes) == 0
m.forward_device(dx.value, 1, a.rows, a.cols, dy.value)          # warm-up (workspace allocation)
lib.pf_synchronize(h)
t0 = time.perf_counter()
for _ in range(a.iters):
    m.forward_device(dx.value, 1, a.rows, a.cols, dy.value)
lib.pf_synchronize(h)
wall = (time.perf_counter() - t0) / a.iters
lib.pf_set_option(h, b"profile", 1)
lib.pf_profile_reset(h)
for _ in range(a.iters):
    m.forward_device(dx.value, 1, a.rows, a.cols, dy.value)
lib.pf_synchronize(h)
per = {}
for k in ("mha_qkv", "mha_attn", "mha_out"):
    n, ms = C.c_int64(), C.c_double()
    lib.pf_profile_get(h, k.encode(), C.byref(n), C.byref(ms))
    per[k] = round(ms.value / max(1, n.value), 4)
tokens = a.rows * a.cols
flops_attn = tokens * 2 * 2 * a.cols * 64
flops_proj = tokens * 4 * 2 * 64 * 64
print(json.dumps({"shape": [1, a.rows, a.cols, 64], "ms_per_call": round(wall * 1e3, 4), "kernel_ms": per,
                  "tokens_per_s": round(tokens / wall), "algorithmic_tflops": round((flops_attn + flops_proj) / wall / 1e12, 2),
                  "attn_kernel_tflops": round(flops_attn / (per["mha_attn"] * 1e-3) / 1e12, 2) if per["mha_attn"] else None}))
