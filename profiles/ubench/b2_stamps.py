#!/usr/bin/env python3
"""profiles/ubench/b2_stamps.py K [N] - in-kernel timelines (100 MHz stamps, -DMPST_B2_DEBUG library built by build_b2dbg.sh) of the sliced
bond GEMM kernels on the last bond of a batched sweep of K fits (K = 1: the single-fit kernels)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mpstime_jl_amd._lib as L  # noqa: E402

L.LIB_PATH = os.path.join(ROOT, "profiles", "ubench", "b2dbg" + os.environ.get("B2DBG_TAG", ""), "libmpstime_hip_b2dbg.so")
import mpstime_jl_amd as mt  # noqa: E402
from bench import make_inputs  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T, chi, d, Cc = 100, 32, 4, 2
ds = make_inputs(N, T, d)
W0 = mt.generate_startingMPS(4, T, d, Cc, 1234)
engs = []
for k in range(K):
    e = mt.SweepEngine(0)
    if K > 1:
        e.set_batch_hint(K)
    e.set_options(chi_max=chi, eta=0.01, cutoff=1e-10, update_iters=1, loss="KLD", bbopt="TSGO", rescale=(False, True))
    e.set_dataset(0, ds.phi, ds.label_index, Cc)
    e.set_mps(W0)
    e.build_caches()
    engs.append(e)
for _ in range(3):
    mt.sweep_batch(engs) if K > 1 else engs[0].sweep()
lib = L.load()
lib.mpst_debug_b2.restype = C.c_int
lib.mpst_debug_b2.argtypes = [C.c_void_p, C.c_void_p]
buf = np.zeros(8192 * 8, dtype=np.uint64)
rc = lib.mpst_debug_b2(engs[0].ctx, buf.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
st = buf.reshape(8192, 8).astype(np.int64)
g = st[:4096]
g = g[g[:, 0] > 0]
y = st[4096:]
y = y[y[:, 0] > 0]
out = {"fits": K, "N": N}
if len(g):
    t0 = g[:, 0].min()
    n = (g[:, 5] >> 48)
    stage = (g[:, 5] & ((1 << 48) - 1))
    out["grad_s"] = {"workgroups_of_fit0": int(len(g)), "start_us": [float(0.01 * (g[:, 0] - t0).min()), float(0.01 * (g[:, 0] - t0).max())],
                     "first_loads_arrived_after_us": float(0.01 * np.median(g[:, 6] - g[:, 0])),
                     "stages": float(np.median(n)), "load_wait_per_stage_us": float(0.01 * np.median(g[:, 7] / np.maximum(n, 1))),
                     "wait_plus_staging_per_stage_us": float(0.01 * np.median(stage / np.maximum(n, 1))),
                     "matrix_loop_end_us": float(0.01 * np.median(g[:, 1] - g[:, 0])), "end_us": float(0.01 * np.median(g[:, 4] - g[:, 0])),
                     "last_end_us": float(0.01 * (g[:, 4] - t0).max())}
if len(y):
    t0 = y[:, 0].min()
    dd = 0.01 * (y - y[:, :1])
    out["yhat_s"] = {"workgroups_of_fit0": int(len(y)), "start_us_max": float(0.01 * (y[:, 0] - t0).max()),
                     "median_us_since_start": {"dims": float(np.median(dd[:, 1])), "g0_rows_arrived": float(np.median(dd[:, 2])), "g0_B_fragment": float(np.median(dd[:, 3])),
                                               "g0_done": float(np.median(dd[:, 4])), "g1_rows_arrived": float(np.median(dd[:, 5])), "g1_B": float(np.median(dd[:, 6])),
                                               "g1_done": float(np.median(dd[:, 7]))}}
print(json.dumps(out, indent=1))
for e in engs:
    e.close()
