"""Bring-up: per-phase stamps of k_yhat_s / k_grad_s (library built with lab/build_dbg.sh b2dbg -DMPST_B2_DEBUG)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpstime_jl_amd as mt
mt._lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmpstime_hip_b2dbg.so")
import bench
N = int(os.environ.get("PROF_N", "4096")); chi = 32; d = 4; T = 100
full = bench.make_inputs(N, T, d)
W0 = mt.generate_startingMPS(4, T, d, 2, 1234)
eng = mt.SweepEngine(0)
eng.set_options(chi_max=chi, eta=0.01)
eng.set_dataset(0, full.phi, full.label_index, 2)
eng.set_mps(W0); eng.build_caches()
for _ in range(3):
    eng.sweep()
for lid in range(T - 2, 49, -1):     # the label sits on the last site after a sweep: walk it to a bulk bond
    eng.bond_step(lid, True)
out = (C.c_ulonglong * (8192 * 8))()
eng.lib.mpst_debug_b2.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
assert eng.lib.mpst_debug_b2(eng.ctx, out) == 0
a = np.array(out, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
g = a[:4096]
gl = g[g[:, 0] > 0]
print("k_grad_s workgroups with stamps:", len(gl))
t0 = gl[:, 0].min()
def us(x): return np.round(0.01 * x, 2)
print(" start spread (us): max", us(gl[:, 0].max() - t0))
for i, nm in ((6, "first stage loaded"), (7, "first stage in LDS"), (1, "loop end"), (2, "stores drained"), (3, "ticket known"), (4, "block published"), (5, "loss ticket done")):
    m = gl[:, i] > 0
    if m.any():
        dlt = gl[m, i] - gl[m, 0]
        print(f" {nm}: n={m.sum()} median {us(np.median(dlt))} max {us(dlt.max())} (since own start); last at {us(gl[m, i].max() - t0)} since first start")
y = a[4096:]
yl = y[y[:, 0] > 0]
print("k_yhat_s workgroups with stamps:", len(yl))
t0 = yl[:, 0].min()
print(" start spread max", us(yl[:, 0].max() - t0))
for i, nm in ((1, "dims known"), (2, "loads returned, B in LDS"), (3, "tile in LDS, B fragment read"), (4, "mfma + rowdot + store issued")):
    dlt = yl[:, i] - yl[:, 0]
    print(f" {nm}: median {us(np.median(dlt))} max {us(dlt.max())}; last at {us(yl[:, i].max() - t0)}")
