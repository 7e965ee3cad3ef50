"""Bring-up helper: per-phase cycle stamps of one tridiagonalisation step (build with EXTRA=-DMPST_TRI_DEBUG)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpstime_jl_amd as mt
if os.environ.get("MPST_LIB"): mt._lib.LIB_PATH = os.environ["MPST_LIB"]
from tests.helpers import make_problem
ds, W = make_problem(512, 16, 4, 4, 2, seed=5)
eng = mt.SweepEngine(0)
eng.set_options(chi_max=32, eta=0.01)
eng.set_dataset(0, ds.phi, ds.label_index, 2)
eng.set_mps(W)
eng.build_caches()
for _ in range(3):
    eng.sweep()
for lid in range(14, 6, -1):      # the label sits on the last site after a sweep: walk left to a full-size bond
    eng.bond_step(lid, True)
out = (C.c_ulonglong * 64)()
eng.lib.mpst_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
eng.lib.mpst_debug_stamps(eng.ctx, out)
s = list(out)
print("chi", eng.get_chi()[0])
print("total cycles", s[7] - s[6], "per step", (s[7] - s[6]) / 127)
for name, b in (("live w12", 16), ("helper w0", 24), ("owner w7", 32)):
    t0 = s[b]
    print(name, [int(x - t0) if x else None for x in s[b:b + 8]])
print("phases", eng.eig_phases())
v = s[40:48]
print("k_eig_vec (10ns ticks) from stamp2:", [int(x - s[2]) for x in v], "end-of-bisect/eigvec/backtr:", int(s[3]-s[2]), int(s[4]-s[2]), int(s[5]-s[2]))
print("gaps (10ns ticks): tri_end->vec_start", int(s[2]-s[1]), " vec_blk0_end->fin_start", int(s[8]-s[5]), " fin", int(s[9]-s[8]), " tri", int(s[1]-s[0]), " vec blk0", int(s[5]-s[2]))
