"""Bring-up: in-kernel phase stamps of k_bond_fused (build with EXTRA=-DMPST_FUSED_DEBUG)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpstime_jl_amd as mt
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
full = bench.make_inputs(N, 100, 4)
W0 = mt.generate_startingMPS(4, 100, 4, 2, 1234)
eng = mt.SweepEngine(0)
eng.set_options(chi_max=32, eta=0.01)
eng.set_dataset(0, full.phi, full.label_index, 2)
eng.set_mps(W0); eng.build_caches()
for _ in range(3): eng.sweep()
eng.lib.mpst_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
for rep in range(3):
    eng.set_mps(W0); eng.build_caches()
    for q in range(99): eng.bond_step(98 - q, True)
    for q in range(30 + rep): eng.bond_step(q, False)
    st = (C.c_ulonglong * 64)()
    eng.lib.mpst_debug_stamps(eng.ctx, st)
    t = np.array(st[32:62], dtype=np.float64)
    t = t[t > 0]
    print("phases (us):", np.round((t[1:] - t[:-1]) * 0.01, 2), "total", (t[-1] - t[0]) * 0.01)
eng.set_profile(0x7FF)
for q in range(32, 60): eng.bond_step(q, False)
print({k: (round(v[0] / max(v[1], 1), 2), v[1]) for k, v in eng.get_profile().items()})
