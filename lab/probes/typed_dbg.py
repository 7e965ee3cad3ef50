"""Probe: first failing bond of a typed sweep on the bench's inputs. usage: typed_dbg.py N T chi d dtype"""
import sys
import numpy as np
sys.path.insert(0, ".")
import mpstime_jl_amd as mt
import bench

N, T, chi, d = map(int, sys.argv[1:5])
dt = np.dtype(bench.NP_DT[sys.argv[5]])
cx = dt.kind == "c"
C = 1 if cx else 2
full = bench.typed_inputs(N, T, d, C, cx)
W0 = mt.generate_startingMPS(4, T, d, C, 1234, dt)
eng = mt.SweepEngine(0)
eng.set_options(chi_max=chi, eta=0.01)
eng.set_dataset(0, full.phi, full.label_index, C, dtype=dt)
eng.set_mps(W0)
eng.build_caches()
print("eval", eng.eval(0)[:3], flush=True)
for q in range(2 * (T - 1)):
    gl = q < T - 1
    lid = T - 2 - q if gl else q - (T - 1)
    try:
        tr = eng.bond_step(lid, gl)
    except Exception as e:
        print("FAILED at q", q, "lid", lid, e, flush=True)
        break
    if q % 20 == 0 or not np.isfinite(tr["loss"]):
        print(q, lid, tr["loss"], tr["grad_norm"], tr["chi"], tr["S"][:3], tr["S"][-1], flush=True)
print(eng.info())
