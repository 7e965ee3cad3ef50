"""Which bonds does the subspace eigensolver hand back to the exact one, and why?  (MPST_BIG_SYNC=1 MPST_SS_DEBUG=1: verdict per bond on stderr)
usage: MPST_BIG_SYNC=1 MPST_SS_DEBUG=1 python lab/probes/subspace_rejects.py N T chi d C dtype sweeps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpstime_jl_amd as mt
import bench
N, T, chi, d, C = [int(x) for x in sys.argv[1:6]]
dt = np.dtype(sys.argv[6]); sweeps = int(sys.argv[7])
full = bench.typed_inputs(N, T, d, C, dt.kind == "c")
W0 = mt.generate_startingMPS(4, T, d, C, 1234, dt)
eng = mt.SweepEngine(0)
eng.set_options(chi_max=chi, eta=0.01, cutoff=1e-10)
eng.set_dataset(0, full.phi, full.label_index, C, dtype=dt)
eng.set_mps(W0)
eng.build_caches()
for s in range(sweeps):
    print("=== sweep", s, file=sys.stderr, flush=True)
    st = eng.sweep()
    i = eng.info()
    print("sweep", s, st, i["subspace_attempted"], i["subspace_accepted"], flush=True)
eng.close()
