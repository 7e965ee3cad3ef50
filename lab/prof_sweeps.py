"""Profiling driver: a few steady-state sweeps of the headline config (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpstime_jl_amd as mt
import bench
N = int(os.environ.get("PROF_N", "4096")); chi = int(os.environ.get("PROF_CHI", "32")); d = int(os.environ.get("PROF_D", "4"))
T = int(os.environ.get("PROF_T", "100")); nsw = int(os.environ.get("PROF_SWEEPS", "6"))
full = bench.make_inputs(N, T, d)
W0 = mt.generate_startingMPS(4, T, d, 2, 1234)
eng = mt.SweepEngine(0)
eng.set_options(chi_max=chi, eta=0.01)
eng.set_dataset(0, full.phi, full.label_index, 2)
eng.set_mps(W0); eng.build_caches()
ts = [eng.sweep()["seconds"] for _ in range(nsw)]
print("sweep ms:", np.round(1e3 * np.array(ts), 2), "KLD", eng.eval(0)[1])
