import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mpstime_jl_amd as mt
import bench
full = bench.make_inputs(4096, 100, 4)
W0 = mt.generate_startingMPS(4, 100, 4, 2, 1234)
eng = mt.SweepEngine(0)
eng.set_options(chi_max=32, eta=0.01, cutoff=1e-10)
eng.set_dataset(0, full.phi, full.label_index, 2)
eng.set_mps(W0); eng.build_caches()
for _ in range(2): eng.sweep()
torch.cuda.synchronize(); t0 = time.perf_counter()
dev = 0
for _ in range(5): dev += eng.sweep()["seconds"]
torch.cuda.synchronize(); t1 = time.perf_counter()
print("graph" if not os.environ.get("MPST_NO_GRAPH") else "stream", "sweeps/s %.3f  device ms/sweep %.3f" % (5 / (t1 - t0), dev / 5 * 1e3), eng.eval(0)[1])
