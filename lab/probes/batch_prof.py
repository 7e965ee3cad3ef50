"""Probe: K fits advanced by mpst_sweep_batch (run under rocprofv3 --kernel-trace --stats). usage: batch_prof.py K"""
import sys
import time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mpstime_jl_amd as mt
import bench

K = int(sys.argv[1])
full = bench.make_inputs(4096, 100, 4)
W0 = mt.generate_startingMPS(4, 100, 4, 2, 1234)
engs = []
for k in range(K):
    e = mt.SweepEngine(0)
    e.set_batch_hint(K)
    e.set_options(chi_max=32, eta=0.01)
    e.set_dataset(0, full.phi, full.label_index, 2)
    e.set_mps(W0)
    e.build_caches()
    engs.append(e)
for _ in range(3):
    mt.sweep_batch(engs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    mt.sweep_batch(engs)
torch.cuda.synchronize()
print(f"K={K}: {1e3 * (time.perf_counter() - t0) / 3:.2f} ms per batched sweep", flush=True)
