"""Probe: aggregate sweeps/s of K independent fits sharing the GPU (threads, one context + stream + hipGraph each).
usage: concurrent_fits.py K [N T chi d]   (environment: GPU_MAX_HW_QUEUES, MPST_EIG_SPLIT, ...)"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
import torch
import mpstime_jl_amd as mt
import bench

K = int(sys.argv[1])
N, T, chi, d = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (4096, 100, 32, 4)
full = bench.make_inputs(N, T, d)
W0 = mt.generate_startingMPS(4, T, d, 2, 1234)
engs = []
for k in range(K):
    e = mt.SweepEngine(0)
    e.set_options(chi_max=chi, eta=0.01)
    e.set_dataset(0, full.phi, full.label_index, 2)
    e.set_mps(W0)
    e.build_caches()
    for _ in range(3):
        e.sweep()
    engs.append(e)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    engs[0].sweep()
torch.cuda.synchronize()
single = 5 / (time.perf_counter() - t0)
nsw = 5


def run(e):
    for _ in range(nsw):
        e.sweep()


t0 = time.perf_counter()
ths = [threading.Thread(target=run, args=(e,)) for e in engs]
[t.start() for t in ths]
[t.join() for t in ths]
torch.cuda.synchronize()
agg = K * nsw / (time.perf_counter() - t0)
print(f"K={K} env(GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}, MPST_EIG_SPLIT={os.environ.get('MPST_EIG_SPLIT')}): "
      f"single {single:.2f} sweeps/s, aggregate {agg:.2f} sweeps/s, ratio {agg / single:.2f}", flush=True)
