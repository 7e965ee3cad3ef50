#!/usr/bin/env python3
"""profiles/r06_batched_probe.py K [N] [sweeps] - K independent fits of the headline shape (N = 4096 unless given, T = 100, chi = 32, d = 4)
advanced by one launch chain (mpst_sweep_batch); prints the aggregate rate.  Run it under rocprofv3 --kernel-trace --stats (or the
--pmc passes of profiles/collect_profiles.sh) for the per-kernel figures of the batched (_b) kernels."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpstime_jl_amd as mt  # noqa: E402
from bench import make_inputs  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
nsw = int(sys.argv[3]) if len(sys.argv) > 3 else 3
T, chi, d, C = 100, 32, 4, 2
ds = make_inputs(N, T, d)
W0 = mt.generate_startingMPS(4, T, d, C, 1234)
engs = []
for k in range(K):
    e = mt.SweepEngine(0)
    e.set_batch_hint(K)
    e.set_options(chi_max=chi, eta=0.01, cutoff=1e-10, update_iters=1, loss="KLD", bbopt="TSGO", rescale=(False, True))
    e.set_dataset(0, ds.phi, ds.label_index, C)
    e.set_mps(W0)
    e.build_caches()
    engs.append(e)
for _ in range(3):                  # bond dimensions reach chi_max, graph captured
    mt.sweep_batch(engs)
t0 = time.perf_counter()
for _ in range(nsw):
    st = mt.sweep_batch(engs)
dt = time.perf_counter() - t0
print(json.dumps({"fits": K, "N": N, "aggregate_sweeps_per_s": K * nsw / dt, "ms_per_batched_sweep": 1e3 * dt / nsw,
                  "device_ms": 1e3 * st[0]["seconds"], "info": {k: v for k, v in engs[0].info().items() if k in ("ksplit", "sliced_gemms", "graph")}}))
for e in engs:
    e.close()
