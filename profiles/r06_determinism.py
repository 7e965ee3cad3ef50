#!/usr/bin/env python3
"""profiles/r06_determinism.py [reps] - repeated fits on the same inputs must give the same bits: the four-launch chain (with and without
the cache rebuilds), the six-launch chain, batched sweeps of 8 and 20 fits (XCD-contiguous order, four-wave gradient kernel, two eigenpairs
at a time), N = 32768 (persistent pair).  Prints one line per configuration: distinct digests over the repetitions (must be 1)."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpstime_jl_amd as mt          # noqa: E402
from bench import make_inputs        # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def digest(W):
    h = hashlib.sha256()
    for t in W:
        h.update(np.ascontiguousarray(t).tobytes())
    return h.hexdigest()[:16]


def fit(ds, W0, chi, sweeps, rebuild, env):
    for k, v in env.items():
        os.environ[k] = v
    try:
        e = mt.SweepEngine(0)
        e.set_options(chi_max=chi, eta=0.01, cutoff=1e-10, update_iters=1, loss="KLD", bbopt="TSGO", rescale=(False, True), rebuild_caches=rebuild)
        e.set_dataset(0, ds.phi, ds.label_index, 2)
        e.set_mps(W0)
        e.build_caches()
        for _ in range(sweeps):
            e.sweep()
        d = digest(e.get_mps())
        e.close()
        return d
    finally:
        for k in env:
            os.environ.pop(k, None)


def batched(ds, W0, chi, K, sweeps):
    engs = []
    for k in range(K):
        e = mt.SweepEngine(0)
        e.set_batch_hint(K)
        e.set_options(chi_max=chi, eta=[0.01, 0.02, 0.005][k % 3], cutoff=1e-10, update_iters=1, loss="KLD", bbopt="TSGO", rescale=(False, True))
        e.set_dataset(0, ds.phi, ds.label_index, 2)
        e.set_mps(W0)
        e.build_caches()
        engs.append(e)
    for _ in range(sweeps):
        mt.sweep_batch(engs)
    d = hashlib.sha256("".join(digest(e.get_mps()) for e in engs).encode()).hexdigest()[:16]
    for e in engs:
        e.close()
    return d


T, d, chi = 100, 4, 32
ds = make_inputs(4096, T, d)
W0 = mt.generate_startingMPS(4, T, d, 2, 1234)
configs = [
    ("four launches, cache rebuilds, 4 sweeps", lambda: fit(ds, W0, chi, 4, True, {})),
    ("four launches, no rebuilds, 4 sweeps", lambda: fit(ds, W0, chi, 4, False, {})),
    ("six launches (MPST_CHAIN4=0), 4 sweeps", lambda: fit(ds, W0, chi, 4, False, {"MPST_CHAIN4": "0"})),
    ("8 fits per launch, 3 sweeps", lambda: batched(ds, W0, chi, 8, 3)),
    ("20 fits per launch, 3 sweeps", lambda: batched(ds, W0, chi, 20, 3)),
]
for name, fn in configs:
    got = [fn() for _ in range(reps)]
    print(f"{name}: {reps} repetitions, distinct digests {len(set(got))} ({got[0]})", flush=True)
# rebuilds on / off must agree with each other (bit-identical by construction)
a, b = fit(ds, W0, chi, 3, True, {}), fit(ds, W0, chi, 3, False, {})
print("rebuilds on == off:", a == b, flush=True)
ds2 = make_inputs(32768, T, d)
got = [fit(ds2, W0, chi, 2, False, {}) for _ in range(max(3, reps // 4))]
print(f"N = 32768 (persistent pair), 2 sweeps: {len(got)} repetitions, distinct digests {len(set(got))} ({got[0]})")
