import os, sys, subprocess, hashlib
sys.path.insert(0, "/root/repo")
import numpy as np
code = r'''
import sys; sys.path.insert(0, ".")
import numpy as np, hashlib
import mpstime_jl_amd as mt, bench
full = bench.make_inputs(4096, 100, 4)
W0 = mt.generate_startingMPS(4, 100, 4, 2, 1234)
eng = mt.SweepEngine(0); eng.set_options(chi_max=32, eta=0.01, cutoff=1e-10); eng.set_dataset(0, full.phi, full.label_index, 2)
for rep in range(3):
    eng.set_mps(W0); eng.build_caches()
    for s in range(2): eng.sweep()
    W = eng.get_mps()
    h = hashlib.sha256(b"".join(np.ascontiguousarray(w).tobytes() for w in W)).hexdigest()[:16]
    print("digest", h, eng.info().get("svd_fallbacks"))
'''
for env in ({}, {"MPST_EIG_SPLIT": "1"}):
    e = dict(os.environ); e.update(env)
    print(env, subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, cwd=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")).stdout)
