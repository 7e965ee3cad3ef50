"""SHA-256 of the MPS after two sweeps through the large-bond path (d*chi = 168 and 512-wide bonds are too slow here:
d=12, chi_max=14).  Run once per exchange variant of the blocked tridiagonalisation and compare the digests:
    python lab/probes/path_hash.py; MPST_BT_NO_XCD=1 python lab/probes/path_hash.py; MPST_BT_NO_COOP=1 python lab/probes/path_hash.py
The arithmetic does not depend on which workgroup owns which row, so the three digests must be equal."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpstime_jl_amd as mt
from oracle import ref_numpy as R
from tests.helpers import load_engine, make_problem

h = hashlib.sha256()
for (N, T, d, chi0, chimax) in ((40, 3, 12, 10, 14), (64, 6, 8, 20, 37)):
    ds, W0 = make_problem(N, T, d, chi0, 1, seed=N + d)
    opts = R.SweepOptions(nsweeps=1, chi_max=chimax, eta=0.05, loss_grad="KLD", bbopt="TSGO")
    eng = mt.SweepEngine(0)
    load_engine(eng, ds, W0, opts)
    eng.build_caches()
    eng.sweep(); eng.sweep()
    for t in eng.get_mps():
        h.update(np.ascontiguousarray(t).tobytes())
    info = eng.info()
    eng.close()
print("digest", h.hexdigest()[:32], {k: info[k] for k in ("large_bond", "persistent_tridiag_aborts", "xcd_local_misplaced", "library_eig_fallbacks")},
      {k: v for k, v in os.environ.items() if k.startswith("MPST_")})
