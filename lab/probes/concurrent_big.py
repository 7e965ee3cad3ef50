"""Several large-bond fits at once on one GPU (own context, stream and thread each): the XCD-local tridiagonalisations of
different contexts pick different XCDs (context ordinal), so they should neither abort nor misplace.  Prints the
aggregate sweeps/s and every engine's info()."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpstime_jl_amd as mt
from oracle import ref_numpy as R
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N, T, d, chi = 1024, 10, 8, 40
engs = []
for f in range(F):
    rng = np.random.default_rng(f)
    X = rng.uniform(-0.9, 0.9, (N, T))
    phi = R.legendre_encode(X, d)
    W = R.random_mps(T, d, chi, 1, rng)
    e = mt.SweepEngine(0)
    e.set_options(chi_max=chi, eta=0.01)
    e.set_dataset(0, phi, np.zeros(N, dtype=np.int32), 1)
    e.set_mps(W); e.build_caches(); e.sweep()
    engs.append(e)
def run(e, k):
    for _ in range(k):
        e.sweep()
t0 = time.time(); run(engs[0], 3); t1 = time.time() - t0
th = [threading.Thread(target=run, args=(e, 3)) for e in engs]
t0 = time.time()
for t in th: t.start()
for t in th: t.join()
tF = time.time() - t0
print(f"single fit {3 / t1:.2f} sweeps/s; {F} concurrent fits {3 * F / tF:.2f} sweeps/s aggregate")
for e in engs:
    i = e.info(); print({k: i[k] for k in ("persistent_tridiag_aborts", "xcd_local_misplaced", "library_eig_fallbacks")})
