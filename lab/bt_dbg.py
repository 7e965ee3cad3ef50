"""Bring-up of the blocked large-bond eigensolver: accuracy and wall time of mpst_selftest_eig."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpstime_jl_amd as mt
eng = mt.SweepEngine(0)
for n in [int(a) for a in sys.argv[1:]] or [130]:
    rng = np.random.default_rng(n)
    A = rng.standard_normal((2 * n, n)) * (0.9 ** np.arange(n))
    G = A.T @ A
    for alg in (3, 2):
        t0 = time.perf_counter()
        lam, E, info = eng.selftest_eig(G, alg=alg)
        dt = time.perf_counter() - t0
        w = np.linalg.eigvalsh(G)[::-1]
        K = min(n, 128)
        Ek = E[:, :K]
        print(f"n={n} alg={alg} info={info} wall={dt*1e3:.1f} ms  lam_err={np.abs(lam[:K]-w[:K]).max()/w[0]:.2e} "
              f"orth={np.abs(Ek.T@Ek-np.eye(K)).max():.2e} resid={np.abs(G@Ek-Ek*lam[:K]).max()/w[0]:.2e} dbg={lam[K:K+2] if n > K + 1 else None}", flush=True)
        if alg == 3:
            print("   lam err per k (first 8):", np.abs(lam[:8] - w[:8]) / w[0], "worst k", int(np.argmax(np.abs(lam[:K]-w[:K]))))
