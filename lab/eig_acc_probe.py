import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpstime_jl_amd as mt
eng = mt.SweepEngine(0)
for n, K in [(128, 32), (100, 25), (64, 32), (40, 16)]:
    errs, ress, orth = [], [], []
    for seed in range(20):
        rng = np.random.default_rng(seed)
        A = rng.standard_normal((2 * n, n)) * (0.85 ** np.arange(n))
        G = A.T @ A
        lam, E, info = eng.selftest_eig(G)
        w = np.linalg.eigvalsh(G)[::-1]
        k = min(K, len(lam))
        errs.append(np.abs(lam[:k] - w[:k]).max() / w[0])
        Ek = E[:, :k]
        ress.append(np.abs(G @ Ek - Ek * lam[:k]).max() / w[0])
        orth.append(np.abs(Ek.T @ Ek - np.eye(k)).max())
    print(f"n={n}: max eigenvalue error / lam_max {max(errs):.2e}, residual {max(ress):.2e}, orthogonality {max(orth):.2e}, info {info}", flush=True)
eng.close()
