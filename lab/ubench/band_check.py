"""Check gpurun_out/band2tri_T.txt: the tridiagonal (d, e) each band2tri case produced has the spectrum of its band input."""
import sys
import numpy as np

lines = open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/band2tri_T.txt").read().split("\n")
i = 0
bad = 0
while i < len(lines):
    if not lines[i].startswith("case"):
        i += 1
        continue
    _, n, b = lines[i].split()
    n, b = int(n), int(b)
    band = np.array([[float(x) for x in lines[i + 1 + j].split()] for j in range(n)])
    de = np.array([[float(x) for x in lines[i + 1 + n + j].split()] for j in range(n)])
    i += 1 + 2 * n
    A = np.zeros((n, n))
    for j in range(n):
        for d in range(b + 1):
            if j + d < n:
                A[j + d, j] = A[j, j + d] = band[j, d]
    T = np.diag(de[:, 0]) + np.diag(de[:-1, 1], 1) + np.diag(de[:-1, 1], -1)
    la, lt = np.linalg.eigvalsh(A), np.linalg.eigvalsh(T)
    err = np.abs(la - lt).max() / np.abs(la).max()
    ok = err < 1e-13
    bad += not ok
    print(f"n={n} b={b}: max |lam(T) - lam(band)| / lam_max = {err:.3e}  {'OK' if ok else 'WRONG'}")
sys.exit(1 if bad else 0)
