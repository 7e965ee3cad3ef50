"""Checks gpurun_out/tri_ab_T.txt (written by tri_ab.bin): eigenvalues of every variant's T against eigvalsh(G)."""
import sys
import numpy as np
from scipy.linalg import eigvalsh_tridiagonal
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tri_ab_T.txt"
lines = open(path).read().split("\n")
i = 0
Ts = {}
G = None
while i < len(lines):
    ln = lines[i].strip()
    if ln.startswith("variant"):
        k = int(ln.split()[1]); i += 1; rows = []
        while i < len(lines) and lines[i] and not lines[i].startswith(("variant", "G")):
            rows.append([float(x) for x in lines[i].split()]); i += 1
        Ts[k] = np.array(rows)
    elif ln.startswith("G"):
        n = int(ln.split()[1]); vals = [float(x) for x in lines[i + 1:i + 1 + n * n]]
        G = np.array(vals).reshape(n, n); i += 1 + n * n
    else:
        i += 1
lam = np.linalg.eigvalsh(G)
for k, T in Ts.items():
    d, e = T[:, 0], T[:-1, 1]
    lt = eigvalsh_tridiagonal(d, e)
    err = np.abs(lt - lam).max() / np.abs(lam).max()
    print(f"variant {k}: n={len(d)} max |lam(T) - lam(G)| / lam_max = {err:.3e}  {'OK' if err < 1e-13 else 'BAD'}")
