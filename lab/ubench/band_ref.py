"""NumPy model of stage 2 of a two-stage tridiagonalisation: symmetric band (half bandwidth b) -> tridiagonal by
Householder bulge chasing (one column per sweep, only the first column of each bulge is annihilated), plus a check of
the pipelined schedule the HIP prototype uses (sweep s, step k runs at tick stagger*s + k)."""
import sys
import numpy as np


def steps_of(n, b, s):
    out = []
    k = 0
    while True:
        r0 = s + 1 if k == 0 else s + k * b + 1
        nr = min(b, n - r0)
        if nr < 2:
            break
        c0 = s if k == 0 else r0 - b
        out.append((k, r0, nr, c0))
        k += 1
    return out


def do_step(A, n, b, r0, nr, c0):
    x = A[r0:r0 + nr, c0].copy()
    al = x[0]
    s2 = float(np.dot(x[1:], x[1:]))
    if s2 == 0.0:
        return
    nrm = np.sqrt(al * al + s2)
    beta = -np.copysign(nrm, al)
    tau = (beta - al) / beta
    v = x / (al - beta)
    v[0] = 1.0
    H = np.eye(nr) - tau * np.outer(v, v)
    lo = c0
    hi = min(n, r0 + nr + b)
    A[r0:r0 + nr, lo:hi] = H @ A[r0:r0 + nr, lo:hi]
    A[lo:hi, r0:r0 + nr] = A[lo:hi, r0:r0 + nr] @ H


def reduce_sequential(A0, b):
    A = A0.copy()
    n = A.shape[0]
    for s in range(n - 2):
        for (k, r0, nr, c0) in steps_of(n, b, s):
            do_step(A, n, b, r0, nr, c0)
    return A


def reduce_pipelined(A0, b, stagger, rng):
    A = A0.copy()
    n = A.shape[0]
    ticks = {}
    for s in range(n - 2):
        for st in steps_of(n, b, s):
            ticks.setdefault(stagger * s + st[0], []).append(st)
    for t in sorted(ticks):
        lst = ticks[t]
        for i in rng.permutation(len(lst)):
            (k, r0, nr, c0) = lst[i]
            do_step(A, n, b, r0, nr, c0)
    return A


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    rng = np.random.default_rng(3)
    for b in (2, 4, 8, 16):
        M = rng.standard_normal((n, n))
        M = M + M.T
        i, j = np.indices((n, n))
        M[np.abs(i - j) > b] = 0.0
        T = reduce_sequential(M, b)
        off = np.abs(np.tril(T, -2)).max()
        ev = np.abs(np.linalg.eigvalsh(np.tril(np.triu(T, -1), 1)) - np.linalg.eigvalsh(M)).max()
        width = max(abs(ii - jj) for ii, jj in zip(*np.nonzero(np.abs(T) > 0))) if off > 0 else 1
        nst = sum(len(steps_of(n, b, s)) for s in range(n - 2))
        print(f"b={b}: sequential: max below 2nd subdiagonal {off:.1e}, eig err {ev:.1e}, steps {nst}")
        for stg in (2, 3):
            Tp = reduce_pipelined(M, b, stg, rng)
            print(f"   stagger {stg}: max |T_pipelined - T_sequential| = {np.abs(Tp - T).max():.2e}")
