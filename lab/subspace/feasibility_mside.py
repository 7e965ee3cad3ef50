"""Step 0, second part: the iteration a device kernel would actually run at large bonds (n = d chi > 128).
G X has the conditioning of the SQUARED spectrum (lambda_1 / lambda_p up to 1e12: a Cholesky-QR of it breaks down in fp64),
so the half-steps go through M itself:  Y = M X, Q = cholqr(Y), Z = M^H Q, X = cholqr(Z)  (cond sigma_1 / sigma_p each),
Rayleigh-Ritz on H = (M X)^H (M X).  Orthonormalisation is done the way the device would do it - Gram matrix, Cholesky with
clamped pivots (rank-deficient blocks of the growth phase), triangular solve, optionally twice - NOT with LAPACK's Householder QR.
Start blocks: the row space of the label-carrying site (known before the solve) completed to p columns by M^H Omega (a random
sample of M's row space) or by plain random vectors.
usage: python lab/subspace/feasibility_mside.py dump.npz out.json chi_max p1,p2,... [max_bonds]"""
import json, sys
import numpy as np
from feasibility import trunc

def cholqr(Y, passes=1, tol=1e-13):
    for _ in range(passes):
        S = Y.conj().T @ Y
        S = 0.5 * (S + S.conj().T)
        p = S.shape[0]
        d0 = np.real(np.diag(S)).copy()
        L = np.zeros_like(S)
        dead = np.zeros(p, dtype=bool)
        A = S.copy()
        for j in range(p):                       # right-looking Cholesky, pivots clamped: a dependent column is dropped (zeroed)
            piv = np.real(A[j, j])
            if not (piv > tol * max(d0[j], 1e-300)) or d0[j] <= 1e-300 * 1.0:
                dead[j] = True
                A[j:, j] = 0; A[j, j:] = 0
                L[j, j] = 1.0
                continue
            l = A[j:, j] / np.sqrt(piv)
            L[j:, j] = l
            A[j + 1:, j + 1:] -= np.outer(l[1:], l[1:].conj())
        Y = np.linalg.solve(L, Y.conj().T).conj().T   # Y L^-H
        Y[:, dead] = 0
    return Y

def start(M, R0, p, kind, rng):
    n = M.shape[1]
    X = R0.conj().T
    if X.shape[1] >= p:
        return cholqr(X[:, :p], 2)
    extra = p - X.shape[1]
    if kind == "mtom":
        Om = rng.standard_normal((M.shape[0], extra))
        X = np.concatenate([X, M.conj().T @ Om], axis=1)
    else:
        X = np.concatenate([X, rng.standard_normal((n, extra))], axis=1)
    # columns of very different scale: normalise them first (a diagonal scaling, free on the device)
    X = X / np.maximum(np.linalg.norm(X, axis=0), 1e-300)
    return cholqr(X, 2)

def judge(M, sig, V, k, chi_max, cutoff, X, total):
    B = M @ X
    H = B.conj().T @ B
    H = 0.5 * (H + H.conj().T)
    w, Zv = np.linalg.eigh(H)
    w, Zv = np.maximum(w[::-1], 0), Zv[:, ::-1]
    kk = trunc(w, total, chi_max, cutoff)
    if kk != k:
        return False
    errS = np.abs(np.sqrt(w[:k]) - sig[:k]).max() / sig[0]
    Xk = X @ Zv[:, :k]
    errP = np.linalg.norm(M @ Xk @ Xk.conj().T - (M @ V[:, :k]) @ V[:, :k].conj().T) / np.sqrt(total)
    return errS <= 1e-10 and errP <= 1e-9

def need(M, R0, sig, V, k, chi_max, cutoff, p, kind, passes, qmax, rng, total):
    X = start(M, R0, p, kind, rng)
    for q in range(qmax + 1):
        if q:
            Q = cholqr(M @ X, passes)
            X = cholqr(M.conj().T @ Q, passes)
        if judge(M, sig, V, k, chi_max, cutoff, X, total):
            return q
    return qmax + 1

def main():
    z = np.load(sys.argv[1])
    meta = z["meta"]
    chi_max = int(sys.argv[3])
    PS = [int(x) for x in sys.argv[4].split(",")]
    nb = len(meta) if len(sys.argv) < 6 else min(len(meta), int(sys.argv[5]))
    step = max(1, len(meta) // nb)
    rng = np.random.default_rng(3)
    rows = []
    for i in range(0, len(meta), step):
        M, R0 = z[f"M{i}"], z[f"R{i}"]
        n = M.shape[1]
        U, sig, Vh = np.linalg.svd(M, full_matrices=False)
        V = Vh.conj().T
        lam = sig ** 2
        total = lam.sum()
        k = trunc(lam, total, chi_max, 1e-10)
        row = dict(i=int(i), lid=int(meta[i][0]), gl=int(meta[i][1]), sweep=int(meta[i][2]), n=int(n), m=int(M.shape[0]), k=int(k), r0=int(R0.shape[0]),
                   s_k=float(sig[k - 1] / sig[0]), s_k1=float(sig[k] / sig[0]) if k < len(sig) else 0.0)
        for p in PS:
            if p >= n or p <= k:
                continue
            row[f"s_{p}"] = float(sig[p] / sig[0]) if p < len(sig) else 0.0
            for kind in ("mtom", "rand"):
                for passes in (1, 2):
                    row[f"{kind}{p}x{passes}"] = need(M, R0, sig, V, k, chi_max, 1e-10, p, kind, passes, 12, rng, total)
        rows.append(row)
        print(len(rows), {a: (round(b, 7) if isinstance(b, float) else b) for a, b in row.items()}, flush=True)
    keys = sorted({a for r in rows for a in r if a[:4] in ("mtom", "rand")})
    def summ(key):
        v = np.array([r[key] for r in rows if key in r])
        return dict(count=int(len(v)), median=float(np.median(v)), p90=float(np.percentile(v, 90)), max=int(v.max()), mean=float(v.mean()),
                    hist=np.bincount(v).tolist())
    out = dict(source=sys.argv[1], bonds=len(rows), chi_max=chi_max,
               criteria="kept sigma 1e-10 sigma_1, ||M P~ - M P||_F <= 1e-9 ||M||_F, same kept dimension; value = two-sided iterations needed (13 = not within 12)",
               iterations={a: summ(a) for a in keys}, rows=rows)
    json.dump(out, open(sys.argv[2], "w"), indent=0)
    print(json.dumps(out["iterations"], indent=1))

if __name__ == "__main__":
    main()
