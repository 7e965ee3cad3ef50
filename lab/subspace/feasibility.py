"""Step 0: how many block iterations does a warm-started top-chi subspace solver need on the Gram
matrices of real sweeps?  CPU only, NumPy.  Input: dumps of lab/subspace/dump_grams.py.
For every bond: G = M^H M, exact eigh as the truth; candidates:
  subspace iteration (block p, start = row space of the label-carrying site, CholQR-equivalent orthonormalisation each
  step, Rayleigh-Ritz at the end), block Krylov (block b, s blocks, full reorthogonalisation, Rayleigh-Ritz on b*s).
Pass criteria (10x inside the GPU tests' tolerances): kept singular values to 1e-10 sigma_1, projected two-site tensor
||M P~ - M P||_F <= 1e-9 ||M||_F, same kept dimension by the NDTensors rule on (trace, Ritz values).
usage: python lab/subspace/feasibility.py dump.npz out.json [max_bonds]"""
import json, sys
import numpy as np

def trunc(P, total, maxdim, cutoff):
    n = len(P)
    err = total - P.sum()
    while n > maxdim:
        err += P[n - 1]; n -= 1
    while n > 1 and err + P[n - 1] <= cutoff * total:
        err += P[n - 1]; n -= 1
    return n

def orth(X):
    Q, _ = np.linalg.qr(X)
    return Q

def start_block(R0, n, p, rng):
    X = R0.conj().T
    if X.shape[1] > p:
        U, s, _ = np.linalg.svd(X, full_matrices=False)
        X = U[:, :p]
    elif X.shape[1] < p:
        X = np.concatenate([X, rng.standard_normal((n, p - X.shape[1]))], axis=1)
    return orth(X)

def ritz(G, Q):
    H = Q.conj().T @ G @ Q
    H = 0.5 * (H + H.conj().T)
    w, Z = np.linalg.eigh(H)
    return w[::-1], Q @ Z[:, ::-1]

def judge(M, lam, V, k, chi_max, cutoff, w, X, total):
    """(ok, errS, errP, k~)"""
    w = np.maximum(w, 0.0)
    kk = trunc(w[:min(len(w), max(chi_max, 1) + 0) if False else len(w)], total, chi_max, cutoff)
    nk = min(k, kk)
    errS = np.abs(np.sqrt(w[:nk]) - np.sqrt(lam[:nk])).max() / np.sqrt(lam[0])
    A = M @ X[:, :k] @ X[:, :k].conj().T if kk >= k else None
    if A is None:
        return False, errS, np.inf, kk
    errP = np.linalg.norm(A - M @ V[:, :k] @ V[:, :k].conj().T) / np.linalg.norm(M)
    return (kk == k and errS <= 1e-10 and errP <= 1e-9), errS, errP, kk

def need_subspace(M, G, R0, lam, V, k, chi_max, cutoff, p, qmax, rng, total):
    X = start_block(R0, G.shape[0], p, rng)
    for q in range(0, qmax + 1):
        if q:
            X = orth(G @ X)
        w, Y = ritz(G, X)
        ok, eS, eP, kk = judge(M, lam, V, k, chi_max, cutoff, w, Y, total)
        if ok:
            return q
    return qmax + 1

def need_krylov(M, G, R0, lam, V, k, chi_max, cutoff, b, smax, rng, total):
    X = start_block(R0, G.shape[0], b, rng)
    K = X
    for s in range(1, smax + 1):
        if s > 1:
            Y = G @ K[:, -b:]
            Y = Y - K @ (K.conj().T @ Y)
            Y = Y - K @ (K.conj().T @ Y)
            K = np.concatenate([K, orth(Y)], axis=1)
        w, Yv = ritz(G, K)
        ok, *_ = judge(M, lam, V, k, chi_max, cutoff, w, Yv, total)
        if ok:
            return s
    return smax + 1

def main():
    z = np.load(sys.argv[1])
    meta = z["meta"]
    nb = len(meta) if len(sys.argv) < 4 else min(len(meta), int(sys.argv[3]))
    chi_max = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    cutoff = 1e-10
    rng = np.random.default_rng(7)
    rows = []
    step = max(1, len(meta) // nb)
    for i in range(0, len(meta), step):
        M, R0 = z[f"M{i}"], z[f"R{i}"]
        n = M.shape[1]
        G = M.conj().T @ M
        G = 0.5 * (G + G.conj().T)
        lam, V = np.linalg.eigh(G)
        lam, V = np.maximum(lam[::-1], 0), V[:, ::-1]
        total = lam.sum()
        k = trunc(lam, total, chi_max, cutoff)
        row = dict(i=int(i), lid=int(meta[i][0]), gl=int(meta[i][1]), sweep=int(meta[i][2]), n=int(n), m=int(M.shape[0]), k=int(k),
                   r0=int(R0.shape[0]),
                   s_k=float(np.sqrt(lam[k - 1] / lam[0])), s_k1=float(np.sqrt(lam[k] / lam[0])) if k < n else 0.0)
        for p in PS:
            if p >= n:
                continue
            row[f"sub{p}"] = need_subspace(M, G, R0, lam, V, k, chi_max, cutoff, p, 24, rng, total)
            row[f"s_{p}"] = float(np.sqrt(lam[p] / lam[0]))
        for b, smax in KR:
            if b * 2 > n:
                continue
            row[f"kry{b}"] = need_krylov(M, G, R0, lam, V, k, chi_max, cutoff, b, smax, rng, total)
        rows.append(row)
        if len(rows) % 20 == 0:
            print(len(rows), row, flush=True)
    def summ(key):
        v = np.array([r[key] for r in rows if key in r])
        if not len(v):
            return None
        return dict(count=int(len(v)), median=float(np.median(v)), p90=float(np.percentile(v, 90)), p99=float(np.percentile(v, 99)),
                    max=int(v.max()), frac_le2=float((v <= 2).mean()), frac_le4=float((v <= 4).mean()), frac_le8=float((v <= 8).mean()))
    out = dict(source=sys.argv[1], bonds=len(rows), chi_max=chi_max,
               criteria="kept sigma 1e-10 sigma_1, ||M P~ - M P||_F <= 1e-9 ||M||_F, same kept dimension",
               subspace_iterations={str(p): summ(f"sub{p}") for p in PS},
               krylov_blocks={str(b): summ(f"kry{b}") for b, _ in KR},
               gap_sigma_k_over_k1=summ_ratio(rows), rows=rows)
    json.dump(out, open(sys.argv[2], "w"), indent=0)
    print(json.dumps({k: v for k, v in out.items() if k != "rows"}, indent=1))

def summ_ratio(rows):
    v = np.array([r["s_k"] / max(r["s_k1"], 1e-300) for r in rows if r["s_k1"] > 0])
    return dict(median=float(np.median(v)), p10=float(np.percentile(v, 10)), min=float(v.min())) if len(v) else None

PS = [40, 48, 64, 96]
KR = [(32, 4), (48, 3), (64, 3)]
if __name__ == "__main__":
    main()
