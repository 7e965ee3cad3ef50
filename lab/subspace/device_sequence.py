"""The exact launch sequence planned for the device (randomised SVD with Cholesky-QR between half-steps, Rayleigh-Ritz read
off the Gram matrix of the last Z = M^H Q), in NumPy, on dumped bond matrices: pass rate for a FIXED number of half-steps, and
calibration of the on-device acceptance test (residuals of the kept Ritz pairs + Frobenius certificate) against the true errors.
usage: python device_sequence.py dump.npz chi_max pextra [stride]"""
import sys, json
import numpy as np
from feasibility import trunc
from feasibility_mside import cholqr

def omega(m, p):
    # deterministic +-1 pattern (what the kernel generates by hashing (row, column))
    i = np.arange(m, dtype=np.uint64)[:, None]; j = np.arange(p, dtype=np.uint64)[None, :]
    h = (i * np.uint64(0x9E3779B97F4A7C15) + j * np.uint64(0xC2B2AE3D27D4EB4F)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    h ^= h >> np.uint64(29); h = (h * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF); h ^= h >> np.uint64(32)
    return np.where((h & np.uint64(1)) == 1, 1.0, -1.0)

def run(M, k_true, chi_max, p, pairs):
    """pairs = number of (M, M^H) rounds after the start block; returns dict of errors and acceptance quantities."""
    m, n = M.shape
    X = cholqr(M.conj().T @ omega(m, p), 1)
    for _ in range(pairs - 1):
        Q = cholqr(M @ X, 1)
        X = cholqr(M.conj().T @ Q, 1)
    Q = cholqr(M @ X, 1)
    Z = M.conj().T @ Q
    S = Z.conj().T @ Z
    S = 0.5 * (S + S.conj().T)
    th, W = np.linalg.eigh(S)
    th, W = np.maximum(th[::-1], 0), W[:, ::-1]
    return th, Z, W

def main():
    z = np.load(sys.argv[1]); meta = z["meta"]
    chi_max, pe = int(sys.argv[2]), int(sys.argv[3])
    stride = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    out = []
    for i in range(0, len(meta), stride):
        M = z[f"M{i}"]; m, n = M.shape
        U, sig, Vh = np.linalg.svd(M, full_matrices=False); V = Vh.conj().T
        lam = sig ** 2; total = lam.sum(); k = trunc(lam, total, chi_max, 1e-10)
        p = min(chi_max, min(m, n)) + pe
        if n < 2 * p:
            continue
        G = M.conj().T @ M
        row = dict(i=i, n=n, m=m, k=k, s_k=sig[k - 1] / sig[0], s_k1=sig[k] / sig[0] if k < len(sig) else 0)
        for pairs in (1, 2, 3):
            th, Z, W = run(M, k, chi_max, p, pairs)
            kk = trunc(th[:min(len(th), chi_max)], total, chi_max, 1e-10) if True else 0
            # truncation with the tail mass known from the trace
            P = th[:chi_max]; err = total - P.sum(); nk = len(P)
            while nk > 1 and err + P[nk - 1] <= 1e-10 * total:
                err += P[nk - 1]; nk -= 1
            Vk = (Z @ W[:, :nk]) / np.sqrt(np.maximum(th[:nk], 1e-300))
            # Loewdin polish as the device does (two rounds)
            for _ in range(2):
                D = Vk.conj().T @ Vk - np.eye(nk); Vk = Vk - 0.5 * Vk @ D
            R = G @ Vk - Vk * th[:nk]
            res = np.linalg.norm(R, axis=0) / lam[0]
            rho = np.sqrt(max(np.sum(G.real ** 2 + G.imag ** 2) - np.sum(th ** 2), 0.0))        # Frobenius bound on the unseen eigenvalues
            errS = np.abs(np.sqrt(th[:min(nk, k)]) - sig[:min(nk, k)]).max() / sig[0]
            errP = np.linalg.norm(M @ Vk @ Vk.conj().T - (M @ V[:, :k]) @ V[:, :k].conj().T) / np.sqrt(total) if nk == k else np.inf
            est = np.sqrt(np.sum((np.linalg.norm(R, axis=0) ** 2) / np.maximum(th[:nk], 1e-300))) / np.sqrt(total)
            row[f"h{pairs}"] = dict(chi_ok=bool(nk == k), errS=float(errS), errP=float(errP), res_max=float(res.max()), est=float(est),
                                    cert=bool(rho < th[nk - 1]), rho_over_thk=float(rho / max(th[nk - 1], 1e-300)))
        out.append(row)
        print({a: (b if not isinstance(b, dict) else {x: (f"{y:.2e}" if isinstance(y, float) else y) for x, y in b.items()}) for a, b in row.items()}, flush=True)
    for pairs in (1, 2, 3):
        ok = [r[f"h{pairs}"]["chi_ok"] and r[f"h{pairs}"]["errS"] <= 1e-10 and r[f"h{pairs}"]["errP"] <= 1e-9 for r in out]
        print(f"pairs={pairs}: pass {sum(ok)}/{len(ok)}")
    json.dump(out, open("/tmp/devseq.json", "w"))

main()
