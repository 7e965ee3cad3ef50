"""Probe (not a test): how the fp64 eigensolvers behave on the real 2n x 2n embedding of a complex Hermitian Gram matrix,
whose eigenvalues all come in exact pairs.  Prints, per size, whether the tridiagonal path survived, the residuals of the
even-indexed eigenvectors and their complex orthogonality."""
import sys
import numpy as np

sys.path.insert(0, ".")
import mpstime_jl_amd as mt


def embedded(nc, rows, seed, decay):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((rows, nc)) + 1j * rng.standard_normal((rows, nc))
    U, s, Vh = np.linalg.svd(A, full_matrices=False)
    s = np.exp(-decay * np.arange(len(s)))
    A = (U * s) @ Vh
    G = A.conj().T @ A
    M = np.block([[G.real, -G.imag], [G.imag, G.real]])
    return G, 0.5 * (M + M.T)


def main():
    eng = mt.SweepEngine(0)
    for nc, decay in ((8, 0.5), (32, 0.3), (64, 0.3), (64, 0.05), (128, 0.1), (256, 0.1), (512, 0.05)):
        G, M = embedded(nc, 2 * nc, 1, decay)
        for alg in (4,):
            try:
                lam, E, info = eng.selftest_eig(M, alg=alg)
            except Exception as ex:
                print(nc, "failed", ex)
                continue
            n = 2 * nc
            ref = np.linalg.eigvalsh(G)[::-1]
            K = min(n, 128) if n > 128 else min(n, 32)
            K &= ~1
            lam = lam[:K]
            # pairs
            pe = np.abs(lam[0::2] - lam[1::2]).max() / lam[0]
            le = np.abs(lam[0::2] - ref[: K // 2]).max() / ref[0]
            U = E[:, :K]
            res = np.linalg.norm(M @ U - U * lam, axis=0).max() / lam[0]
            Ue = U[:, 0::2]
            Cv = Ue[:nc] + 1j * Ue[nc:]
            D = Cv.conj().T @ Cv - np.eye(Cv.shape[1])
            Dr = U.T @ U - np.eye(K)
            print(f"nc={nc:4d} decay={decay} info={info} K={K} pair_gap={pe:.2e} lam_err={le:.2e} res={res:.2e} "
                  f"real_orth={np.abs(Dr).max():.2e} cplx_orth_even={np.abs(D).max():.2e}", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
