"""GPU unit checks of the MFMA tile map and the on-device eigensolver (through the C ABI)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(engine_cls):
    e = engine_cls(0)
    yield e
    e.close()


@pytest.mark.parametrize("K", [4, 8, 36, 128])
def test_mfma_f64_tile_layout(eng, K):
    # exact integer data, asymmetric B: catches a swapped row/col or the f32 row map on f64
    rng = np.random.default_rng(K)
    A = rng.integers(-8, 9, (16, K)).astype(float)
    B = rng.integers(-8, 9, (K, 16)).astype(float)
    B[:, 3] += 100.0
    out = eng.selftest_mfma(A, B)
    assert np.array_equal(out, A @ B)


def _graded_gram(n, seed, decades=3):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((2 * n, n)) * np.logspace(0, -decades, n)[None, :]
    A = A @ np.linalg.qr(rng.standard_normal((n, n)))[0]
    return A.T @ A


@pytest.mark.parametrize("n", [2, 3, 4, 16, 33, 64, 128])
def test_jacobi_eigensolver_against_lapack(eng, n):
    G = _graded_gram(n, n)
    lam, E, sweeps = eng.selftest_eig(G, alg=1)
    ref = np.linalg.eigvalsh(G)[::-1]
    assert 0 < sweeps < 40
    assert np.allclose(lam, ref, rtol=1e-10, atol=1e-14 * ref[0])
    assert np.all(np.diff(lam) <= 0)
    assert np.abs(E.T @ E - np.eye(n)).max() < 1e-13
    assert np.abs(G @ E - E * lam).max() < 1e-13 * ref[0]


@pytest.mark.parametrize("n", [2, 3, 4, 5, 16, 31, 33, 64, 100, 128])
@pytest.mark.parametrize("decades", [1, 4])
def test_tridiagonal_eigensolver_against_lapack(eng, n, decades):
    """Default path: top K = min(n, 32) eigenpairs, verified on the device (info == -1)."""
    G = _graded_gram(n, 100 + n, decades)
    lam, E, info = eng.selftest_eig(G, alg=0)
    K = min(n, 32)
    ref = np.linalg.eigvalsh(G)[::-1]
    assert info == -1, "device-side verification fell back to Jacobi"
    assert np.abs(lam[:K] - ref[:K]).max() < 1e-13 * ref[0]
    Ek = E[:, :K]
    assert np.abs(Ek.T @ Ek - np.eye(K)).max() < 1e-12
    assert np.abs(G @ Ek - Ek * lam[:K]).max() < 1e-12 * ref[0]


def test_tridiagonal_eigensolver_clustered_falls_back(eng):
    """Exactly repeated eigenvalues: twisted factorisation cannot give orthogonal vectors; the
    on-device check must catch it and the Jacobi path must deliver a valid basis."""
    rng = np.random.default_rng(3)
    Q = np.linalg.qr(rng.standard_normal((64, 64)))[0]
    lam_true = np.concatenate([np.full(8, 2.0), np.full(8, 1.0), np.linspace(0.5, 0.01, 48)])
    G = (Q * lam_true) @ Q.T
    G = 0.5 * (G + G.T)
    lam, E, info = eng.selftest_eig(G, alg=0)
    K = 32
    assert np.abs(lam[:K] - lam_true[:K]).max() < 1e-12
    Ek = E[:, :K]
    assert np.abs(Ek.T @ Ek - np.eye(K)).max() < 1e-11
    assert np.abs(G @ Ek - Ek * lam[:K]).max() < 1e-11


def test_eigensolver_rank_deficient(eng):
    rng = np.random.default_rng(7)
    A = rng.standard_normal((8, 128))          # rank 8 Gram of size 128 (first bond of a sweep)
    G = A.T @ A
    ref = np.linalg.eigvalsh(G)[::-1]
    for alg in (0, 1):
        lam, E, info = eng.selftest_eig(G, alg=alg)
        assert np.allclose(lam[:8], ref[:8], rtol=1e-11)
        assert np.all(lam[8:] < 1e-12 * ref[0])
        assert np.abs(E[:, :8].T @ E[:, :8] - np.eye(8)).max() < 1e-12
        assert np.abs(G @ E[:, :8] - E[:, :8] * lam[:8]).max() < 1e-12 * ref[0]


def test_split_eigensolver_chain_stays_correct():
    """The default chain runs the tridiagonalisation inside every eigenvector workgroup (k_eig_trivec); the two separate
    kernels it was merged from are still built and selected by MPST_EIG_SPLIT=1 (read once per process, hence the child):
    the same LAPACK comparisons, the cluster fallback and the rank-deficient case must hold on that chain too."""
    import os
    import subprocess
    import sys
    if os.environ.get("MPST_EIG_SPLIT"):
        pytest.skip("already running the split chain")
    env = dict(os.environ, MPST_EIG_SPLIT="1")
    out = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-m", "gpu", "-x", "-k", "eigensolver and not split"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "failed" not in out.stdout, out.stdout[-500:]
