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


@pytest.mark.parametrize("n", [2, 3, 4, 16, 33, 64, 128])
def test_eigensolver_against_lapack(eng, n):
    rng = np.random.default_rng(n)
    m = 2 * n
    A = rng.standard_normal((m, n)) * np.logspace(0, -3, n)[None, :]
    A = A @ np.linalg.qr(rng.standard_normal((n, n)))[0]
    G = A.T @ A
    lam, E, sweeps = eng.selftest_eig(G)
    ref = np.linalg.eigvalsh(G)[::-1]
    assert sweeps < 40
    assert np.allclose(lam, ref, rtol=1e-10, atol=1e-14 * ref[0])
    assert np.all(np.diff(lam) <= 0)
    # orthonormal eigenvectors, residual at fp64 level relative to ||G||
    assert np.abs(E.T @ E - np.eye(n)).max() < 1e-13
    assert np.abs(G @ E - E * lam).max() < 1e-13 * ref[0]


def test_eigensolver_rank_deficient(eng):
    rng = np.random.default_rng(7)
    A = rng.standard_normal((8, 128))          # rank 8 Gram of size 128 (first bond of a sweep)
    G = A.T @ A
    lam, E, sweeps = eng.selftest_eig(G)
    ref = np.linalg.eigvalsh(G)[::-1]
    assert sweeps < 40
    assert np.allclose(lam[:8], ref[:8], rtol=1e-11)
    assert np.all(lam[8:] < 1e-12 * ref[0])
    assert np.abs(E[:, :8].T @ E[:, :8] - np.eye(8)).max() < 1e-13
