"""Element-typed sweep on LONG chains (BASELINE configs[4] has T = 200): products of T site contractions leave fp32's range
after ~80 Fourier sites, so the environments are stored scaled by powers of two (csrc/mpst_typed.hip: k_tenv) and the MSE
gradient weights are staged relative to the bond's largest exponent.  Checked against oracle/ref_complex.py in DOUBLE precision
on the fp32-rounded inputs (legacy engine semantics, src/legacy_itensor/loss_functions.jl:433-640,
RealRealLegacyITensor.jl:2-47), with the fp32 tolerances of tests/test_gpu_typed.py:
    loss 2e-5, ||grad|| 1e-4, singular values 2e-5 sigma_1, two-site tensor 1e-4 max|.|, KLD of the whole model 2e-5.
"""
import numpy as np
import pytest

from oracle import ref_complex as RC
from tests.test_gpu_typed import TOL, caches_around, problem, two_site

pytestmark = pytest.mark.gpu
DT = {"float32": np.float32, "complex64": np.complex64}


def compare_bond(eng, W64, ds64, dtype, ls, lid, opts, worst):
    """One bond update from the common state W64 (label on site ls), engine against oracle."""
    eng.set_mps([t.astype(dtype) for t in W64], label_site=ls)
    eng.build_caches()
    LE, RE = caches_around(W64, ds64.phi, ls)
    Wo = [t.copy() for t in W64]
    tr = {}
    RC.bond_step(Wo, LE, RE, lid, ds64, opts, ls == lid + 1, tr)
    got = eng.bond_step(lid, ls == lid + 1)
    assert np.isfinite(got["loss"]) and got["grad_norm"] > 0 and tr["grad_norm"] > 0, (got["loss"], got["grad_norm"], tr["grad_norm"])
    worst["loss"] = max(worst["loss"], abs(got["loss"] - tr["loss"]) / max(1.0, abs(tr["loss"])))
    worst["grad"] = max(worst["grad"], abs(got["grad_norm"] - tr["grad_norm"]) / tr["grad_norm"])
    assert got["chi"] == tr["chi"], (got["chi"], tr["chi"])
    worst["S"] = max(worst["S"], np.abs(got["S"][:tr["chi"]] - tr["S"]).max() / tr["S"][0])
    Wg = eng.get_mps()
    a, b = two_site(Wg[lid], Wg[lid + 1]), two_site(Wo[lid], Wo[lid + 1])
    worst["bond"] = max(worst["bond"], np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize("loss,bbopt", [("KLD", "TSGO"), ("MSE", "TSGO"), ("MSE", "GD")])
@pytest.mark.parametrize("T", [100, 200])
@pytest.mark.parametrize("dtype", ["float32", "complex64"])
def test_long_chain_bond_updates(dtype, T, loss, bbopt):
    """Whole-model KLD, then the bond updates at T-2 (first of the sweep) and at T/2 (reached by free-running the engine; the
    oracle starts from the engine's state there), where both environments carry exponents of hundreds of bits."""
    import mpstime_jl_amd as mt
    dt = DT[dtype]
    wide = np.complex128 if dtype == "complex64" else np.float64
    ds64, W64 = problem(64, T, 4, 4, 2, 1, dt)
    opts = RC.SweepOptions(chi_max=8, eta=0.05 if bbopt == "TSGO" else 0.01, loss_grad=loss, bbopt=bbopt)
    eng = mt.SweepEngine(0)
    try:
        eng.set_options(chi_max=8, eta=opts.eta, loss=loss, bbopt=bbopt)
        eng.set_dataset(0, ds64.phi.astype(dt), ds64.label_index, 2)
        eng.set_mps([t.astype(dt) for t in W64])
        assert eng.info()["typed_kernels"]
        _, kld, _, _ = eng.eval(0)
        ko = RC.mse_loss_acc(W64, ds64)[1]
        assert np.isfinite(kld) and abs(kld - ko) < 2e-5 * max(1.0, abs(ko)), (kld, ko)
        worst = dict(loss=0.0, grad=0.0, S=0.0, bond=0.0)
        compare_bond(eng, W64, ds64, dt, T - 1, T - 2, opts, worst)
        # free-run the backward half-sweep down to the middle of the chain
        eng.set_mps([t.astype(dt) for t in W64])
        eng.build_caches()
        for lid in range(T - 2, T // 2, -1):
            eng.bond_step(lid, True)
        chi, ls = eng.get_chi()
        assert ls == T // 2 + 1
        Wm = [np.asarray(t).astype(wide) for t in eng.get_mps()]
        assert all(np.isfinite(t).all() for t in Wm)
        compare_bond(eng, Wm, ds64, dt, ls, ls - 1, opts, worst)
    finally:
        eng.close()
    print(dtype, T, loss, bbopt, worst)
    for k, tol in TOL["f32"].items():
        assert worst[k] < tol, (k, worst)


def test_configs4_full_size_free_running_sweeps_complex64():
    """BASELINE configs[4]'s training shape at FULL size (N = 8192, T = 200, chi_max = 64, d = 8, one class, Fourier, complex64):
    two free-running sweeps through mpst_sweep.  No oracle at this size: the size-independent properties - finite, no solver
    fall-backs, bond dimensions within chi_max, the KLD decreases sweep over sweep, left-canonical form with the label on the
    last site, unit norm after normalize! (RealRealHighDimension.jl:852)."""
    import bench
    import mpstime_jl_amd as mt
    N, T, d, chi = 8192, 200, 8, 64
    full = bench.typed_inputs(N, T, d, 1, True)
    W0 = mt.generate_startingMPS(4, T, d, 1, 1234, np.complex64)
    eng = mt.SweepEngine(0)
    try:
        eng.set_options(chi_max=chi, eta=0.01, cutoff=1e-10)
        eng.set_dataset(0, full.phi, full.label_index, 1, dtype=np.complex64)
        eng.set_mps(W0)
        eng.build_caches()
        klds = [eng.eval(0)[1]]
        for _ in range(2):
            st = eng.sweep()
            assert st["eig_fallbacks"] == 0 and st["max_chi"] <= chi, st
            klds.append(eng.eval(0)[1])
        info = eng.info()
        assert info["large_bond"] and info["typed_kernels"] and info["library_eig_fallbacks"] == 0, info
        assert np.isfinite(klds).all() and klds[2] < klds[1] < klds[0], klds
        chis, ls = eng.get_chi()
        assert ls == T - 1 and chis.max() == chi
        eng.normalize()
        W = [np.asarray(t).astype(np.complex128) for t in eng.get_mps()]
        assert all(np.isfinite(t).all() for t in W)
        for j in (0, 1, T // 2, T - 2):
            A = W[j].reshape(-1, W[j].shape[2])          # (a, s) x k: left-orthonormal
            assert np.abs(A.conj().T @ A - np.eye(A.shape[1])).max() < 2e-5, j
        assert abs(RC.mps_norm(W) - 1.0) < 1e-5
    finally:
        eng.close()
    print("configs[4] full size, complex64: KLD", klds)
