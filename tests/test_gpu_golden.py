"""HIP engine (through the C ABI and through the fitMPS mirror) against the committed golden
fixtures, and against the C oracle at BASELINE.json's full size.

Tolerances (fp64): per-bond loss 1e-7 and ||grad|| 1e-6 relative, kept singular values 1e-7 of the
largest (differences accumulate along the 2(T-1) bonds of a sweep: every bond is conditioned on all
previous truncations, and with KLD ~ 50 the overlaps are ~1e-11 so 1/yhat amplifies rounding), bond
dimensions / predictions / accuracy exact, per-sweep train KLD 1e-6 relative (north_star's loss-curve
tolerance), final overlaps 1e-6 of the largest.
"""
import glob
import os

import numpy as np
import pytest

import mpstime_jl_amd as mt
from tests.helpers import teacher_forced_segment, teacher_forced_sweep
from oracle import ref_numpy as R
from tests.helpers import load_engine
from tests.test_oracle import load_golden

pytestmark = pytest.mark.gpu
GOLDEN = [p for p in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
          if not os.path.basename(p).startswith(("ref_", "juliaref_", "complex_", "impute_"))]


FREE_RUNNING = [p for p in GOLDEN if "config1" not in p]


def test_config1_teacher_forced_bond_by_bond(engine_cls):
    """BASELINE.json configs[0] (N=200, T=50, chi=4, d=2, two-class trendy sine).  With a train KLD of
    ~50 the overlaps are ~1e-11 and single series with yhat near zero dominate the gradient through
    1/yhat (no clamping in the reference), so a free-running comparison amplifies rounding
    by many orders of magnitude at some bonds.  Here every bond update starts from the oracle's
    state (set_mps + build_caches), which isolates the parity of one update from that conditioning."""
    g, ds, W0, opts = load_golden([p for p in GOLDEN if "config1" in p][0])
    T = ds.phi.shape[1]
    eng = engine_cls(0)
    try:
        load_engine(eng, ds, W0, opts)
        W = [t.copy() for t in W0]
        LE, RE = R.construct_caches(W, ds.phi, True)
        k = 0
        worst = [0.0, 0.0, 0.0, 0.0]
        for going_left, order in ((True, range(T - 2, -1, -1)), (False, range(0, T - 1))):
            if not going_left:
                LE, RE = R.construct_caches(W, ds.phi, False)
            for lid in order:
                eng.set_mps(W)
                eng.build_caches()
                tr = eng.bond_step(lid, going_left)
                tro = {}
                R.bond_step(W, LE, RE, lid, ds, opts, going_left, tro)
                assert tr["chi"] == tro["chi"] == int(g["bond_chi"][k])
                worst[0] = max(worst[0], abs(tr["loss"] - tro["loss"]) / max(1.0, abs(tro["loss"])))
                worst[1] = max(worst[1], abs(tr["grad_norm"] - tro["grad_norm"]) / tro["grad_norm"])
                worst[2] = max(worst[2], np.abs(tr["S"][:tr["chi"]] - tro["S"]).max() / tro["S"][0])
                yo, yg = R.contract_mps(W, ds.phi), R.contract_mps(eng.get_mps(), ds.phi)
                worst[3] = max(worst[3], np.abs(yo - yg).max() / np.abs(yo).max())
                k += 1
        assert worst[0] < 1e-11 and worst[1] < 1e-8 and worst[2] < 1e-9 and worst[3] < 1e-8, worst
    finally:
        eng.close()


@pytest.mark.parametrize("path", FREE_RUNNING, ids=[os.path.basename(p)[:-4] for p in FREE_RUNNING])
def test_engine_reproduces_golden_bond_by_bond(engine_cls, path):
    g, ds, W0, opts = load_golden(path)
    T = ds.phi.shape[1]
    eng = engine_cls(0)
    try:
        load_engine(eng, ds, W0, opts)
        eng.build_caches()
        curve = [eng.eval(0)]
        k = 0
        for _ in range(opts.nsweeps):
            for going_left, order in ((True, range(T - 2, -1, -1)), (False, range(0, T - 1))):
                for lid in order:
                    tr = eng.bond_step(lid, going_left)
                    assert int(g["bond_lid"][k]) == lid and bool(g["bond_left"][k]) == going_left
                    assert tr["chi"] == int(g["bond_chi"][k])
                    assert abs(tr["loss"] - g["bond_loss"][k]) <= 1e-7 * max(1.0, abs(g["bond_loss"][k]))
                    assert abs(tr["grad_norm"] - g["bond_grad_norm"][k]) <= 1e-6 * g["bond_grad_norm"][k]
                    So = g["bond_S"][k, :tr["chi"]]
                    assert np.abs(tr["S"][:tr["chi"]] - So).max() <= 1e-7 * So[0]
                    k += 1
            curve.append(eng.eval(0))
        eng.normalize()
        curve.append(eng.eval(0))
        for i, (mse, kld, acc, _) in enumerate(curve):
            assert abs(kld - g["train_KL_div"][i]) <= 1e-6 * max(1.0, abs(g["train_KL_div"][i]))
            assert abs(mse - g["train_loss"][i]) <= 1e-6
            assert acc == g["train_acc"][i]
        W = eng.get_mps()
        assert np.abs(R.contract_mps(W, ds.phi) - g["overlaps"]).max() <= 1e-6 * np.abs(g["overlaps"]).max()
        assert np.array_equal(R.classify(W, ds.phi), g["pred"])
        chi, ls = eng.get_chi()
        assert ls == T - 1 and np.array_equal(chi, g["final_chi"])
    finally:
        eng.close()


def test_fit_encoded_training_information(capsys):
    """The drop-in seam fitMPS(W, train_states, test_states, opts): keys, lengths and values of
    training_information (RealRealHighDimension.jl:636-655,664,862)."""
    g, ds, W0, opts = load_golden([p for p in GOLDEN if "kld_tsgo_c2" in p][0])
    ets = mt.EncodedTimeSeriesSet(ds.phi, ds.label_index + 1, ds.label_index, np.zeros((len(ds.label_index), 0)),
                                  ds.class_distribution)
    mo = mt.MPSOptions(nsweeps=opts.nsweeps, chi_max=opts.chi_max, eta=opts.eta, d=ds.phi.shape[2], verbosity=-1)
    trained, info, _ = mt.fit_encoded(W0, ets, ets, mo)
    assert set(info) == {"train_loss", "train_acc", "test_loss", "test_acc", "time_taken", "train_KL_div",
                         "test_KL_div", "test_conf"}
    n = opts.nsweeps + 2
    assert all(len(v) == n for v in info.values())
    assert info["time_taken"][0] == 0.0 and np.isnan(info["time_taken"][-1])
    assert np.allclose(info["train_KL_div"], g["train_KL_div"], rtol=1e-6)
    assert np.allclose(info["test_KL_div"], info["train_KL_div"], rtol=1e-12)     # same set supplied as test data
    assert info["test_conf"][-1].sum() == len(ds.label_index)
    assert isinstance(trained, mt.TrainedMPS) and trained.opts == mo
    # classify(mps, states) == classify(mps, X)[sortperm(y)] is exercised in test_gpu_api.py
    assert np.array_equal(mt.classify(trained, ets) - 1, g["pred"])


def _config3(N=4096, T=100, d=4):
    import bench
    full = bench.make_inputs(N, T, d)
    W0 = mt.generate_startingMPS(4, T, d, 2, 1234)
    return full, W0


@pytest.mark.parametrize("chi", [16, 32])
def test_full_size_config3_against_c_oracle(engine_cls, chi):
    """BASELINE.json configs[1] and configs[2] (N=4096, T=100, chi=16 / 32, d=4): every bond update of the first full
    sweep against the C restatement, each starting from the oracle's state (set_mps + build_caches).

    Free-running comparison is meaningless at this size: oracle/sensitivity_study.py shows the oracle
    diverging from ITSELF by O(1) in per-bond loss within one sweep after a 1e-13 perturbation (the
    first sweep starts from overlaps yhat ~ 0 and weights series by 1/yhat).  From a common
    state one update is well conditioned, and that is what is compared; afterwards size-independent
    properties are checked on the engine's own free-running sweep."""
    from oracle.c_oracle import COracle
    full, W0 = _config3()
    T = 100
    co = COracle(W0, full.phi, full.label_index, full.class_distribution, chi, eta=0.01, rebuild_caches=False)
    co.build_caches()
    eng = engine_cls(0)
    sub = slice(0, 4096, 32)
    try:
        eng.set_options(chi_max=chi, eta=0.01)
        eng.set_dataset(0, full.phi, full.label_index, 2)
        worst, chi_flips = teacher_forced_sweep(eng, co, full.phi, T, sub=sub)
        assert worst["loss"] < 1e-10 and worst["grad"] < 1e-8 and worst["S"] < 1e-9 and worst["overlap"] < 1e-8, worst
        assert chi_flips <= 2
        # free-running sweep of the engine itself: properties
        eng.set_mps(W0)
        eng.build_caches()
        kld0 = eng.eval(0)[1]
        eng.sweep()
        Wg = eng.get_mps()
        assert abs(R.mps_norm(Wg) - 1.0) < 1e-10
        for t in Wg[:-1]:
            m = t.reshape(-1, t.shape[2])
            assert np.abs(m.T @ m - np.eye(m.shape[1])).max() < 1e-10      # left-orthonormal after the forward half-sweep
        mse, kld, acc, conf = eng.eval(0)
        yall = R.contract_mps(Wg, full.phi)
        assert abs(kld - np.mean(-np.log(yall[np.arange(4096), full.label_index] ** 2))) < 1e-9 * max(1, abs(kld))
        assert conf.sum() == 4096 and acc == np.mean(np.argmax(np.abs(yall), 1) == full.label_index)
        # it trains: same regime as the oracle (chi=32: KLD about -22 ... -25 and accuracy 0.83 ... 0.97 after one sweep depending
        # on the last bits of the spectra - a free-running fit is chaotic; chi=16 is slower)
        assert kld < kld0 - 10 and acc > (0.75 if chi == 32 else 0.6)
        if chi == 32:
            # the first sweep is the chaotic one (-22.5 ... -24.9 for the same data, depending on the start and on the last bits
            # of the eigensolver); from then on all trajectories descend in parallel, about 0.5-0.9 per sweep.  After five sweeps:
            # three starting seeds x both eigensolver launch chains gave KLD -30.19 ... -30.97, accuracy 0.981 ... 0.992
            # (lab/plateau.py); a build that does not train like the others falls out of this band
            for _ in range(4):
                eng.sweep()
            _, kld5, acc5, _ = eng.eval(0)
            assert -32.5 < kld5 < -29.7 and acc5 > 0.975, (kld5, acc5)
    finally:
        eng.close()


@pytest.mark.parametrize("kw", [dict(update_iters=2), dict(loss="MSE", bbopt="GD", eta=0.05),
                                dict(update_iters=2, loss="MSE", rescale=(True, True))],
                         ids=["iters2", "mse_gd", "mse_iters2_rescale_before"])
def test_full_size_option_paths_against_c_oracle(engine_cls, kw):
    """update_iters > 1, the MSE loss and rescale[1] at BASELINE's full size (N=4096, T=100, chi=32, d=4): 8 bonds where
    the sweep starts and 10 bulk bonds of the forward half-sweep (all at chi_max), each update of engine and C oracle
    from a common state."""
    from oracle.c_oracle import COracle
    full, W0 = _config3()
    T, chi = 100, 32
    eta = kw.get("eta", 0.01)
    okw = dict(eta=eta, update_iters=kw.get("update_iters", 1), loss=kw.get("loss", "KLD"), bbopt=kw.get("bbopt", "TSGO"),
               rescale=kw.get("rescale", (False, True)))
    mk = lambda W: COracle(W, full.phi, full.label_index, full.class_distribution, chi, rebuild_caches=False, **okw)
    eng = engine_cls(0)
    sub = slice(0, 4096, 32)
    try:
        eng.set_options(chi_max=chi, **okw)
        eng.set_dataset(0, full.phi, full.label_index, 2)
        for first, count in ((0, 8), (140, 10)):
            worst, flips = teacher_forced_segment(eng, mk, W0, full.phi, T, first, count, sub=sub)
            assert worst["loss"] < 1e-10 and worst["grad"] < 1e-8 and worst["S"] < 1e-9 and worst["overlap"] < 1e-8, (first, worst)
            assert flips <= 1
    finally:
        eng.close()


def test_sliced_pair_with_multi_stage_shares_against_c_oracle(engine_cls):
    """N = 8192 (the largest size at which the fused chain still picks k_yhat_s + k_grad_s): every gradient block has 8
    shares of 512 series = two stages each, and one class holds 3 series fewer than the other so that shares end inside
    a stage and inside a 16-series tile.  8 bonds where the sweep starts and 8 bulk bonds, engine and C oracle from a
    common state."""
    from oracle.c_oracle import COracle
    import bench
    N, T, chi, d = 8192, 40, 32, 4
    full = bench.make_inputs(N, T, d)
    keep = np.ones(N, dtype=bool)
    keep[[5, 17, 40]] = False                                    # 3 series fewer in the first class
    full = mt.EncodedTimeSeriesSet(full.phi[keep], full.labels[keep], full.label_index[keep], full.original_data[keep],
                                   np.bincount(full.label_index[keep]))
    W0 = mt.generate_startingMPS(4, T, d, 2, 77)
    mk = lambda W: COracle(W, full.phi, full.label_index, full.class_distribution, chi, eta=0.01, rebuild_caches=False)
    eng = engine_cls(0)
    sub = slice(0, N - 3, 64)
    try:
        eng.set_options(chi_max=chi, eta=0.01)
        eng.set_dataset(0, full.phi, full.label_index, 2)
        eng.set_mps(W0)
        info = eng.info()
        assert info["sliced_bond_gemms"] and info["grad_shares"] == 8
        for first, count in ((0, 8), (50, 8)):
            worst, flips = teacher_forced_segment(eng, mk, W0, full.phi, T, first, count, sub=sub)
            assert worst["loss"] < 1e-10 and worst["grad"] < 1e-8 and worst["S"] < 1e-9 and worst["overlap"] < 1e-8, (first, worst)
            assert flips <= 1
    finally:
        eng.close()


def test_config4_n32768_on_one_gpu(engine_cls):
    """BASELINE.json configs[3] (N=32768, T=100, chi=32, d=4) on ONE GPU (it fits: 2 x 839 MB of environments): the first
    and the last 20 bonds of the first sweep against the C oracle from a common state, then the size-independent
    properties of a free-running sweep."""
    from oracle.c_oracle import COracle
    N, T, chi = 32768, 100, 32
    full, W0 = _config3(N=N)
    mk = lambda W: COracle(W, full.phi, full.label_index, full.class_distribution, chi, eta=0.01, rebuild_caches=False)
    eng = engine_cls(0)
    sub = slice(0, N, 256)
    try:
        eng.set_options(chi_max=chi, eta=0.01)
        eng.set_dataset(0, full.phi, full.label_index, 2)
        for first in (0, 2 * (T - 1) - 20):
            worst, flips = teacher_forced_segment(eng, mk, W0, full.phi, T, first, 20, sub=sub)
            assert worst["loss"] < 1e-10 and worst["grad"] < 1e-8 and worst["S"] < 1e-9 and worst["overlap"] < 1e-8, (first, worst)
            assert flips <= 1
        eng.set_mps(W0)
        eng.build_caches()
        kld0 = eng.eval(0)[1]
        st = eng.sweep()
        assert st["eig_fallbacks"] == 0
        Wg = eng.get_mps()
        assert abs(R.mps_norm(Wg) - 1.0) < 1e-10
        for t in Wg[:-1]:
            m = t.reshape(-1, t.shape[2])
            assert np.abs(m.T @ m - np.eye(m.shape[1])).max() < 1e-10
        mse, kld, acc, conf = eng.eval(0)
        yall = R.contract_mps(Wg, full.phi)
        assert abs(kld - np.mean(-np.log(yall[np.arange(N), full.label_index] ** 2))) < 1e-9 * max(1, abs(kld))
        assert conf.sum() == N and acc == np.mean(np.argmax(np.abs(yall), 1) == full.label_index)
        assert kld < kld0 - 10 and acc > 0.8           # observed: KLD 117.9 -> -18.8, accuracy 0.878 after one sweep
    finally:
        eng.close()


def test_full_size_sweeps_are_deterministic(engine_cls):
    """Fixed-order reductions everywhere: two runs of two sweeps give bit-identical tensors."""
    full, W0 = _config3(N=1024)
    outs = []
    for _ in range(2):
        eng = engine_cls(0)
        eng.set_options(chi_max=32, eta=0.01)
        eng.set_dataset(0, full.phi, full.label_index, 2)
        eng.set_mps(W0)
        eng.build_caches()
        eng.sweep()
        eng.sweep()
        outs.append(eng.get_mps())
        eng.close()
    assert all(np.array_equal(a, b) for a, b in zip(*outs))
