"""Element-typed sweep (csrc/mpst_typed.hip) against the CPU restatement of the reference's complex / reduced-precision
semantics (oracle/ref_complex.py: legacy ITensor engine, src/legacy_itensor/loss_functions.jl:433-640).

Every bond update of a sweep is compared with the oracle from a COMMON state (teacher forcing: free-running trajectories
diverge chaotically, oracle/sensitivity_study.py), in gauge-invariant quantities: loss, ||grad||, the kept singular values,
the kept bond dimension and the updated two-site tensor W[lid] W[rid] (the SVD's phase freedom cancels in the product).

Tolerances (relative, stated per element type; observed values are printed with -s):
    quantity            float64 / complex128      float32 / complex64
    loss                1e-11                     2e-5     (sum of N fp32-accurate overlaps, -log|y|^2 in fp64)
    ||grad||            1e-10                     1e-4
    singular values     1e-9 sigma_1              2e-5 sigma_1   (Gram matrix and eigensolver are fp64 in every type)
    two-site tensor     1e-8 max|.|               1e-4 max|.|
For the fp32 types the oracle runs in DOUBLE precision on the fp32-rounded inputs: what is measured is the arithmetic of the
engine, not the rounding of the inputs.
"""
import numpy as np
import pytest

from oracle import ref_complex as RC
from oracle import ref_numpy as R
from tests.helpers import bond_of

pytestmark = pytest.mark.gpu

TOL = {
    "f64": dict(loss=1e-11, grad=1e-10, S=1e-9, bond=1e-8),
    "f32": dict(loss=2e-5, grad=1e-4, S=2e-5, bond=1e-4),
}
DT = {"float64": np.float64, "float32": np.float32, "complex128": np.complex128, "complex64": np.complex64}


def caches_around(W, phi, ls):
    """LE[0..ls-1], RE[ls+1..T-1] for a label on site ls (construct_caches on both sides of it)."""
    T = len(W)
    N = phi.shape[0]
    LE, RE = [None] * T, [None] * T
    prev = np.ones((N, 1), dtype=W[0].dtype)
    for j in range(ls):
        prev = np.einsum("is,ask,ia->ik", np.conj(phi[:, j, :]), W[j], prev)
        LE[j] = prev
    prev = np.ones((N, 1), dtype=W[0].dtype)
    for j in range(T - 1, ls, -1):
        prev = np.einsum("is,ksb,ib->ik", np.conj(phi[:, j, :]), W[j], prev)
        RE[j] = prev
    return LE, RE


def two_site(Wl, Wr):
    bt, shape4 = R.flatten_bt(Wl, Wr)
    return R.unflatten_bt(bt, shape4)


def run_teacher_forced(eng, ds64, W64, dtype, opts, nbonds=None):
    """ds64 / W64: the problem in double precision (already rounded to the element type's precision)."""
    T = len(W64)
    eng.set_options(chi_max=opts.chi_max, eta=opts.eta, cutoff=opts.cutoff, update_iters=opts.update_iters, loss=opts.loss_grad,
                    bbopt=opts.bbopt, rescale=opts.rescale, train_classes_separately=opts.train_classes_separately)
    C = len(ds64.class_distribution)
    eng.set_dataset(0, ds64.phi.astype(dtype), ds64.label_index, C)
    W = [t.copy() for t in W64]
    worst = dict(loss=0.0, grad=0.0, S=0.0, bond=0.0)
    flips = 0
    nb = 2 * (T - 1) if nbonds is None else nbonds
    for q in range(nb):
        lid, going_left = bond_of(q, T)
        ls = lid + 1 if going_left else lid
        eng.set_mps([t.astype(dtype) for t in W], label_site=ls)
        eng.build_caches()
        LE, RE = caches_around(W, ds64.phi, ls)
        tr = {}
        RC.bond_step(W, LE, RE, lid, ds64, opts, going_left, tr)
        got = eng.bond_step(lid, going_left)
        worst["loss"] = max(worst["loss"], abs(got["loss"] - tr["loss"]) / max(1.0, abs(tr["loss"])))
        worst["grad"] = max(worst["grad"], abs(got["grad_norm"] - tr["grad_norm"]) / tr["grad_norm"])
        nk = min(got["chi"], tr["chi"])
        worst["S"] = max(worst["S"], np.abs(got["S"][:nk] - tr["S"][:nk]).max() / tr["S"][0])
        if got["chi"] != tr["chi"]:
            flips += 1
            continue
        Wg = eng.get_mps()
        a, b = two_site(Wg[lid], Wg[lid + 1]), two_site(W[lid], W[lid + 1])
        worst["bond"] = max(worst["bond"], np.abs(a - b).max() / np.abs(b).max())
    return worst, flips


def problem(N, T, d, chi, C, seed, dtype, balanced=True):
    ds, W = RC.make_problem(N, T, d, chi, C, seed=seed, dtype=dtype, balanced=balanced)
    big = np.complex128 if np.issubdtype(np.dtype(dtype), np.complexfloating) else np.float64
    return RC.cast_problem(ds, W, big)       # rounded to `dtype`, held in double precision


CASES = [
    # N, T, d, chi_init, chi_max, C, loss, bbopt, iters, train_sep, balanced
    (96, 6, 2, 3, 6, 2, "KLD", "TSGO", 1, False, True),
    (130, 5, 3, 4, 8, 3, "KLD", "TSGO", 2, False, False),
    (64, 6, 4, 4, 12, 1, "KLD", "TSGO", 1, False, True),
    (80, 5, 2, 2, 5, 2, "MSE", "GD", 1, False, False),
    (100, 4, 3, 3, 7, 2, "KLD", "GD", 1, True, False),
    (70, 2, 3, 1, 3, 2, "KLD", "TSGO", 1, False, True),          # two-site MPS
]


@pytest.mark.parametrize("dtype", ["complex128", "float64", "complex64", "float32"])
@pytest.mark.parametrize("case", CASES, ids=[f"N{c[0]}T{c[1]}d{c[2]}chi{c[4]}C{c[5]}{c[6]}{c[7]}" for c in CASES])
def test_bond_by_bond_against_the_complex_oracle(dtype, case, monkeypatch):
    import mpstime_jl_amd as mt
    N, T, d, chi0, chimax, C, loss, bbopt, iters, sep, bal = case
    if dtype == "float64":
        monkeypatch.setenv("MPST_TYPED", "1")            # the Float64 specialisation of the typed kernels
    ds, W = problem(N, T, d, chi0, C, 11, DT[dtype], bal)
    opts = RC.SweepOptions(chi_max=chimax, eta=0.05 if bbopt == "TSGO" else 0.01, update_iters=iters, loss_grad=loss, bbopt=bbopt,
                           train_classes_separately=sep)
    eng = mt.SweepEngine(0)
    try:
        worst, flips = run_teacher_forced(eng, ds, W, DT[dtype], opts)
        info = eng.info()
    finally:
        eng.close()
    assert info["typed_kernels"]
    tol = TOL["f32" if dtype in ("float32", "complex64") else "f64"]
    print(dtype, case, worst, flips)
    assert flips <= 1
    for k in tol:
        assert worst[k] < tol[k], (k, worst)


@pytest.mark.parametrize("dtype", ["complex128", "complex64"])
def test_free_running_sweeps_and_evaluation(dtype):
    """Two free-running sweeps (hipGraph path of mpst_sweep) of a complex model, then evaluation, classification and
    normalisation against the oracle on the ENGINE's MPS."""
    import mpstime_jl_amd as mt
    ds, W = problem(128, 8, 3, 3, 2, 5, DT[dtype])
    opts = RC.SweepOptions(chi_max=9, eta=0.05)
    eng = mt.SweepEngine(0)
    try:
        eng.set_options(chi_max=opts.chi_max, eta=opts.eta)
        eng.set_dataset(0, ds.phi.astype(DT[dtype]), ds.label_index, 2)
        eng.set_dataset(1, ds.phi[::3].astype(DT[dtype]), ds.label_index[::3], 2)
        eng.set_mps([t.astype(DT[dtype]) for t in W])
        eng.build_caches()
        Wo = [t.copy() for t in W]
        LE = RE = None
        for _ in range(2):
            eng.sweep()
            LE, RE = RC.sweep(Wo, ds, opts, LE, RE)
        Wg = [t.astype(np.complex128) for t in eng.get_mps()]
        mse, kld, acc, conf = eng.eval(0)
        mo, ko, ao, co = RC.mse_loss_acc(Wg, ds, conf=True)
        tol = 1e-10 if dtype == "complex128" else 2e-5
        assert abs(mse - mo) < tol * max(1, abs(mo)) and abs(kld - ko) < tol * max(1, abs(ko)) and acc == ao and (conf == co).all()
        pred, yh = eng.classify(1, return_overlaps=True)
        yo = RC.contract_mps(Wg, ds.phi[::3])
        assert np.abs(yh - yo).max() < tol * np.abs(yo).max()
        assert (pred == RC.classify(Wg, ds.phi[::3])).all()
        # the free-running trajectory itself: same loss to a loose tolerance after two sweeps (chaotic beyond that)
        mo2, ko2, ao2 = RC.mse_loss_acc(Wo, ds)
        assert abs(kld - ko2) < (1e-6 if dtype == "complex128" else 5e-3) * max(1, abs(ko2)), (kld, ko2)
        eng.normalize()
        Wn = [t.astype(np.complex128) for t in eng.get_mps()]
        assert abs(RC.mps_norm(Wn) - 1.0) < (1e-12 if dtype == "complex128" else 1e-5)
    finally:
        eng.close()


def test_real_encoding_needs_no_complex_type_but_complex_encoding_does():
    import mpstime_jl_amd as mt
    eng = mt.SweepEngine(0)
    try:
        eng.set_options(chi_max=4)
        eng.set_dtype(np.float64)
        X = np.random.default_rng(0).uniform(0, 1, (8, 5))
        with pytest.raises(mt.MPSTError, match="complex valued encoding but the MPS is real"):
            eng.encode_dataset(0, X, np.zeros(8, dtype=np.int32), 1, basis="Fourier", d=4, sigmoid_transform=False, minmax=False,
                               enc_range=(-1, 1))
    finally:
        eng.close()


def test_device_fourier_encoding_becomes_a_complex_training_set():
    import mpstime_jl_amd as mt
    rng = np.random.default_rng(3)
    X = rng.uniform(0, 1, (40, 6))
    y = (np.arange(40) >= 20).astype(np.int32)
    for dt in (np.complex128, np.complex64):
        eng = mt.SweepEngine(0)
        try:
            eng.set_options(chi_max=6, eta=0.05)
            if dt == np.complex64:
                eng.set_dtype(dt)
            eng.encode_dataset(0, X, y, 2, basis="Fourier", d=4, sigmoid_transform=False, minmax=False, enc_range=(-1, 1))
            phi = eng.get_encoded(0)
            ref = R.fourier_encode(2 * X - 1, 4)
            assert phi.dtype == dt and np.abs(phi - ref).max() < (1e-14 if dt == np.complex128 else 1e-6)
            W = R.random_mps(6, 4, 3, 2, np.random.default_rng(4), dtype=np.complex128)
            eng.set_mps(W)
            eng.build_caches()
            eng.sweep()
            mse, kld, acc, _ = eng.eval(0)
            assert np.isfinite(kld)
        finally:
            eng.close()


@pytest.mark.parametrize("dtype,chi", [("complex128", 20), ("complex64", 20), ("float32", 40)])
def test_large_bond_tensors_go_through_the_blocked_solver(dtype, chi):
    """(2) d chi_max > 128: the Gram matrix (complex: its 160 x 160 embedding) is decomposed by the blocked solver of
    mpst_eig_blocked.hip in pair mode; first half-sweep of an 8-site chain, teacher forced."""
    import mpstime_jl_amd as mt
    ds, W = problem(192, 8, 4, chi, 2, 21, DT[dtype])
    opts = RC.SweepOptions(chi_max=chi, eta=0.05)
    eng = mt.SweepEngine(0)
    try:
        worst, flips = run_teacher_forced(eng, ds, W, DT[dtype], opts, nbonds=9)
        info = eng.info()
    finally:
        eng.close()
    assert info["large_bond"] and info["typed_kernels"] and info["library_eig_fallbacks"] == 0
    tol = TOL["f32" if dtype in ("float32", "complex64") else "f64"]
    print(dtype, worst, flips)
    assert flips <= 1
    for k in tol:
        assert worst[k] < tol[k], (k, worst)


def test_fitMPS_trains_a_fourier_model():
    """fitMPS(X, y; encoding = :Fourier) end to end in the Python mirror: complex128 (the reference's default dtype for a complex
    encoding, options.jl:117) and complex64, against the oracle's sweeps on the same encoded data."""
    import mpstime_jl_amd as mt
    rng = np.random.default_rng(0)
    X, y = R.trendy_sine_dataset(120, 12, rng)
    Xte, yte = R.trendy_sine_dataset(40, 12, rng)
    for dname, tol in (("ComplexF64", 1e-7), ("ComplexF32", 2e-3)):
        opts = mt.MPSOptions(encoding="Fourier", d=4, chi_max=10, nsweeps=2, eta=0.05, verbosity=-1, dtype=dname, chi_init=3)
        trained, info, test_states = mt.fitMPS(X, y, Xte, yte, opts)
        assert len(info["train_KL_div"]) == 4 and np.isfinite(info["train_KL_div"]).all()
        assert trained.mps[0].dtype == (np.complex128 if dname == "ComplexF64" else np.complex64)
        # the oracle from the same starting MPS and encoded data
        W0 = mt.generate_startingMPS(3, 12, 4, 2, opts.init_rng, np.complex128 if dname == "ComplexF64" else np.complex64)
        ds = RC.EncodedSet(trained.train_data.phi.astype(W0[0].dtype), trained.train_data.label_index,
                           np.bincount(trained.train_data.label_index).astype(np.int64))
        Wo = [t.astype(np.complex128) for t in W0]
        ds64 = RC.EncodedSet(ds.phi.astype(np.complex128), ds.label_index, ds.class_distribution)
        so = RC.SweepOptions(chi_max=10, eta=0.05)
        LE = RE = None
        klds = [RC.mse_loss_acc(Wo, ds64)[1]]
        for _ in range(2):
            LE, RE = RC.sweep(Wo, ds64, so, LE, RE)
            klds.append(RC.mse_loss_acc(Wo, ds64)[1])
        got = np.array(info["train_KL_div"][:3])
        assert np.abs(got - np.array(klds)).max() < tol * np.abs(klds).max(), (got, klds)
        # predictions are those of the trained model on the encoded test states
        pred = mt.classify(trained, test_states)
        assert pred.shape == (40,) and set(np.unique(pred)) <= {1, 2}
        assert np.mean(pred == np.sort(yte)) == info["test_acc"][-1]


def test_config5_bond_shape_complex64():
    """BASELINE configs[4]'s training shape (chi = 64, d = 8, one class, Fourier) on a short chain: 1024 x 1024 embedding through
    the blocked solver, three bonds teacher forced against the double-precision oracle."""
    import mpstime_jl_amd as mt
    ds, W = problem(1024, 6, 8, 64, 1, 3, np.complex64)
    opts = RC.SweepOptions(chi_max=64, eta=0.05)
    eng = mt.SweepEngine(0)
    try:
        worst, flips = run_teacher_forced(eng, ds, W, np.complex64, opts, nbonds=3)
        info = eng.info()
    finally:
        eng.close()
    print(worst, flips, info)
    assert info["large_bond"] and info["library_eig_fallbacks"] == 0
    for k in TOL["f32"]:
        assert worst[k] < TOL["f32"][k], (k, worst)


def test_rescale_before_and_track_cost_in_the_typed_chain():
    """rescale = (true, true) (normalize!(BT) before the update, legacy loss_functions.jl:194-196) and opts.track_cost (the losses
    before every optimiser step and at the updated, rescaled bond tensor) through the element-typed kernels, complex128."""
    import mpstime_jl_amd as mt
    ds, W = problem(90, 6, 3, 3, 2, 17, np.complex128, balanced=False)
    opts = RC.SweepOptions(chi_max=7, eta=0.05, update_iters=2, rescale=(True, True))
    eng = mt.SweepEngine(0)
    try:
        worst, flips = run_teacher_forced(eng, ds, W, np.complex128, opts)
        assert flips == 0 and all(worst[k] < TOL["f64"][k] for k in TOL["f64"]), worst
        # one free-running sweep with the trace: every recorded loss is the oracle's at that point
        eng.set_options(chi_max=7, eta=0.05, update_iters=2, rescale=(True, True), track_cost=True)
        eng.set_mps(W)
        eng.build_caches()
        eng.sweep()
        trace = eng.loss_trace()
        Wo = [t.copy() for t in W]
        LE, RE = RC.construct_caches(Wo, ds.phi, True)
        T = len(W)
        for q in range(T - 1):                                   # the backward half-sweep
            lid = T - 2 - q
            bt, shape4 = RC.flatten_bt(Wo[lid], Wo[lid + 1])
            bt5 = RC.unflatten_bt(bt, shape4)
            bt5 = bt5 / np.linalg.norm(bt5)
            LEp = LE[lid - 1] if lid > 0 else None
            REp = RE[lid + 2] if lid + 1 < T - 1 else None
            l0, g0 = RC.loss_grad(bt5, LEp, REp, ds, lid, lid + 1)
            b1 = bt5 - opts.eta * g0 / np.linalg.norm(g0)
            l1, g1 = RC.loss_grad(b1, LEp, REp, ds, lid, lid + 1)
            b2 = b1 - opts.eta * g1 / np.linalg.norm(g1)
            b2 = b2 / np.linalg.norm(b2)
            l2, _ = RC.loss_grad(b2, LEp, REp, ds, lid, lid + 1)
            assert np.abs(trace[q] - np.array([l0, l1, l2])).max() < 1e-9 * max(1.0, abs(l0)), (q, trace[q], l0, l1, l2)
            RC.bond_step(Wo, LE, RE, lid, ds, opts, True)
    finally:
        eng.close()


def test_imputation_from_a_complex_context():
    """mpst_impute on a context that holds a complex model and complex data (what a Fourier fit leaves behind) equals
    mpst_impute_model_run on the same tensors - the context's element type reaches the imputation engine."""
    import mpstime_jl_amd as mt
    ds, W = problem(24, 8, 4, 5, 2, 9, np.complex128)
    xs = -1.0 + (2.0 / 200) * np.arange(201)
    gphi = R.fourier_encode(xs, 4)
    rng = np.random.default_rng(0)
    miss = (rng.uniform(size=(24, 8)) < 0.4).astype(np.uint8)
    for dt in (np.complex128, np.complex64):
        eng = mt.SweepEngine(0)
        try:
            eng.set_options(chi_max=5)
            eng.set_dataset(0, ds.phi.astype(dt), ds.label_index, 2)
            eng.set_mps([t.astype(dt) for t in W])
            x1, e1, _ = eng.impute(0, miss, xs, gphi, method=0)
            Wm = [t.astype(dt).astype(np.complex128) for t in W]
            x2, e2, _ = eng.impute_model(Wm, ds.phi.astype(dt).astype(np.complex128), ds.label_index, miss, xs, gphi, method=0,
                                         compute="f64" if dt == np.complex128 else "f32")
            assert np.array_equal(x1, x2) and np.array_equal(e1, e2)
            assert np.all(x1[miss == 0] == 0.0) and np.any(x1[miss == 1] != 0.0)
        finally:
            eng.close()
