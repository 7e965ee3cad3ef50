"""Pin against REFERENCE-PRODUCED data: tests/golden/ref_ecg200_trained_mps.npz holds the tensors of
the reference's own serialised TrainedMPS (test/Data/ecg200/mps_saves/test_dataset.jld2, extracted by
tests/golden/extract_jld2_fixture.py): class-sorted raw ECG200 series, the product states the reference
encoded from them (default MPSOptions: RobustSigmoid + MinMax, Legendre_No_Norm d=5), and the MPS its
fitMPS returned (chi_max=25, label index on the last site).

What this pins: A14/A15 preprocessing+encoding bit-for-bit (1e-13), A12/A17/A18
container conventions and the canonical form the sweep leaves behind (left-orthonormal sites, unit
norm after normalize!), A13 contract_mps / classify on a real trained MPS.  The optimiser trajectory
of the sweep itself (A1-A11) has no reference vectors and stays unpinned.
"""
import os

import numpy as np
import pytest

import mpstime_jl_amd as mt
from oracle import ref_numpy as R

FIX = os.path.join(os.path.dirname(__file__), "golden", "ref_ecg200_trained_mps.npz")


@pytest.fixture(scope="module")
def ref():
    z = np.load(FIX)
    T = z["pstates"].shape[1]
    W = [z[f"W_{j}"] for j in range(T)]
    cd = z["class_distribution"]
    label_index = np.repeat(np.arange(len(cd)), cd)          # original_data is class-sorted (encode_dataset)
    return dict(X=z["original_data"], phi=z["pstates"], W=W, cd=cd, label_index=label_index,
                d=int(z["d"]), chi_max=int(z["chi_max"]), chi=z["chi"])


def test_oracle_encoding_reproduces_reference_pstates(ref):
    Xs, _ = R.transform_train_data(ref["X"])
    phi = R.legendre_encode(Xs, ref["d"])
    assert phi.shape == ref["phi"].shape
    assert np.max(np.abs(phi - ref["phi"])) < 1e-13


def test_package_encoding_reproduces_reference_pstates(ref):
    opts = mt.MPSOptions(verbosity=-1)                       # reference defaults: d=5, Legendre_No_Norm
    assert opts.d == ref["d"] and opts.chi_max == ref["chi_max"]
    enc = mt.model_encoding(opts.encoding)
    Xs, _ = mt.transform_train_data(ref["X"], opts, enc.range)
    y = ref["label_index"]
    ds = mt.encode_dataset(ref["X"], Xs, y, enc, opts.d, {0: 0, 1: 1})
    assert list(ds.class_distribution) == list(ref["cd"])
    assert np.max(np.abs(ds.phi - ref["phi"])) < 1e-13
    assert np.array_equal(ds.original_data, ref["X"])


def test_reference_mps_canonical_form(ref):
    W, chi = ref["W"], ref["chi"]
    assert chi[0] == 1 and chi[-1] == 1 and chi.max() == ref["chi_max"]
    assert W[-1].ndim == 4 and all(t.ndim == 3 for t in W[:-1])
    # chi_max and the rank bound (C*d rows on the last bond) of decomposeBT's truncation
    assert chi[-2] == len(ref["cd"]) * ref["d"]
    for t in W[:-1]:                                          # a forward sweep leaves left-orthonormal sites
        m = t.reshape(-1, t.shape[2])
        assert np.max(np.abs(m.T @ m - np.eye(m.shape[1]))) < 1e-12
    assert abs(R.mps_norm(W) - 1.0) < 1e-12                   # normalize!(W) at the end of fitMPS


def test_truncation_rule_on_reference_spectra(ref):
    """The only reference OUTPUT of decomposeBT's truncation in the tree: the bond spectra of the MPS the reference's
    fitMPS returned (cutoff 1e-10, maxdim 25).  With every site left of the last one left-orthonormal, the Schmidt
    spectrum of bond j is the singular-value set of the centre matrix moved there.  The last bond's spectrum is exactly
    what the final SVD of the run kept (then normalize!); the oracle's restatement of NDTensors' truncate! must be
    idempotent on all of them (keep everything the reference kept), the weight it kept must be 1 to rounding, and where
    the reference kept FEWER states than maxdim and than the rank bound (bond 2: 23 < 25 = d^2) the relative cutoff
    was the active rule: the smallest kept weight sits above it."""
    W, chi, chi_max = ref["W"], ref["chi"], ref["chi_max"]
    cur = W[-1].reshape(W[-1].shape[0], -1)
    spectra = {}
    for j in range(len(W) - 1, 0, -1):
        U, S, _ = np.linalg.svd(cur, full_matrices=False)
        spectra[j] = S
        cur = np.einsum("asb,bk->ask", W[j - 1], U * S).reshape(W[j - 1].shape[0], -1)
    for j, S in spectra.items():
        assert len(S) == chi[j]
        assert R.truncate_spectrum(S, chi_max, 1e-10) == chi[j]
        assert abs(np.sum(S ** 2) - 1.0) < 1e-12
        P = S ** 2
        assert P.min() > 1e-10 * P.sum()
    # the C restatement applies the same rule (same decisions on the same spectra)
    import ctypes as C
    from oracle import c_oracle
    lib = c_oracle.load()
    if hasattr(lib, "orc_truncate"):
        lib.orc_truncate.restype = C.c_int
        for j, S in spectra.items():
            buf = np.ascontiguousarray(S)
            assert lib.orc_truncate(buf.ctypes.data_as(C.POINTER(C.c_double)), len(S), chi_max, C.c_double(1e-10)) == chi[j]
    assert chi[2] == 23 < min(chi_max, ref["d"] ** 2)          # cutoff-limited, not maxdim- or rank-limited
    # had the two discarded states carried more than the cutoff, they would have been kept: re-attach a tail and check
    S2 = np.concatenate([spectra[2], [2e-5, 1e-6]])              # weights 4e-10 and 1e-12 of a unit-norm state
    assert R.truncate_spectrum(S2, chi_max, 1e-10) == 24        # the 1e-12 one goes, the 4e-10 one stays
    S3 = np.concatenate([spectra[2], [7e-6, 6e-6]])              # 4.9e-11 + 3.6e-11 = 8.5e-11 <= 1e-10: both go
    assert R.truncate_spectrum(S3, chi_max, 1e-10) == 23


def test_oracle_contract_mps_on_reference_mps(ref):
    ds = R.EncodedSet(ref["phi"], ref["label_index"], ref["cd"])
    yhat = R.contract_mps(ref["W"], ds.phi)
    pred = np.argmax(np.abs(yhat), axis=1)
    assert np.array_equal(pred, ref["label_index"])           # the saved model fits its training set exactly
    mse, kld, acc = R.mse_loss_acc(ref["W"], ds)[:3]
    assert acc == 1.0
    assert abs(kld - (-48.58386481729281)) < 1e-9


@pytest.mark.gpu
def test_engine_eval_on_reference_mps(ref):
    eng = mt.SweepEngine(0)
    eng.set_options(chi_max=ref["chi_max"])
    eng.set_dataset(0, ref["phi"], ref["label_index"], len(ref["cd"]))
    eng.set_mps(ref["W"])
    mse, kld, acc, conf = eng.eval(0)
    ds = R.EncodedSet(ref["phi"], ref["label_index"], ref["cd"])
    mse_o, kld_o, acc_o = R.mse_loss_acc(ref["W"], ds)[:3]
    assert acc == 1.0 and np.array_equal(conf, np.diag(ref["cd"]))
    assert abs(kld - kld_o) < 1e-10 * abs(kld_o)
    assert abs(mse - mse_o) < 1e-10 * abs(mse_o)
    pred, yh = eng.classify(0, return_overlaps=True)
    yo = R.contract_mps(ref["W"], ref["phi"])
    assert np.array_equal(pred, ref["label_index"])
    assert np.max(np.abs(yh - yo) / np.abs(yo).max()) < 1e-11
    eng.close()


@pytest.mark.gpu
def test_engine_sweep_from_reference_mps_matches_oracle(ref):
    """One teacher-free backward half sweep on the reference's real data from the reference's MPS:
    per-bond loss / truncation agree with the oracle (d=5, chi=25: d*chi=125, the engine's non-power-of-two path)."""
    from oracle.c_oracle import COracle
    opts = R.SweepOptions(chi_max=ref["chi_max"], eta=0.01)
    eng = mt.SweepEngine(0)
    eng.set_options(chi_max=opts.chi_max, eta=opts.eta, cutoff=opts.cutoff)
    eng.set_dataset(0, ref["phi"], ref["label_index"], len(ref["cd"]))
    eng.set_mps(ref["W"])
    eng.build_caches()
    co = COracle(ref["W"], ref["phi"], ref["label_index"], ref["cd"], opts.chi_max, eta=opts.eta)
    co.build_caches()
    T = len(ref["W"])
    nb = 12
    rec = co.sweep(max_bonds=nb, record=True)
    for k in range(nb):
        lid = T - 2 - k
        g = eng.bond_step(lid, True)
        o = rec["bonds_rec"][k]
        assert g["chi"] == o["chi"], (k, g["chi"], o["chi"])
        assert abs(g["loss"] - o["loss"]) <= 1e-8 * max(1.0, abs(o["loss"])), (k, g["loss"], o["loss"])
    eng.close()


@pytest.mark.gpu
def test_device_encoding_reproduces_reference_pstates(ref):
    """mpst_encode_dataset (device-side preprocessing + Legendre_No_Norm d=5) against the product states the
    reference itself stored: rows A14/A15 pinned for the HIP path too."""
    eng = mt.SweepEngine(0)
    norms, sec = eng.encode_dataset(0, ref["X"], ref["label_index"], len(ref["cd"]), basis="Legendre_No_Norm", d=ref["d"])
    phi = eng.get_encoded(0)
    assert phi.shape == ref["phi"].shape
    assert np.max(np.abs(phi - ref["phi"])) < 1e-13
    # and the engine can evaluate the reference's MPS on what it encoded itself
    eng.set_options(chi_max=ref["chi_max"])
    eng.set_mps(ref["W"])
    mse, kld, acc, conf = eng.eval(0)
    assert acc == 1.0 and abs(kld - (-48.58386481729281)) < 1e-8
    eng.close()


@pytest.mark.gpu
def test_sweeps_on_reference_ecg200_data_match_oracle(ref):
    """The real data of the reference's fixture (ECG200 train set, N=100, T=96) with the reference's default
    hyper-parameters (d=5, chi_max=25, KLD/TSGO, eta=0.01) from a fresh random MPS: every bond update of the
    first two sweeps against the C oracle, teacher-forced (a free-running fit diverges chaotically here as well:
    train KLD -44.25 vs -44.81 after one sweep).  d*chi = 125 exercises the non-power-of-two eigensolver path."""
    from oracle.c_oracle import COracle
    from tests.helpers import teacher_forced_sweep
    opts = mt.MPSOptions(verbosity=-1)
    T = ref["phi"].shape[1]
    W0 = mt.generate_startingMPS(opts.chi_init, T, opts.d, 2, 4321)
    co = COracle(W0, ref["phi"], ref["label_index"], ref["cd"], opts.chi_max, eta=opts.eta, rebuild_caches=False)
    co.build_caches()
    eng = mt.SweepEngine(0)
    eng.set_options(chi_max=opts.chi_max, eta=opts.eta, cutoff=opts.cutoff)
    eng.set_dataset(0, ref["phi"], ref["label_index"], 2)
    for sweep in range(2):
        worst, flips = teacher_forced_sweep(eng, co, ref["phi"], T, overlap_every=7)
        assert worst["loss"] < 1e-10 and worst["grad"] < 1e-8 and worst["S"] < 1e-9 and worst["overlap"] < 1e-8, (sweep, worst)
        assert flips <= 2
    eng.close()


@pytest.mark.gpu
def test_fit_outcome_brackets_the_reference_fit(ref):
    """Outcome-level evidence for the sweep (its trajectory cannot be pinned: Julia's RNG stream for the initial MPS is not
    reproducible): the MPS the reference trained on this data with the default MPSOptions scores KLD -48.58 / accuracy 1.0
    on its training set; fits of this engine on the same data with the same options from four of its own random starts
    pass that value between their 5th and 10th sweep (-47.8 ... -48.1 after 5, -49.0 ... -49.2 after 10), accuracy 1.0."""
    T = ref["phi"].shape[1]
    eng = mt.SweepEngine(0)
    try:
        eng.set_options(chi_max=25, eta=0.01, cutoff=1e-10)
        eng.set_dataset(0, ref["phi"], ref["label_index"], 2)
        eng.set_mps(ref["W"])
        _, kld_ref, acc_ref, _ = eng.eval(0)
    finally:
        eng.close()
    assert acc_ref == 1.0 and abs(kld_ref - (-48.58386)) < 1e-4
    y = np.repeat(np.arange(len(ref["cd"])), ref["cd"])
    for seed in (1234, 1, 2, 3):
        opts = mt.MPSOptions(verbosity=-1, nsweeps=10, init_rng=seed)
        tr, info, _ = mt.fitMPS(ref["X"], y, None, None, opts)
        kl = info["train_KL_div"]
        assert info["train_acc"][-1] == 1.0
        assert kl[5] > kld_ref > kl[-1], (seed, kl)            # entry k = after k sweeps (entry 0: the initial MPS)
        assert abs(kl[-1] - kld_ref) < 1.0


REF_MPS_DIGEST = "0559cc1372c9561503946707a2d636d4413f8d9712b72c076c622d1619e412b6"


def test_reference_mps_content_digest_and_npz_round_trip(ref, tmp_path):
    """The container-independent digest of the reference's own trained MPS (mps_content_digest: shapes + entries in the
    index order (left bond, site, right bond[, label])).  mpstime.jl_amd/julia/roundtrip_check.jl computes the same digest in
    Julia from test/Data/ecg200/mps_saves/test_dataset.jld2 and from a .npz this package wrote: a maintainer with Julia
    closes the save / load round trip (test/save_load.jl:17-24) by comparing the three values."""
    assert mt.mps_content_digest(ref["W"]) == REF_MPS_DIGEST
    opts = mt.MPSOptions(d=ref["d"], chi_max=ref["chi_max"], encoding="Legendre_No_Norm", verbosity=-1)
    td = mt.EncodedTimeSeriesSet(ref["phi"], ref["label_index"].astype(np.int64), ref["label_index"].astype(np.int32), ref["X"], ref["cd"])
    path = tmp_path / "ref_model.npz"
    mt.save_trained_mps(str(path), mt.TrainedMPS(ref["W"], opts, td))
    back = mt.load_trained_mps(str(path))
    assert mt.mps_content_digest(back.mps) == REF_MPS_DIGEST
