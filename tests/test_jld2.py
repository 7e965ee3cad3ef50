"""The HDF5-free JLD2 reader (mpstime.jl_amd/jld2.py) on the reference's own save file
test/Data/ecg200/mps_saves/test_dataset.jld2 (kept as tests/golden/ref_test_dataset.jld2 - a data file of the
reference's test tree): named arrays, the TrainedMPS struct by field name, and the model it yields."""
import os

import numpy as np
import pytest

import mpstime_jl_amd as mt
from oracle import ref_numpy as R

HERE = os.path.dirname(__file__)
JLD = os.path.join(HERE, "golden", "ref_test_dataset.jld2")
NPZ = os.path.join(HERE, "golden", "ref_ecg200_trained_mps.npz")


def test_named_arrays_come_back_in_julia_shape():
    f = mt.JLD2File(JLD)
    assert f.keys() == ["mps", "X_train", "y_train", "X_test", "y_test"]
    X, y, Xt, yt = f.read("X_train"), f.read("y_train"), f.read("X_test"), f.read("y_test")
    assert X.shape == (100, 96) and X.dtype == np.float64 and y.shape == (100,) and y.dtype == np.int64     # ECG200: 100 series of 96 points
    assert Xt.shape == (100, 96) and yt.shape == (100,)
    assert set(int(v) for v in np.unique(y)) == {0, 1} and np.isfinite(X).all()
    with pytest.raises(KeyError):
        f.read("nope")
    assert set(mt.read_jld2(JLD)) == set(f.keys())


def test_trained_mps_struct_is_read_by_field_name():
    raw = mt.JLD2File(JLD).read("mps")
    assert set(raw) == {"mps", "opts", "train_data"}
    o = raw["opts"]
    assert (o["d"], o["chi_max"], o["encoding"], o["loss_grad"], o["bbopt"], o["svd_alg"]) == (5, 25, "Legendre", "KLD", "TSGO", "divide_and_conquer")
    assert o["rescale"] == {"1": False, "2": True} and o["dtype"]["name"] == "Core.Float64"
    assert len(raw["mps"]["data"]) == 96 and raw["mps"]["llim"] == 0 and raw["mps"]["rlim"] == 97


def test_loaded_model_equals_the_extracted_fixture_and_the_reference_defaults():
    tm = mt.load_trained_mps(JLD)                       # .jld2 -> load_trained_mps_jld2
    z = np.load(NPZ)                                    # the same file, read by tests/golden/extract_jld2_fixture.py
    assert len(tm.mps) == 96
    assert all(np.array_equal(tm.mps[j], z[f"W_{j}"]) for j in range(96))
    assert tm.mps[-1].shape == (10, 5, 1, 2) and tm.mps[0].shape == (1, 5, 5)
    assert np.array_equal(tm.train_data.phi, z["pstates"]) and np.array_equal(tm.train_data.original_data, z["original_data"])
    assert list(tm.train_data.class_distribution) == [31, 69]
    assert np.array_equal(tm.train_data.label_index, np.repeat([0, 1], [31, 69]))          # class-sorted, 0-based here
    # every option the reference stored equals its documented default except the ones this fit changed
    ref_opts = mt.MPSOptions(verbosity=-1, encoding="Legendre", log_level=0)
    assert tm.opts == ref_opts
    # the stored model classifies its own training set as well as the reference's fit did (contract_mps, summary.jl:4-35)
    yhat = R.contract_mps(tm.mps, tm.train_data.phi)
    acc = np.mean(np.argmax(np.abs(yhat), 1) == tm.train_data.label_index)
    assert acc > 0.95


def test_round_trip_through_the_npz_wire_format(tmp_path):
    tm = mt.load_trained_mps(JLD)
    p = str(tmp_path / "model.npz")
    mt.save_trained_mps(p, tm)
    back = mt.load_trained_mps(p)
    assert back == tm and np.array_equal(back.train_data.phi, tm.train_data.phi)
    # the digest julia/roundtrip_check.jl expects from the same MPS on the Julia side
    assert mt.mps_content_digest(back.mps) == mt.mps_content_digest(tm.mps) == "0559cc1372c9561503946707a2d636d4413f8d9712b72c076c622d1619e412b6"


def test_unsupported_files_fail_loudly(tmp_path):
    p = tmp_path / "x.jld2"
    p.write_bytes(b"not a jld2 file" * 100)
    with pytest.raises(ValueError, match="superblock"):
        mt.JLD2File(str(p))


@pytest.mark.gpu
def test_reference_saved_model_classifies_the_reference_test_set_on_the_device():
    """classify(mps, X_test) (summary.jl:155-177) with the model and the test split both read from the reference's file."""
    f = mt.JLD2File(JLD)
    tm = mt.load_trained_mps_jld2(JLD)
    Xt, yt = f.read("X_test"), f.read("y_test")
    pred = mt.classify(tm, Xt)
    acc = float(np.mean(pred == yt))
    # oracle on the same encoded states
    opts = tm.opts
    enc = mt.model_encoding(opts.encoding)
    _, Xs, _, _ = mt.transform_data(tm.train_data.original_data, Xt, opts, enc.range)
    st = mt.encode_dataset(Xt, Xs, np.full(len(Xt), -1), enc, opts.d, {-1: 0})
    ref = np.argmax(np.abs(R.contract_mps(tm.mps, st.phi)), 1)
    assert np.array_equal(pred, np.unique(tm.train_data.labels)[ref])
    assert acc > 0.8, acc
