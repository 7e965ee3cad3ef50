"""fitMPS / TrainedMPS / classify - the reference's public training API driven by the HIP
sweep engine (src/Training/RealRealHighDimension.jl:383-890, src/summary.jl:116-177).

The overload chain of the reference (raw data -> rescaled data -> encoded states -> sweep)
is kept as three functions; the last one, ``fit_encoded``, is the drop-in seam
``fitMPS(W::MPS, training_states_meta, testing_states_meta, opts)`` (:587).
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from .encodings import (EncodedTimeSeriesSet, Encoding, encode_dataset, model_encoding, transform_data)
from .engine import SweepEngine
from .options import MPSOptions, engine_options, numpy_dtype, safe_options


@dataclass
class TrainedMPS:
    """TrainedMPS (src/Structs/options.jl:422-427): (mps, opts, train_data)."""

    mps: List[np.ndarray]          # site tensors (Dl, d, Dr[, C]); label on the last site
    opts: MPSOptions
    train_data: EncodedTimeSeriesSet

    def __eq__(self, other):       # src/Structs/operations.jl:4-36
        return (isinstance(other, TrainedMPS) and self.opts == other.opts and len(self.mps) == len(other.mps)
                and all(np.array_equal(a, b) for a, b in zip(self.mps, other.mps)))


def generate_startingMPS(chi_init, T, d, num_classes, init_rng=None, dtype=np.float64):
    """generate_startingMPS (RealRealHighDimension.jl:1-41): random Gaussian MPS with bond
    dimension chi_init, label index on the last site, normalised, orthogonality centre on the
    last site.  Julia's MersenneTwister/ITensors.random_mps stream cannot be reproduced outside
    Julia; the distribution and canonical form are the same (NumPy default_rng(init_rng))."""
    rng = np.random.default_rng(init_rng)
    dims = [1] + [int(min(chi_init, d ** min(j, T - j, 40))) for j in range(1, T)] + [1]
    dtype = np.dtype(dtype)
    wide = np.complex128 if dtype.kind == "c" else np.float64       # canonicalise in double precision, store in opts.dtype
    W = []
    for j in range(T):
        shape = (dims[j], d, dims[j + 1]) + ((num_classes,) if j == T - 1 else ())
        t = rng.standard_normal(shape)
        if dtype.kind == "c":                    # random_mps(ComplexF64, ...): Gaussian real and imaginary parts (:17)
            t = t + 1j * rng.standard_normal(shape)
        W.append(t.astype(wide))
    for j in range(T - 1):                       # orthogonalize!(W, T) (:37)
        Dl, dd, Dr = W[j].shape
        Q, Rm = np.linalg.qr(W[j].reshape(Dl * dd, Dr))
        W[j] = Q.reshape(Dl, dd, Q.shape[1])
        # the carried factor is renormalised at every site (the final normalize! fixes the scale anyway): its norm grows like
        # (d chi)^(j/2) and its SQUARE overflowed fp64 for d = 8, T = 200 - an all-zero last site, yhat = 0, KLD = inf
        W[j + 1] = np.tensordot(Rm / np.linalg.norm(Rm), W[j + 1], axes=(1, 0))
    W[-1] = W[-1] / np.linalg.norm(W[-1])        # normalize!(W) (:32)
    return [t.astype(dtype) for t in W]


_INFO_KEYS = ("train_loss", "train_acc", "test_loss", "time_taken", "train_KL_div")
_INFO_TEST_KEYS = ("test_acc", "test_KL_div", "test_conf")


def fit_encoded(W, training_states_meta: EncodedTimeSeriesSet, testing_states_meta: Optional[EncodedTimeSeriesSet],
                opts: MPSOptions = MPSOptions(), engine: Optional[SweepEngine] = None, device: int = 0,
                shard=None, preloaded: bool = False):
    """fitMPS(W::MPS, training_states_meta, testing_states_meta, opts) (:587-890).

    Returns (TrainedMPS, training_information, testing_states_meta).  training_information has the
    reference's keys and lengths (nsweeps+2 when log_level > 0; time_taken 0.0 first, NaN last).
    ``shard`` = (rank, world_size, communicator-setup callable) for batch sharding (see distributed.py).
    ``preloaded``: the engine already holds both data sets (device-side encoding, fitMPS(device_encode=True)).
    """
    opts = safe_options(opts)
    eopt = engine_options(opts)
    tr = training_states_meta
    te = testing_states_meta if testing_states_meta is not None else EncodedTimeSeriesSet.empty()
    if np.any(np.diff(tr.label_index) < 0):
        raise AssertionError("Training data must be sorted by class!")            # :624
    if len(te) and np.any(np.diff(te.label_index) < 0):
        raise AssertionError("Testing data must be sorted by class!")             # :625
    dt = numpy_dtype(opts.dtype)                                                  # :442: everything is built in opts.dtype
    if np.iscomplexobj(tr.phi) and dt.kind != "c":
        raise RuntimeError("Using a complex valued encoding but the MPS is real. If using a complex-valued custom "
                           "encoding, set 'dtype <: Complex' in MPSOptions")     # :462-464
    has_test = len(te) > 0
    C = int(W[-1].shape[3]) if np.ndim(W[-1]) == 4 else None
    if C is None:
        raise ValueError("the label index must sit on the last site of the starting MPS (find_label, utils.jl:342-354)")
    verbosity = opts.verbosity
    own_engine = engine is None
    eng = engine or SweepEngine(device)
    try:
        eng.set_options(rebuild_caches=False, track_cost=opts.track_cost, **eopt)
        gcounts = None
        if shard is not None:
            tr_local, gcounts = shard.split(tr)
            te_local = shard.split(te)[0] if has_test else te
            shard.attach(eng)
        else:
            tr_local, te_local = tr, te
        if not preloaded:
            eng.set_dataset(0, tr_local.phi, tr_local.label_index, C, gcounts, dtype=dt)
            if has_test:
                eng.set_dataset(1, te_local.phi, te_local.label_index, C, dtype=dt)
        eng.set_mps(W)
        if shard is not None and getattr(shard, "oneshot", False):
            shard.attach_oneshot(eng)                                           # inbox slots are sized from the gradient buffer
        verbosity > -1 and print(f"Using {opts.update_iters} iterations per update.")
        eng.build_caches()                                                       # :631

        info = {k: [] for k in _INFO_KEYS}
        if has_test:
            info.update({k: [] for k in _INFO_TEST_KEYS})

        def log(time_taken):
            if opts.log_level <= 0:
                return None
            mse, kld, acc, _ = eng.eval(0)
            info["train_loss"].append(mse)
            info["train_acc"].append(acc)
            info["time_taken"].append(time_taken)
            info["train_KL_div"].append(kld)
            if verbosity > -1:
                print(f"Training KL Div. {kld} | Training acc. {acc}.")
            if has_test:
                tm, tk, ta, conf = eng.eval(1)
                info["test_loss"].append(tm)
                info["test_acc"].append(ta)
                info["test_KL_div"].append(tk)
                info["test_conf"].append(conf)
                if verbosity > -1:
                    print(f"Test KL Div. {tk} | Testing acc. {ta}.\n\nTest conf: {conf.tolist()}.")
            return acc

        log(0.0)                                                                 # :657-689
        for its in range(opts.nsweeps):                                          # :726
            if verbosity > -1:
                print(f"Using optimiser CustomGD with the \"{engine_options(opts, its)['bbopt']}\" algorithm")
                print(f"Starting backward sweeep: [{its + 1}/{opts.nsweeps}]")
            if its > 0 and (isinstance(opts.loss_grad, tuple) or isinstance(opts.bbopt, tuple)):
                eng.set_options(rebuild_caches=False, track_cost=opts.track_cost, **engine_options(opts, its))   # :727-728: this sweep's loss / optimiser
            st = eng.sweep()                                                     # :727-808
            # both half-sweeps run inside the one call: the reference's mid-sweep lines (:766, :772) are printed where they fall in its
            # output - between the backward half's and the forward half's per-bond lines when those are printed, straight after otherwise
            mid = lambda: verbosity > -1 and (print("Backward sweep finished."),
                                              print(f"Starting forward sweep: [{its + 1}/{opts.nsweeps}]"))
            if opts.track_cost and verbosity >= 1:
                # what custGD / TSGO (loss_functions.jl:50-52,80-82) and apply_update (:181-184) print, bond by bond
                trace = eng.loss_trace()
                nb = len(W) - 1
                for q in range(2 * nb):
                    if q == nb:
                        mid()
                    lid = nb - 1 - q if q < nb else q - nb
                    for it in range(opts.update_iters):
                        print(f"Loss before step {it + 1}: {trace[q, it]}")
                    print(f"Loss at site {lid + 1}*{lid + 2}: {trace[q, opts.update_iters]}")
            else:
                mid()
            if verbosity > -1:
                print(f"Finished sweep {its + 1}. Time for sweep: {round(st['seconds'], 2)}s")
            acc = log(st["seconds"])
            if opts.exit_early and acc == 1.0:                                   # :847
                break
        eng.normalize()                                                          # :852
        verbosity > -1 and print("\nMPS normalised!\n")
        log(float("nan"))                                                        # :854-885
        Wout = eng.get_mps()
    finally:
        if own_engine:
            eng.close()
    return TrainedMPS(Wout, opts, tr), info, te


def fitMPS(X_train, y_train=None, X_test=None, y_test=None, opts: MPSOptions = MPSOptions(),
           custom_encoding: Optional[Encoding] = None, W=None, device_encode: bool = False, **kw):
    """fitMPS(X_train, y_train, X_test, y_test, opts[, custom_encoding]) (:383-416) and the
    overloads without test data / labels (:413,:416).  X_* are (N, T) matrices, rows = series.
    Returns (TrainedMPS, training_information, encoded_test_states).

    ``device_encode=True`` (the closed-form bases: Legendre, Fourier, Stoudenmire, Sahand, Uniform) preprocesses and encodes on the GPU
    (mpst_encode_dataset): the raw matrices are uploaded, the product states are downloaded once for the
    returned EncodedTimeSeriesSets."""
    opts = safe_options(opts)
    X_train = np.asarray(X_train, dtype=np.float64)
    N, T = X_train.shape
    y_train = np.zeros(N, dtype=np.int64) if y_train is None else np.asarray(y_train)
    X_test = np.zeros((0, T)) if X_test is None else np.asarray(X_test, dtype=np.float64)
    y_test = np.zeros(X_test.shape[0], dtype=np.int64) if y_test is None else np.asarray(y_test)
    if custom_encoding is not None and opts.encoding.lower() != "custom":
        raise ValueError("To use a custom encoding, you must set 'encoding = :Custom' in MPSOptions")   # :393
    if X_train.shape[0] != y_train.shape[0]:
        raise AssertionError("Size of training dataset and number of training labels are different!")   # :460
    if X_test.shape[0] != y_test.shape[0]:
        raise AssertionError("Size of testing dataset and number of testing labels are different!")     # :461
    if X_test.size and X_test.shape[1] != T:
        raise AssertionError("The number of sites supported by the MPS, training, and testing data do not match! ")
    enc = model_encoding(opts.encoding, custom_encoding)
    if enc.iscomplex and opts.dtype == "Float64":
        raise RuntimeError("Using a complex valued encoding but the MPS is real. If using a complex-valued custom "
                           "encoding, set 'dtype <: Complex' in MPSOptions")                              # :466-468
    classes = np.unique(y_train)                                                                         # :474
    if not np.issubdtype(classes.dtype, np.integer):
        raise AssertionError("Classes must be integers")                                                 # :484
    if len(np.setdiff1d(np.unique(y_test), classes)):
        raise ValueError("Test set has classes not present in the training set, this is currently unsupported.")
    class_keys = {c: i for i, c in enumerate(classes.tolist())}                                          # :485-486
    num_classes = len(classes)
    if W is None:
        W = generate_startingMPS(opts.chi_init, T, opts.d, num_classes, opts.init_rng, numpy_dtype(opts.dtype))   # :433-435
    if device_encode:
        return _fit_device_encoded(W, X_train, y_train, X_test, y_test, opts, enc, class_keys, **kw)
    Xtr_s, Xte_s, norms, oob = transform_data(X_train, X_test, opts, enc.range)                          # :445
    train_states = encode_dataset(X_train, Xtr_s, y_train, enc, opts.d, class_keys)                      # :489
    test_states = encode_dataset(X_test, Xte_s, y_test, enc, opts.d, class_keys) if X_test.size else \
        EncodedTimeSeriesSet.empty()
    return fit_encoded(W, train_states, test_states, opts, **kw)                                         # :556-560


def _fit_device_encoded(W, X_train, y_train, X_test, y_test, opts, enc, class_keys, engine=None, device=0, **kw):
    """fitMPS with transform_data + encode_dataset (:445, :489) done by the engine."""
    if kw.get("shard") is not None:
        raise NotImplementedError("device_encode together with batch sharding: encode per shard with mpst_encode_dataset")
    C = len(class_keys)

    def sorted_set(X, y):
        order = np.argsort(y, kind="stable")                                      # encodings.jl:43
        ys = np.asarray(y)[order]
        li = np.array([class_keys[v] for v in ys.tolist()], dtype=np.int32)
        _, counts = np.unique(ys, return_counts=True)
        return np.ascontiguousarray(X[order]), ys, li, counts.astype(np.int64)

    own = engine is None
    eng = engine or SweepEngine(device)
    try:
        Xs, ys, li, counts = sorted_set(X_train, y_train)
        eng.set_dtype(numpy_dtype(opts.dtype))
        common = dict(basis=enc.name, d=opts.d, sigmoid_transform=opts.sigmoid_transform, minmax=opts.minmax,
                      data_bounds=opts.data_bounds, enc_range=enc.range)
        norms, _ = eng.encode_dataset(0, Xs, li, C, **common)
        train_states = EncodedTimeSeriesSet(eng.get_encoded(0), ys, li, Xs, counts)
        test_states = EncodedTimeSeriesSet.empty()
        if X_test.size:
            Xt, yt, lt, ct = sorted_set(X_test, y_test)
            eng.encode_dataset(1, Xt, lt, C, norms=norms, **common)
            test_states = EncodedTimeSeriesSet(eng.get_encoded(1), yt, lt, Xt, ct)
        return fit_encoded(W, train_states, test_states, opts, engine=eng, preloaded=True)
    finally:
        if own:
            eng.close()


def classify(mps: TrainedMPS, X_or_states, engine: Optional[SweepEngine] = None, device: int = 0):
    """classify(mps, test_states) (summary.jl:116-136) and classify(mps, X_test) (:155-177):
    predicted labels (original label values) by maximum overlap |yhat|^2."""
    opts = safe_options(mps.opts)
    labels = np.unique(mps.train_data.labels)
    if isinstance(X_or_states, EncodedTimeSeriesSet):
        states = X_or_states
    else:
        X_test = np.asarray(X_or_states, dtype=np.float64)
        enc = model_encoding(opts.encoding)
        _, Xte_s, _, _ = transform_data(mps.train_data.original_data, X_test, opts, enc.range)          # :160
        n = X_test.shape[0]
        states = encode_dataset(X_test, Xte_s, np.full(n, -1), enc, opts.d, {-1: 0})                     # :175 (unsorted: all one label)
    if len(states) == 0:
        return np.zeros(0, dtype=labels.dtype)
    own = engine is None
    eng = engine or SweepEngine(device)
    try:
        C = int(mps.mps[-1].shape[3])
        eng.set_options(**engine_options(opts))
        eng.set_dataset(0, mps.train_data.phi[:1], mps.train_data.label_index[:1], C)
        eng.set_dataset(1, states.phi, np.zeros(len(states), dtype=np.int32), C)
        eng.set_mps(mps.mps)
        pred = eng.classify(1)
    finally:
        if own:
            eng.close()
    return labels[pred]


def trendy_sine(T, n, period=None, slope=None, phase=None, sigma=0.0, rng=None):
    """trendy_sine (src/Simulation/toy_data.jl:53-85): x_t = sin(2 pi t/tau + psi) + m t/T + sigma n_t.
    period/slope/phase: None -> default uniform range, float -> fixed, tuple -> uniform bounds,
    list -> discrete choice (toy_data.jl:16-28)."""
    rng = rng or np.random.default_rng()

    def draw(spec, default):
        if spec is None:
            return rng.uniform(*default)
        if isinstance(spec, tuple):
            return rng.uniform(*spec)
        if isinstance(spec, (list, np.ndarray)):
            return rng.choice(spec)
        return float(spec)

    per = [draw(period, (1.0, 50.0)) for _ in range(n)]
    slo = [draw(slope, (-5.0, 5.0)) for _ in range(n)]
    pha = [draw(phase, (0.0, 2 * math.pi)) for _ in range(n)]
    ts = np.arange(1, T + 1, dtype=np.float64)
    X = np.empty((n, T))
    for i in range(n):
        X[i] = np.sin(2 * np.pi / per[i] * ts + pha[i]) + slo[i] * ts / T + sigma * rng.standard_normal(T)
    return X, {"period": per, "slope": slo, "phase": pha, "sigma": sigma, "T": T, "n": n}


def save_trained_mps(path, trained: TrainedMPS):
    """The counterpart of ``@save fpath mps`` (test/save_load.jl:17-24) for this package: the three fields of TrainedMPS
    (src/Structs/options.jl:422-427) - site tensors, the MPSOptions, the EncodedTimeSeriesSet - in one .npz.  (JLD2 /
    HDF5 cannot be written here: no HDF5 library in the image; the reference's own .jld2 fixture is read by
    tests/golden/extract_jld2_fixture.py.)"""
    import json
    td = trained.train_data
    opts = {k: (list(v) if isinstance(v, tuple) else v) for k, v in trained.opts.asdict().items()}
    np.savez_compressed(path, n_sites=len(trained.mps), opts=json.dumps(opts), timeseries=td.phi, labels=td.labels,
                        label_index=td.label_index, original_data=td.original_data, class_distribution=td.class_distribution,
                        **{f"mps_{j}": t for j, t in enumerate(trained.mps)})


def mps_content_digest(mps) -> str:
    """SHA-256 over the site tensors in order: for every site its shape (int64, little endian) followed by its float64
    (or complex128) entries in C order of (left bond, site, right bond[, label]).  Independent of the container (.npz zip
    metadata, JLD2): the Julia snippet in INTEGRATION.md computes the same digest from an ITensors MPS, which closes the
    save / load round trip (test/save_load.jl:17-24) for a maintainer who has Julia."""
    import hashlib
    h = hashlib.sha256()
    for t in mps:
        a = np.ascontiguousarray(t, dtype=np.complex128 if np.iscomplexobj(t) else np.float64)
        h.update(np.asarray(a.shape, dtype="<i8").tobytes())
        h.update(a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes())
    return h.hexdigest()


def load_trained_mps(path) -> TrainedMPS:
    """A model saved by save_trained_mps (.npz) - or, for a path ending in .jld2, a TrainedMPS the reference itself saved
    with JLD2 (jld2.py)."""
    import json
    if str(path).endswith(".jld2"):
        from .jld2 import load_trained_mps_jld2
        return load_trained_mps_jld2(path)
    z = np.load(path if str(path).endswith(".npz") else str(path) + ".npz", allow_pickle=False)
    o = json.loads(str(z["opts"]))
    for k in ("rescale", "data_bounds"):
        o[k] = tuple(o[k])
    for k in ("loss_grad", "bbopt"):
        if isinstance(o[k], list):
            o[k] = tuple(o[k])
    td = EncodedTimeSeriesSet(z["timeseries"], z["labels"], z["label_index"], z["original_data"], z["class_distribution"])
    return TrainedMPS([z[f"mps_{j}"] for j in range(int(z["n_sites"]))], MPSOptions(**o), td)
