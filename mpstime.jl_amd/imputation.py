"""Imputation API of the reference (src/Imputation/imputation.jl) on top of the device-side engine ``mpst_impute``.

``init_imputation_problem`` / ``MPS_impute`` keep the reference's names and argument meaning (one instance at a time, as
the reference's callers use them); ``impute_dataset`` is the batched entry the engine is built for: every test instance
with its own set of missing sites in one call.  Values cross the C ABI in the encoding's domain; the pre-processing
of ``get_predictions`` (imputation.jl:264-410: mask with the training mean, transform with the train-fitted
normalisations, invert them on the way out) stays on the host.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib as L
from .encodings import model_encoding, transform_test_data, transform_train_data
from .engine import SweepEngine
from .options import MPSOptions, engine_options, safe_options

METHODS = {"median": 0, "mode": 1, "ITS": 2, "mean": 3}
ORDERS = {"forwards": 0, "backwards": 1}


def invert_test_transform(X_scaled, oob_rescales, norms, opts: MPSOptions, enc_range):
    """utils.jl:299-334 (rows are series here)."""
    X = np.array(X_scaled, dtype=np.float64, copy=True)
    one = X.ndim == 1
    X = np.atleast_2d(X)
    a, b = enc_range
    X = (X - a) / (b - a)
    for i, lb_shift, ub_scale in oob_rescales:
        X[int(i)] = X[int(i)] * ub_scale + lb_shift
    if opts.minmax:
        lb, ub = opts.data_bounds
        X = (X - lb) / (ub - lb)
    if norms.minmax is not None:                    # denormalize!, in reverse order of application
        lo, hi = norms.minmax
        X = X * (hi - lo) + lo
    if norms.sigmoid is not None:
        med, iqr = norms.sigmoid
        with np.errstate(divide="ignore", invalid="ignore"):
            X = med - (iqr / 1.35) * np.log(1.0 / X - 1.0)
    return X[0] if one else X


def mar(X, fraction_missing=0.5, rng=None):
    """mar(X, fraction, BlockMissingMAR()) (src/Simulation/missing_data_mechanisms.jl:114-152): a block of consecutive
    missing values whose start is uniform over the valid positions.  Returns (X_corrupted, missing_idxs) (0-based)."""
    if not 0.0 <= fraction_missing <= 1.0:
        raise ValueError("fraction_missing must be between 0 and 1")
    rng = rng or np.random.default_rng()
    X = np.asarray(X, dtype=np.float64)
    n = len(X)
    npts = int(round(n * fraction_missing))
    start = int(rng.integers(0, n - npts + 1))
    idx = np.arange(start, start + npts)
    Xc = X.copy()
    Xc[idx] = np.nan
    return Xc, idx


@dataclass
class EncodedDataRange:                 # imputation.jl:2-8
    dx: float
    guess_range: tuple
    xvals: np.ndarray
    xvals_enc: np.ndarray               # (ngrid, d): time-independent encodings share one table (:100-106)


@dataclass
class ImputationProblem:                # imputation.jl:10-20
    mps: list                           # the trained MPS with its label index (the engine slices classes itself)
    X_train: np.ndarray
    y_train: np.ndarray
    X_test: np.ndarray
    y_test: np.ndarray
    opts: MPSOptions
    x_guess_range: EncodedDataRange
    class_map: dict


def init_imputation_problem(W, X_test, y_test=None, dx: float = 1e-4, guess_range=None, verbosity: int = 1):
    """init_imputation_problem(W::TrainedMPS, X_test, y_test; dx, guess_range) (imputation.jl:143-190): the candidate
    values ``range(guess_range...; step=dx)`` and their encoded states are tabulated once."""
    opts = safe_options(W.opts)
    enc = model_encoding(opts.encoding)
    if guess_range is None:
        guess_range = tuple(enc.range)
    X_test = np.asarray(X_test, dtype=np.float64)
    y_test = np.zeros(X_test.shape[0], dtype=np.int64) if y_test is None else np.asarray(y_test)
    n = int(np.floor((guess_range[1] - guess_range[0]) / dx + 1e-9)) + 1
    xvals = guess_range[0] + dx * np.arange(n)
    td = W.train_data
    classes = np.unique(td.labels)
    states = enc.encode(xvals, opts.d)
    rng = EncodedDataRange(dx, guess_range, xvals, np.ascontiguousarray(states, dtype=np.complex128 if np.iscomplexobj(states) else np.float64))
    if verbosity > 0:
        print(f" - Dataset has {td.original_data.shape[0]} training samples and {X_test.shape[0]} testing samples.")
        print(f" - {len(classes)} class(es) were detected.")
    return ImputationProblem(W.mps, td.original_data, np.asarray(td.labels), X_test, y_test, opts, rng,
                             {c: i for i, c in enumerate(classes.tolist())})


def _scaled_instances(imp: ImputationProblem, rows, masks):
    """get_predictions' pre-processing (imputation.jl:283-297) for several instances at once: the missing region is
    overwritten with the training mean BEFORE the test transform (its per-series out-of-bounds rescale sees the masked
    series), the full series is transformed separately as the target in the encoding's domain."""
    enc = model_encoding(imp.opts.encoding)
    _, norms = transform_train_data(imp.X_train, imp.opts, enc.range)
    raw = imp.X_test[rows]
    full, _ = transform_test_data(raw, norms, imp.opts, enc.range)
    masked = raw.copy()
    masked[masks] = np.mean(imp.X_train)
    scaled, oob = transform_test_data(masked, norms, imp.opts, enc.range)
    return enc, norms, raw, full, scaled, oob


def impute_dataset(imp: ImputationProblem, missing_mask, method: str = "median", rows=None, invert_transform: bool = True,
                   get_wmad: bool = True, rng=None, engine: Optional[SweepEngine] = None, device: int = 0, return_seconds=False,
                   impute_order: str = "forwards", rejection_threshold=None, max_trials: int = 10, compute: str = "f64", shard=None):
    """Impute every instance of ``imp.X_test[rows]`` (default: all) at the sites where ``missing_mask`` is True, each
    with the MPS of its class.  Returns (X_imputed, pred_err) in the original units (``invert_transform``) or in the
    encoding's domain; pred_err is the weighted median absolute deviation for ``method="median"``, the standard
    deviation for ``"mean"`` and None otherwise.  ``"ITS"`` draws one trajectory per instance from ``rng``
    (``rejection_threshold`` None is the reference's ``:none``).  Complex encodings (Fourier, Sahand) and
    ``compute="f32"`` (fp32 chain contractions, fp64 densities) go through ``mpst_impute_model_run``.  With a ``shard``
    (distributed.Shard) every rank imputes its slice of the rows - instances are independent, there is no collective on
    the data path - and the results are gathered on every rank."""
    if method not in METHODS:
        raise ValueError("Invalid method. Choose :mean, :mode, :median, :kNearestNeighbour, :flatBaseline or :ITS"
                         if method not in ("kNearestNeighbour", "flatBaseline") else
                         f"method {method!r} is evaluated on the host in the reference and is not part of the device engine")
    if impute_order not in ORDERS:
        raise ValueError('impute_order must be either ":forwards" or ":backwards"')
    rows = np.arange(imp.X_test.shape[0]) if rows is None else np.asarray(rows)
    mask = np.asarray(missing_mask, dtype=bool)
    assert mask.shape == (len(rows), imp.X_test.shape[1])
    if shard is not None and shard.world > 1:
        return _impute_sharded(imp, mask, method, rows, shard, invert_transform=invert_transform, get_wmad=get_wmad, rng=rng,
                               engine=engine, device=device, return_seconds=return_seconds, impute_order=impute_order,
                               rejection_threshold=rejection_threshold, max_trials=max_trials, compute=compute)
    enc, norms, raw, full, scaled, oob = _scaled_instances(imp, rows, mask)
    lab = np.array([imp.class_map[c] for c in np.asarray(imp.y_test)[rows].tolist()], dtype=np.int32)
    order = np.argsort(lab, kind="stable")                      # the engine wants class-sorted data sets
    phi = enc.encode(scaled[order], imp.opts.d)
    cx = np.iscomplexobj(phi) or np.iscomplexobj(imp.x_guess_range.xvals_enc) or any(np.iscomplexobj(t) for t in imp.mps)
    phi = np.ascontiguousarray(phi, dtype=np.complex128 if cx else np.float64)
    m8 = np.ascontiguousarray(mask[order], dtype=np.uint8)
    N, T = m8.shape
    u = None
    code, trials, thr, basis = METHODS[method], 1, 0.0, 1
    if method == "ITS":
        if rejection_threshold is not None:
            code, trials, thr = 4, int(max_trials), float(rejection_threshold)
        u = np.ascontiguousarray((rng or np.random.default_rng()).uniform(0.0, 1.0, (N, T, trials)))
    if method == "mean":
        codes = {"Legendre_Norm": 0, "Legendre_No_Norm": 1, "Fourier": 2, "Stoudenmire": 3, "Sahand": 4, "Uniform": 5}    # MPST_BASIS_*
        if enc.name not in codes:
            raise NotImplementedError("method 'mean' re-encodes the expectation value on the device: closed-form bases only "
                                      f"({', '.join(codes)}), not {enc.name}")
        basis = codes[enc.name]
    own = engine is None
    eng = engine or SweepEngine(device)
    try:
        kw = dict(order=ORDERS[impute_order], max_trials=trials, rejection_threshold=thr, mean_basis=basis)
        if cx or compute != "f64":
            x, err, secs = eng.impute_model(imp.mps, phi, lab[order], m8, imp.x_guess_range.xvals, imp.x_guess_range.xvals_enc, code,
                                            get_wmad, u, compute=compute, **kw)
        else:
            Cn = int(imp.mps[-1].shape[3])
            eng.set_options(**engine_options(imp.opts))
            eng.set_dataset(1, phi, lab[order], Cn)
            eng.set_mps(imp.mps)
            x, err, secs = eng.impute(1, m8, imp.x_guess_range.xvals, imp.x_guess_range.xvals_enc, code, get_wmad, u, **kw)
    finally:
        if own:
            eng.close()
    inv = np.empty_like(order)
    inv[order] = np.arange(len(order))
    x, err = x[inv], err[inv]
    ts = np.where(mask, x, scaled)                               # x_samps: known values as given, imputed ones filled in
    pred = err if (method in ("median", "mean") and get_wmad) else None
    if invert_transform:
        hi = None
        if pred is not None:
            hi = invert_test_transform(ts + pred, oob, norms, imp.opts, enc.range)     # :339-341: add, invert, subtract
        ts = invert_test_transform(ts, oob, norms, imp.opts, enc.range)
        if pred is not None:
            pred = hi - ts
    out = (ts, pred)
    return out + (secs,) if return_seconds else out


def _impute_sharded(imp, mask, method, rows, shard, return_seconds=False, rng=None, **kw):
    """Rows i with i % world == rank on every rank (the classes stay balanced), results gathered with the host-side
    process group.  The uniform numbers of the sampling method (ITS) come from one seed shared by all ranks (rank 0's draw
    from `rng`, broadcast) and one stream per rank derived from it - reproducible for a given `rng` state and world size, not
    the single-process stream."""
    import torch.distributed as dist
    mine = np.arange(shard.rank, len(rows), shard.world)
    shard_rng = None
    if method == "ITS":
        # one seed for the whole call - rank 0's draw (from `rng` if given), broadcast over the host-side group - and one
        # stream per rank derived from it: a sharded run with the same `rng` state and world size repeats itself
        seed = [int((rng or np.random.default_rng()).integers(0, 2 ** 62))]
        dist.broadcast_object_list(seed, src=0, group=shard.group)
        shard_rng = np.random.default_rng([seed[0], shard.rank])
    ts = pred = None
    secs = 0.0
    if len(mine):
        out = impute_dataset(imp, mask[mine], method, rows=np.asarray(rows)[mine], return_seconds=True, rng=shard_rng, **kw)
        ts, pred, secs = out
    parts = [None] * shard.world
    dist.all_gather_object(parts, (mine, ts, pred, secs), group=shard.group)
    T = mask.shape[1]
    full = np.zeros((len(rows), T))
    perr = np.zeros((len(rows), T))
    have_err = False
    for idx, t, e, _ in parts:
        if t is None:
            continue
        full[idx] = t
        if e is not None:
            perr[idx] = e
            have_err = True
    out = (full, perr if have_err else None)
    return out + (max(p[3] for p in parts),) if return_seconds else out


def kNN_impute(imp: ImputationProblem, class_, instance: int, missing_sites, k: int = 1):
    """kNN_impute (imputation.jl:215-254): the k training series of the class closest (MSE over the known sites)."""
    cl = np.flatnonzero(np.asarray(imp.y_test) == class_)
    target = imp.X_test[cl[instance]]
    known = np.setdiff1d(np.arange(imp.X_test.shape[1]), np.asarray(missing_sites))
    c_inds = np.flatnonzero(np.asarray(imp.y_train) == class_)
    mses = np.mean((imp.X_train[c_inds][:, known] - target[known]) ** 2, axis=1)
    return [imp.X_train[c_inds[j]].copy() for j in np.argsort(mses, kind="stable")[:k]]


def mae(forecast, actual):
    return float(np.mean(np.abs(np.asarray(forecast) - np.asarray(actual))))


def mape(forecast, actual):
    return float(np.mean(np.abs(np.asarray(actual) - np.asarray(forecast)) / np.abs(np.asarray(actual))))


def MPS_impute(imp: ImputationProblem, class_, instance: int, missing_sites, method: str = "median", invert_transform: bool = True,
               impute_order: str = "forwards", NN_baseline: bool = True, n_baselines: int = 1, get_metrics: bool = True,
               engine: Optional[SweepEngine] = None, device: int = 0, **kw):
    """MPS_impute(imp, class, instance, missing_sites, method) (imputation.jl:467-550) without the plots:
    returns (ts, pred_err, target, metrics) with ``ts`` / ``pred_err`` lists of series as in the reference."""
    missing_sites = np.asarray(missing_sites, dtype=np.int64)
    cl = np.flatnonzero(np.asarray(imp.y_test) == class_)
    row = int(cl[instance])
    T = imp.X_test.shape[1]
    mask = np.zeros((1, T), dtype=bool)
    mask[0, missing_sites] = True
    if method == "kNearestNeighbour":
        ts, pred = kNN_impute(imp, class_, instance, missing_sites, k=kw.get("k", 1)), [None]
        target = imp.X_test[row]
    elif method == "flatBaseline":
        t = imp.X_test[row].copy()
        t[missing_sites] = np.mean(imp.X_train)
        ts, pred, target = [t], [None], imp.X_test[row]
    else:
        t, e = impute_dataset(imp, mask, method, rows=[row], invert_transform=invert_transform, engine=engine, device=device,
                              impute_order=impute_order,
                              **{k: v for k, v in kw.items() if k in ("get_wmad", "rng", "rejection_threshold", "max_trials")})
        ts, pred = [t[0]], [None if e is None else e[0]]
        if invert_transform:
            target = imp.X_test[row]
        else:
            target = _scaled_instances(imp, [row], mask)[3][0]
    metrics = []
    if get_metrics:
        for t in ts:
            metrics.append({"MAE": mae(t[missing_sites], target[missing_sites]), "MAPE": mape(t[missing_sites], target[missing_sites])})
    if NN_baseline and method not in ("kNearestNeighbour",):
        nn = kNN_impute(imp, class_, instance, missing_sites, k=n_baselines)
        if get_metrics:
            metrics[0]["NN_MAE"] = mae(nn[0][missing_sites], imp.X_test[row][missing_sites])
            metrics[0]["NN_MAPE"] = mape(nn[0][missing_sites], imp.X_test[row][missing_sites])
    return ts, pred, target, metrics
