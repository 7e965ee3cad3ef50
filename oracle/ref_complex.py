"""CPU restatement (NumPy) of MPSTime.jl's training sweep for COMPLEX encodings and reduced precision.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package imports this file; it is the checker for tests/ and the
tolerance study of bench.py.

PARITY UNPINNED.  The reference trains complex encodings (Fourier, Sahand, Stoudenmire; ``opts.dtype <: Complex``) only
through its legacy ITensor engine - the array engine raises for them (src/Training/RealRealHighDimension.jl:461-466 and the
Float64-only ``Ref`` of src/Training/loss_functions.jl:203-217) - so the semantics restated here are those of
    src/legacy_itensor/loss_functions.jl:433-462   yhat_phitilde: phi~ = conj(ps_l) (x) LE (x) conj(ps_r) (x) RE,
                                                   ``yhat = BT * phi_tilde  # NOT a complex inner product``
    src/legacy_itensor/loss_functions.jl:468-491   KLD_iter!: loss = -log(abs2(yhat)), phit_scaled += phi~/yhat
    src/legacy_itensor/loss_functions.jl:494-596   Loss_Grad_KLD: grad[c] = -conj(sum_i phi~_i/yhat_i)/N (or /n_c)
    src/legacy_itensor/loss_functions.jl:599-640   MSE_iter / Loss_Grad_MSE: 0.5 sum_c |yhat_c - y_c|^2, (yhat - y) conj(phi~)
    src/legacy_itensor/loss_functions.jl:105-175   custGD / TSGO (norm = Frobenius norm of the complex tensor)
    src/legacy_itensor/RealRealLegacyITensor.jl:2-47    construct_caches_IT: LE[j] = LE[j-1] * (conj(ps[j]) * W[j])
    src/legacy_itensor/RealRealLegacyITensor.jl:49-82   update_caches_IT!
    src/legacy_itensor/RealRealLegacyITensor.jl:84-142  decomposeBT_IT: ITensors.svd, U*S / V placement
No output of the Julia reference exists for this path in this container (no Julia, no network): the restatement is
cross-checked against an independent formulation instead (full-chain contraction + finite differences / complex autograd,
tests/test_oracle_complex.py), exactly as oracle/ref_numpy.py is for the real path.

The sample loop of the reference (mapreduce over product states, one ITensor contraction per sample) is restated as
batched matrix products over the same sums; the real-dtype specialisation of these functions reproduces
oracle/ref_numpy.py's per-sample restatement of the array engine (checked in tests/test_oracle_complex.py), which is how the
two engines of the reference relate (test/classification.jl:24).

Conventions are those of oracle/ref_numpy.py (0-based sites, W[j] of shape (Dl, d, Dr[, C]), data (N, T, d)).
``dtype`` may be float32 / float64 / complex64 / complex128: every array is kept in that type (the reference's
``opts.dtype``, RealRealHighDimension.jl:442); only the SVD can be asked to run in double precision (``svd_double``), which
is what the device engine does (Gram matrix and eigensolver in fp64).
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg

from . import ref_numpy as R
from .ref_numpy import EncodedSet, SweepOptions, truncate_spectrum  # noqa: F401  (re-exported)

construct_caches = R.construct_caches          # same formula as construct_caches_IT (conj on the product state only)
update_caches = R.update_caches                # same formula as update_caches_IT!
flatten_bt = R.flatten_bt
unflatten_bt = R.unflatten_bt
contract_mps = R.contract_mps                  # summary.jl:4-14 (conj(PS.pstate[i]))
mse_loss_acc = R.mse_loss_acc                  # summary.jl:33-114 (abs2 / abs: complex-safe)
classify = R.classify
mps_norm = R.mps_norm
normalize_mps = R.normalize_mps


def khatri_rao_operands(LEp, REp, phi_l, phi_r):
    """X_i[(a,s)] = LE_i[a] conj(ps_i[lid][s]),  Y_i[(t,b)] = conj(ps_i[rid][t]) RE_i[b]:
    phi~_i = X_i (x) Y_i  (legacy loss_functions.jl:441-456; the environments already hold the conjugated states of
    the sites they cover and are NOT conjugated again)."""
    N = phi_l.shape[0]
    le = np.ones((N, 1), dtype=phi_l.dtype) if LEp is None else LEp
    re = np.ones((N, 1), dtype=phi_l.dtype) if REp is None else REp
    X = (le[:, :, None] * np.conj(phi_l)[:, None, :]).reshape(N, -1)          # (N, Dl*d), s fastest
    Y = (np.conj(phi_r)[:, :, None] * re[:, None, :]).reshape(N, -1)          # (N, d*Dr), b fastest
    return X, Y


def _bmat(bt5, c):
    """B_c[(a,s), (t,b)] of the bond tensor (s_l, a, s_r, b, c)."""
    d_l, Da, d_r, Db, _ = bt5.shape
    return bt5[..., c].transpose(1, 0, 2, 3).reshape(Da * d_l, d_r * Db)


def _unbmat(G, shape4):
    d_l, Da, d_r, Db = shape4
    return G.reshape(Da, d_l, d_r, Db).transpose(1, 0, 2, 3)


def loss_grad(bt5, LEp, REp, data: EncodedSet, lid, rid, loss="KLD", train_separate=False):
    """Loss_Grad_KLD (legacy loss_functions.jl:494-596) / Loss_Grad_MSE (:625-640) on the bond tensor (s_l,a,s_r,b,c).
    Returns (loss, grad) with grad in the same 5-index layout and dtype as bt5."""
    phi = data.phi
    N = phi.shape[0]
    C = bt5.shape[4]
    X, Y = khatri_rao_operands(LEp, REp, phi[:, lid, :], phi[:, rid, :])
    grad = np.zeros_like(bt5)
    rdt = np.zeros(1, dtype=bt5.dtype).real.dtype
    if loss == "KLD":
        losses = rdt.type(0)
        i0 = 0
        for ci, cn in enumerate(data.class_distribution):
            cn = int(cn)
            sl = slice(i0, i0 + cn)
            B = _bmat(bt5, ci)
            yhat = np.einsum("ix,ix->i", X[sl] @ B, Y[sl])                      # yhat = BT * phi_tilde  (:458)
            lc = np.sum(-np.log(np.abs(yhat) ** 2), dtype=rdt)                  # :479
            ps = (X[sl] / yhat[:, None]).T @ Y[sl]                              # phit_scaled += phi~/f_ln  (:483)
            if train_separate:
                losses += lc / rdt.type(cn)                                     # :534
                grad[..., ci] = _unbmat(-np.conj(ps) / rdt.type(cn), bt5.shape[:4])   # :535
            else:
                losses += lc                                                    # :574
                grad[..., ci] = _unbmat(-np.conj(ps) / rdt.type(N), bt5.shape[:4])    # :575, :591
            i0 += cn
        if not train_separate:
            losses = losses / rdt.type(N)                                       # :590
        return losses, grad
    if loss == "MSE":
        if train_separate:
            raise NotImplementedError("no Loss_Grad_MSE method for TrainSeparate{true}")
        yh = np.stack([np.einsum("ix,ix->i", X @ _bmat(bt5, c), Y) for c in range(C)], axis=1)    # (N, C)
        y = np.zeros((N, C), dtype=rdt)
        y[np.arange(N), data.label_index] = 1
        diff = yh - y
        losses = rdt.type(0.5) * np.sum(np.abs(diff) ** 2, dtype=rdt) / rdt.type(N)    # :611-613, :636
        for c in range(C):
            g = (np.conj(X) * diff[:, c][:, None]).T @ np.conj(Y)               # (yhat - y) * conj(phi_tilde)  :616
            grad[..., c] = _unbmat(g / rdt.type(N), bt5.shape[:4])              # :637
        return losses, grad
    raise ValueError(loss)


def apply_update(bt5, LEp, REp, lid, rid, data, opts: SweepOptions, trace=None):
    """apply_update_IT with custGD / TSGO (legacy loss_functions.jl:105-175,178-262)."""
    bt = bt5.copy()
    rdt = np.zeros(1, dtype=bt.dtype).real.dtype
    if opts.rescale[0]:
        bt = bt / np.linalg.norm(bt).astype(rdt)                                # :194-196
    fl = opts.bbopt.upper()
    if fl not in ("GD", "TSGO"):
        raise RuntimeError("Optim/OptimKit based solvers are outside the engine")
    eta = rdt.type(opts.eta)
    for it in range(opts.update_iters):
        loss, grad = loss_grad(bt, LEp, REp, data, lid, rid, opts.loss_grad, opts.train_classes_separately)
        if trace is not None and it == 0:
            trace["loss"] = float(loss)
            trace["grad_norm"] = float(np.linalg.norm(grad))
        if fl == "GD":
            bt = bt - eta * grad                                                # :123
        else:
            bt = bt - eta * (grad / np.linalg.norm(grad).astype(rdt))           # :151
    if opts.rescale[1]:
        bt = bt / np.linalg.norm(bt).astype(rdt)                                # :251-253
    return bt


def decompose_bt(bt5, chi_max, cutoff, going_left=True, svd_double=False):
    """decomposeBT_IT, RealRealLegacyITensor.jl:84-142.  ITensors' ``U, S, V = svd(A, rowinds)`` returns A = U*S*V, i.e.
    the ITensor V holds the conjugated right singular vectors (V^H as a matrix) - scipy's ``Vh``."""
    d_l, Da, d_r, Db, C = bt5.shape
    dt = bt5.dtype
    if going_left:
        M = bt5.transpose(1, 4, 0, 2, 3).reshape(Da * C * d_l, d_r * Db)        # rows (a, c, s_l) | cols (s_r, b)  :103-107
    else:
        M = bt5.transpose(3, 4, 2, 0, 1).reshape(Db * C * d_r, d_l * Da)        # rows (b, c, s_r) | cols (s_l, a)  :122-126
    if svd_double:
        M = M.astype(np.complex128 if np.iscomplexobj(M) else np.float64)
    U, S, Vh = scipy.linalg.svd(M, full_matrices=False, lapack_driver="gesdd")
    n = truncate_spectrum(S, chi_max, cutoff)
    U, S, Vh = U[:, :n], S[:n], Vh[:n]
    if going_left:
        left = (U * S).reshape(Da, C, d_l, n).transpose(0, 2, 3, 1).astype(dt)  # U*S carries the label  :109
        right = Vh.reshape(n, d_r, Db).astype(dt)                               # V  :110
    else:
        right = (U * S).reshape(Db, C, d_r, n).transpose(3, 2, 0, 1).astype(dt)  # V*S carries the label  :129
        left = Vh.reshape(n, d_l, Da).transpose(2, 1, 0).astype(dt)             # U  :128
    return left, right, S


def bond_step(W, LE, RE, lid, data, opts: SweepOptions, going_left, trace=None, svd_double=False):
    """One bond of fitMPS_IT's sweep body (RealRealLegacyITensor.jl: BT = W[lid]*W[rid]; apply_update_IT; decomposeBT_IT;
    update_caches_IT!) - the same sequence as RealRealHighDimension.jl:733-762 / :777-801."""
    rid = lid + 1
    T = len(W)
    bt, shape4 = flatten_bt(W[lid], W[rid])
    bt5 = unflatten_bt(bt, shape4)
    LEp = LE[lid - 1] if lid > 0 else None
    REp = RE[rid + 1] if rid < T - 1 else None
    bt_new = apply_update(bt5, LEp, REp, lid, rid, data, opts, trace)
    lsn, rsn, S = decompose_bt(bt_new, opts.chi_max, opts.cutoff, going_left, svd_double)
    update_caches(lsn, rsn, LE, RE, lid, rid, data.phi, going_left)
    W[lid], W[rid] = lsn, rsn
    if trace is not None:
        trace["S"] = np.asarray(S, dtype=np.float64)
        trace["bt_new_norm"] = float(np.linalg.norm(bt_new))
        trace["chi"] = len(S)
        trace["bt_new"] = bt_new
    return trace


def sweep(W, data: EncodedSet, opts: SweepOptions, LE=None, RE=None, record=None, svd_double=False):
    """One full sweep: backward half-sweep, cache rebuild, forward half-sweep, cache rebuild."""
    T = len(W)
    if LE is None:
        LE, RE = construct_caches(W, data.phi, going_left=True)
    for j in range(T - 2, -1, -1):
        tr = {} if record is not None else None
        bond_step(W, LE, RE, j, data, opts, True, tr, svd_double)
        if record is not None:
            tr.update(lid=j, going_left=True)
            record.append(tr)
    LE, RE = construct_caches(W, data.phi, going_left=False)
    for j in range(0, T - 1):
        tr = {} if record is not None else None
        bond_step(W, LE, RE, j, data, opts, False, tr, svd_double)
        if record is not None:
            tr.update(lid=j, going_left=False)
            record.append(tr)
    LE, RE = construct_caches(W, data.phi, going_left=True)
    return LE, RE


def cast_problem(data: EncodedSet, W, dtype):
    """The data set and MPS in the element type ``dtype`` (opts.dtype)."""
    dt = np.dtype(dtype)
    if not np.issubdtype(dt, np.complexfloating) and np.iscomplexobj(data.phi):
        raise ValueError("Using a complex valued encoding but the MPS is real")         # RealRealHighDimension.jl:462-464
    ds = EncodedSet(data.phi.astype(dt), data.label_index, data.class_distribution, data.labels, data.original_data)
    return ds, [t.astype(dt) for t in W]


def make_problem(N, T, d, chi_init, C, seed=0, dtype=np.complex128, encoding="fourier", balanced=True):
    """Seeded data set + initial MPS in the given element type (Fourier basis for complex types, Legendre for real)."""
    rng = np.random.default_rng(seed)
    Xr = rng.uniform(-1, 1, (N, T))
    if balanced:
        y = np.arange(N) % C
    else:
        y = rng.integers(0, C, N)
        y[:C] = np.arange(C)
    rng.shuffle(y)
    cx = np.issubdtype(np.dtype(dtype), np.complexfloating)
    enc = (lambda x: R.fourier_encode(x, d)) if (cx and encoding == "fourier") else (lambda x: R.legendre_encode(x, d))
    ds = R.encode_dataset(Xr, Xr, y, enc, (-1, 1))
    W = R.random_mps(T, d, chi_init, C, np.random.default_rng(seed + 1000), dtype=np.complex128 if cx else np.float64)
    return cast_problem(ds, W, dtype)


def full_chain_loss(W, bt5, lid, data: EncodedSet, loss="KLD", train_separate=False):
    """Independent formulation (no caches, no phi-tilde): the loss as a function of the bond tensor by contracting the
    whole chain left to right, in double precision.  Used with finite differences to check ``loss_grad``:
    for a real-valued L of a complex tensor B, dL = 2 Re(sum conj(g) dB) with g = dL/d(conj B) - and the reference's
    gradient IS that conjugate Wirtinger derivative: -conj(phi~/yhat) (legacy loss_functions.jl:482)."""
    phi = data.phi.astype(np.complex128 if np.iscomplexobj(data.phi) else np.float64)
    T = len(W)
    N = phi.shape[0]
    cdt = np.result_type(phi.dtype, bt5.dtype, np.float64)
    left = np.ones((N, 1), dtype=cdt)
    for j in range(lid):
        left = np.einsum("ia,is,ask->ik", left, np.conj(phi[:, j, :]), W[j].astype(cdt))
    right = np.ones((N, 1), dtype=cdt)
    for j in range(T - 1, lid + 1, -1):
        right = np.einsum("ib,is,ksb->ik", right, np.conj(phi[:, j, :]), W[j].astype(cdt))
    yhat = np.einsum("ia,is,it,ib,satbc->ic", left, np.conj(phi[:, lid, :]), np.conj(phi[:, lid + 1, :]), right, bt5.astype(cdt))
    C = yhat.shape[1]
    idx = np.asarray(data.label_index)
    if loss == "KLD":
        per = -np.log(np.abs(yhat[np.arange(N), idx]) ** 2)
        if train_separate:
            return float(np.sum(per / np.asarray(data.class_distribution, dtype=float)[idx])), yhat
        return float(per.mean()), yhat
    y = np.zeros((N, C))
    y[np.arange(N), idx] = 1.0
    return float(0.5 * np.sum(np.abs(yhat - y) ** 2) / N), yhat
