"""CPU restatement (NumPy/SciPy) of the MPSTime.jl training sweep.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package imports this file;
it is the checker for tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py.

PARITY PARTIALLY PINNED.  The Julia/ITensors reference cannot be executed in
the build container (no Julia, no network) and none of its own known-answer
tests for this path is evaluable without it (they need the ItalyPowerDemand
download plus Julia's seeded ``random_mps``).  What IS pinned against
reference-produced tensors (tests/golden/ref_ecg200_trained_mps.npz, extracted
from the reference's serialised TrainedMPS test/Data/ecg200/mps_saves/
test_dataset.jld2 by tests/golden/extract_jld2_fixture.py; checked in
tests/test_reference_fixture.py):
  * transform_train_data + legendre_encode reproduce the reference's stored
    product states to 3e-15 (rows A14/A15 of SURVEY.md section 8);
  * the container conventions, the canonical form a finished fitMPS leaves
    (left-orthonormal sites, label on the last site, unit norm) and
    contract_mps / mse_loss_acc / classify on that MPS (100 % train accuracy,
    rows A12/A13/A17/A18).
The sweep's optimiser trajectory itself (loss/gradient, update, truncated SVD:
rows A1-A11) is still PARITY UNPINNED: it follows the cited source lines
literally and is cross-checked against an independent formulation
(oracle/naive.py), but has never been compared with output of the reference.

Every function cites the reference file:line it follows (paths relative to
/root/reference/).

Conventions (0-based sites here, 1-based in the reference):
  * A site tensor is an ndarray ``W[j]`` of shape (Dl, d, Dr) or, when it
    carries the label index, (Dl, d, Dr, C).  Dl = 1 on site 0, Dr = 1 on
    site T-1.
  * A product state is an ndarray (T, d); a data set is (N, T, d) sorted by
    class with ``label_index`` (N,) 0-based and ``class_distribution`` (C,).
  * Environment caches are arrays LE[j] / RE[j] of shape (N, D).
  * The flat bond-tensor layout is the reference's: column-major over
    (s_lid, l_{lid-1}, s_rid, l_rid), one column per class
    (src/Training/RealRealHighDimension.jl:221-238 with the kron order of
    src/Training/loss_functions.jl:248-262).
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass, field

import numpy as np
import scipy.linalg

# --------------------------------------------------------------------------
# containers  (src/Structs/structs.jl:2-33)
# --------------------------------------------------------------------------


@dataclass
class EncodedSet:
    """EncodedTimeSeriesSet + PState fields, as arrays (structs.jl:12-33)."""

    phi: np.ndarray                 # (N, T, d) encoded states, class-sorted
    label_index: np.ndarray         # (N,) int, 0-based class slot, nondecreasing
    class_distribution: np.ndarray  # (C,) counts per class slot
    labels: np.ndarray | None = None  # original labels (sorted), optional
    original_data: np.ndarray | None = None

    @property
    def N(self):
        return self.phi.shape[0]


@dataclass
class SweepOptions:
    """The MPSOptions fields consumed on the sweep path
    (src/Structs/options.jl:106-143; RealRealHighDimension.jl:591-592,606,716-722)."""

    nsweeps: int = 10
    chi_max: int = 25
    eta: float = 0.01
    cutoff: float = 1e-10
    update_iters: int = 1
    loss_grad: str = "KLD"          # "KLD" | "MSE"
    bbopt: str = "TSGO"             # "TSGO" | "GD"
    rescale: tuple = (False, True)
    train_classes_separately: bool = False
    exit_early: bool = False
    log_level: int = 3


# --------------------------------------------------------------------------
# caches  (RealRealHighDimension.jl:45-103)
# --------------------------------------------------------------------------

def construct_caches(W, phi, going_left=True):
    """construct_caches, RealRealHighDimension.jl:45-103.

    going_left=True fills LE[0..T-2]; going_left=False fills RE[T-1..1].
    ``ps' * sl`` is an adjoint product => conj on the product state (:69,:75).
    Both caches are (re)allocated on every call as in the reference (:62-63).
    """
    T = len(W)
    N = phi.shape[0]
    LE = [None] * T
    RE = [None] * T
    if going_left:
        w1 = W[0].reshape(W[0].shape[1], W[0].shape[2])           # matrix(W[1]) (s, l1)
        LE[0] = np.conj(phi[:, 0, :]) @ w1                         # :68-70
        for j in range(1, T - 1):                                  # :72-77
            t = W[j]                                               # (a, s, k)
            # LE[j,i][k] = sum_{s,a} conj(ps[s]) t[a,s,k] LE[j-1,i][a]
            LE[j] = np.einsum("is,ask,ia->ik", np.conj(phi[:, j, :]), t, LE[j - 1])
    else:
        wT = W[T - 1]
        assert wT.ndim == 3, "label must not sit on the last site when building RE"
        wend = wT.reshape(wT.shape[0], wT.shape[1])                # (l_{T-1}, s)
        RE[T - 1] = np.conj(phi[:, T - 1, :]) @ wend.T             # :85-88
        for j in range(T - 2, 0, -1):                              # :90-98
            t = W[j]                                               # (k, s, b)
            RE[j] = np.einsum("is,ksb,ib->ik", np.conj(phi[:, j, :]), t, RE[j + 1])
    return LE, RE


def update_caches(left_new, right_new, LE, RE, lid, rid, phi, going_left=True):
    """update_caches!, RealRealHighDimension.jl:107-144."""
    T = phi.shape[1]
    if going_left:
        if rid == T - 1:                                           # :124-125
            m = right_new.reshape(right_new.shape[0], right_new.shape[1])  # (k, s)
            RE[T - 1] = np.conj(phi[:, rid, :]) @ m.T
        else:                                                      # :127
            RE[rid] = np.einsum("is,ksb,ib->ik", np.conj(phi[:, rid, :]), right_new, RE[rid + 1])
    else:
        if lid == 0:                                               # :134-136
            m = left_new.reshape(left_new.shape[1], left_new.shape[2])     # (s, k)
            LE[0] = np.conj(phi[:, 0, :]) @ m
        else:                                                      # :138
            LE[lid] = np.einsum("is,ask,ia->ik", np.conj(phi[:, lid, :]), left_new, LE[lid - 1])


# --------------------------------------------------------------------------
# bond tensor flatten / unflatten  (RealRealHighDimension.jl:221-243)
# --------------------------------------------------------------------------

def flatten_bt(Wl, Wr):
    """flatten_bt, RealRealHighDimension.jl:221-238.

    Returns bt (L, C) with L = d*Dl*d*Dr flattened column-major over
    (s_lid, l_{lid-1}, s_rid, l_rid) and the 4-index shape.
    The label may sit on either site (going_left: on the right site,
    going right: on the left site) - the contraction is the same.
    """
    if Wl.ndim == 4:
        bt5 = np.einsum("asmc,mtb->satbc", Wl, Wr)
    else:
        assert Wr.ndim == 4, "one of the two sites must carry the label"
        bt5 = np.einsum("asm,mtbc->satbc", Wl, Wr)
    shape4 = bt5.shape[:4]
    C = bt5.shape[4]
    bt = np.stack([bt5[..., c].reshape(-1, order="F") for c in range(C)], axis=1)
    return bt, shape4


def unflatten_bt(bt, shape4):
    """unflatten_bt, RealRealHighDimension.jl:240-243 -> (s_l, a, s_r, b, c)."""
    C = bt.shape[1]
    return np.stack([bt[:, c].reshape(shape4, order="F") for c in range(C)], axis=-1)


# --------------------------------------------------------------------------
# fused phi-tilde / yhat / gradient loops  (loss_functions.jl:193-296,435-531)
# --------------------------------------------------------------------------

def kron_conj2(x1, x2):
    """kron_conj2, loss_functions.jl:193-200: out[j + l2*(i-1)] = conj(x1[i]*x2[j])."""
    return np.conj(np.outer(x1, x2)).reshape(-1)


def _phi_tilde(LEp, REp, ps, lid, rid, T):
    """The phi-tilde vector of one sample in the flat layout; site-case
    dispatch of yhat_phitilde_KLD!!, loss_functions.jl:279-295 (KLD) and
    :514-530 (MSE) - identical vectors in both.

    Last kron argument is the fastest index (idx = j + lb*(i-1))."""
    if lid == 0:
        if rid != T - 1:
            # kron_scaleadd_firstsite_KLD!(REP[rid+1], ps[rid], ps[lid]) :283,:217-224
            x2a = kron_conj2(ps[rid], ps[lid])
            return np.outer(REp[rid + 1], x2a).reshape(-1)         # phi = x1[i]*x2a[j]
        # two-site MPS :285 -> 2-vector form :203-215
        return kron_conj2(ps[rid], ps[lid])
    if rid == T - 1:
        # kron_scaleadd_KLD!(ps[rid], LEP[lid-1], ps[lid]) :290,:231-240
        x2a = kron_conj2(LEp[lid - 1], ps[lid])
        return np.outer(np.conj(ps[rid]), x2a).reshape(-1)
    # bulk :294,:248-262
    xa = kron_conj2(REp[rid + 1], ps[rid])
    xb = kron_conj2(LEp[lid - 1], ps[lid])
    return np.outer(xa, xb).reshape(-1)


def loss_grad_KLD(bt, LE, RE, data: EncodedSet, lid, rid, train_separate=False):
    """Loss_Grad_KLD, loss_functions.jl:322-379 (TrainSeparate{false}) and
    :383-432 ({true}), with the one-sample-late scaled accumulation of
    kron_scaleadd_KLD! (:248-262): k += kprev/scale while phi_i, yhat_i are
    formed; flushed at :367 / :424."""
    phi = data.phi
    N, T = phi.shape[0], phi.shape[1]
    L, C = bt.shape
    losses = 0.0
    phit_scaled = np.zeros_like(bt)
    i_prev = 0
    for ci, cn in enumerate(data.class_distribution):
        yhat = 1.0                                                 # :350
        phit_prev = np.zeros(L, dtype=bt.dtype)                    # :351
        k = phit_scaled[:, ci]
        loss = 0.0
        for i in range(i_prev, i_prev + int(cn)):                  # mapreduce :353-364
            LEp = [None if a is None else a[i] for a in LE]
            REp = [None if a is None else a[i] for a in RE]
            ph = _phi_tilde(LEp, REp, phi[i], lid, rid, T)
            scale = yhat
            k += phit_prev / scale                                 # :258
            yhat = float(np.real(np.dot(bt[:, ci], ph)))           # :257 (Ref{Float64})
            phit_prev = ph                                         # :259
            loss += -math.log(abs(yhat) ** 2)                      # KLD_iter! :318
        if train_separate:
            losses += loss / cn                                    # :423
            phit_scaled[:, ci] = -np.conj(k + phit_prev / yhat) / cn   # :424
        else:
            losses += loss                                         # :366
            phit_scaled[:, ci] = -np.conj(k + phit_prev / yhat) / N    # :367
        i_prev += int(cn)
    if not train_separate:
        losses /= N                                                # :371
    return losses, phit_scaled


def loss_grad_MSE(bt, LE, RE, data: EncodedSet, lid, rid, train_separate=False):
    """Loss_Grad_MSE, loss_functions.jl:561-619 (only TrainSeparate{false}
    exists).  For every class column all N samples are visited; mask=1 for
    members of the class (:590,:608)."""
    if train_separate:
        raise NotImplementedError("reference has no Loss_Grad_MSE for TrainSeparate{true}")
    phi = data.phi
    N, T = phi.shape[0], phi.shape[1]
    L, C = bt.shape
    losses = 0.0
    phit_scaled = np.zeros_like(bt)
    yprev = 0.0                                                    # :582
    class_mask = np.zeros(N)
    i_prev = 0
    for ci, cn in enumerate(data.class_distribution):
        yhat = 1.0                                                 # :587
        phit_prev = np.zeros(L, dtype=bt.dtype)                    # :588
        class_mask[i_prev:i_prev + int(cn)] = 1.0                  # :590
        k = phit_scaled[:, ci]
        loss = 0.0
        for i in range(N):                                         # :592-605
            LEp = [None if a is None else a[i] for a in LE]
            REp = [None if a is None else a[i] for a in RE]
            ph = _phi_tilde(LEp, REp, phi[i], lid, rid, T)
            scale = yhat - yprev                                   # :481
            k += phit_prev * scale                                 # :489
            yhat = float(np.real(np.dot(bt[:, ci], ph)))
            phit_prev = ph
            loss += 0.5 * abs(yhat - class_mask[i]) ** 2           # MSE_iter! :554
            yprev = class_mask[i]                                  # :555
        losses += loss                                             # :607
        phit_scaled[:, ci] = (k + np.conj(phit_prev) * (yhat - yprev)) / N   # :608
        class_mask[i_prev:i_prev + int(cn)] = 0.0
        i_prev += int(cn)
    return losses / N, phit_scaled


LOSS_GRADS = {"KLD": loss_grad_KLD, "MSE": loss_grad_MSE}

# --------------------------------------------------------------------------
# optimisers  (loss_functions.jl:27-188)
# --------------------------------------------------------------------------


def apply_update(bt_init, LE, RE, lid, rid, data, opts: SweepOptions, trace=None):
    """apply_update, loss_functions.jl:88-188, with custGD (:27-57) and TSGO (:59-86)."""
    lg = LOSS_GRADS[opts.loss_grad]
    bt = bt_init.copy()
    if opts.rescale[0]:
        bt /= np.linalg.norm(bt)                                   # :109-111
    fl = opts.bbopt.upper()
    if fl not in ("GD", "TSGO"):
        # :166-170
        raise RuntimeError("Optim/OptimKit based solvers currently unimplemented for this version, "
                           "set 'use_legacy_ITensor=true' in MPSOptions to enable")
    for it in range(opts.update_iters):
        loss, grad = lg(bt, LE, RE, data, lid, rid, opts.train_classes_separately)
        if trace is not None and it == 0:
            trace["loss"] = loss
            trace["grad_norm"] = float(np.linalg.norm(grad))
        if fl == "GD":
            bt = bt - opts.eta * grad                              # :49
        else:
            bt = bt - opts.eta * (grad / np.linalg.norm(grad))     # :79
    if opts.rescale[1]:
        bt = bt / np.linalg.norm(bt)                               # :177-179
    return bt


# --------------------------------------------------------------------------
# SVD split  (RealRealHighDimension.jl:146-203 -> ITensors.svd / NDTensors truncate!)
# --------------------------------------------------------------------------

def truncate_spectrum(S, maxdim, cutoff, mindim=1):
    """NDTensors ``truncate!`` as reached through ITensors.svd(...; maxdim, cutoff)
    with use_relative_cutoff=true, use_absolute_cutoff=false (ITensors =0.6.22 /
    NDTensors 0.3.74, Manifest.toml:932-936,1537-1541 - not vendored; restated
    from the published algorithm).  Returns n kept."""
    P = np.asarray(S, dtype=float) ** 2
    n = len(P)
    if n == 1:
        return 1
    truncerr = 0.0
    while n > maxdim:
        truncerr += P[n - 1]
        n -= 1
    scale = P.sum()
    if scale == 0.0:
        scale = 1.0
    while (truncerr + P[n - 1] <= cutoff * scale) and (n > mindim):
        truncerr += P[n - 1]
        n -= 1
    return n


def decompose_bt(bt5, lid, rid, chi_max, cutoff, going_left=True):
    """decomposeBT, RealRealHighDimension.jl:146-203.

    bt5 has axes (s_l, a, s_r, b, c).  LAPACK gesdd (= Julia's
    DivideAndConquer, the reference's default svd_alg).
    going_left: rows (a, c, s_l) | cols (s_r, b); left = U*S carries the label,
    right = V (:166-176).  going right: rows (b, c, s_r) | cols (s_l, a);
    right = V*S carries the label, left = U (:185-194).
    Returns (left_site, right_site, S_kept)."""
    d_l, Da, d_r, Db, C = bt5.shape
    if going_left:
        M = bt5.transpose(1, 4, 0, 2, 3).reshape(Da * C * d_l, d_r * Db)
        U, S, Vh = scipy.linalg.svd(M, full_matrices=False, lapack_driver="gesdd")
        n = truncate_spectrum(S, chi_max, cutoff)
        U, S, Vh = U[:, :n], S[:n], Vh[:n]
        left = (U * S).reshape(Da, C, d_l, n).transpose(0, 2, 3, 1)      # (a, s, k, c)
        right = Vh.reshape(n, d_r, Db)                                     # (k, s, b): ITensor V with U*S*V = BT
    else:
        M = bt5.transpose(3, 4, 2, 0, 1).reshape(Db * C * d_r, d_l * Da)
        U, S, Vh = scipy.linalg.svd(M, full_matrices=False, lapack_driver="gesdd")
        n = truncate_spectrum(S, chi_max, cutoff)
        U, S, Vh = U[:, :n], S[:n], Vh[:n]
        right = (U * S).reshape(Db, C, d_r, n).transpose(3, 2, 0, 1)     # (k, s, b, c)
        left = Vh.reshape(n, d_l, Da).transpose(2, 1, 0)                  # (a, s, k)
    return left, right, S


# --------------------------------------------------------------------------
# evaluation  (src/summary.jl:4-136)
# --------------------------------------------------------------------------

def contract_mps(W, phi):
    """contract_mps, summary.jl:4-14, batched over samples: returns yhat (N, C)."""
    N = phi.shape[0]
    res = np.ones((N, 1), dtype=W[0].dtype)
    lab = None
    for j, t in enumerate(W):
        p = np.conj(phi[:, j, :])
        if t.ndim == 4:
            assert lab is None
            lab = np.einsum("ia,is,askc->ikc", res, p, t)
            res = None
        elif lab is not None:
            lab = np.einsum("iac,is,ask->ikc", lab, p, t)
        else:
            res = np.einsum("ia,is,ask->ik", res, p, t)
    assert lab is not None, "MPS has no label index"
    return lab[:, 0, :]


def mse_loss_acc(W, data: EncodedSet, conf=False):
    """MSE_loss_acc / MSE_loss_acc_conf, summary.jl:33-114."""
    yhat = contract_mps(W, data.phi)
    N, C = yhat.shape
    y = np.zeros((N, C))
    y[np.arange(N), data.label_index] = 1.0
    mse = 0.5 * np.sum(np.abs(yhat - y) ** 2, axis=1)              # :45-47
    kld = -np.log(np.abs(yhat[np.arange(N), data.label_index]) ** 2)   # :48
    pred = np.argmax(np.abs(yhat), axis=1)                         # :52 (first max on ties)
    acc = np.mean(pred == data.label_index)
    out = (float(mse.mean()), float(kld.mean()), float(acc))
    if conf:
        cm = np.zeros((C, C), dtype=np.int64)
        np.add.at(cm, (data.label_index, pred), 1)                 # :96
        return out + (cm,)
    return out


def classify(W, phi):
    """classify, summary.jl:116-136: argmax |yhat|^2 -> class slot (0-based)."""
    return np.argmax(np.abs(contract_mps(W, phi)) ** 2, axis=1)


def mps_norm(W):
    """sqrt(<W|W>) including the label index (ITensors ``norm(::MPS)``)."""
    E = np.ones((1, 1), dtype=W[0].dtype)
    labelled = False
    for t in W:
        if t.ndim == 4:
            E = np.einsum("ab,askc,bsld->klcd", E, t, np.conj(t))
            E = np.einsum("klcc->kl", E)
            labelled = True
        else:
            E = np.einsum("ab,ask,bsl->kl", E, t, np.conj(t))
    return float(np.sqrt(np.real(E[0, 0])))


def normalize_mps(W):
    """ITensors ``normalize!(::MPS)`` (called at RealRealHighDimension.jl:852):
    every site tensor is divided by exp(lognorm/T)."""
    z = math.exp(math.log(mps_norm(W)) / len(W))
    return [t / z for t in W]


# --------------------------------------------------------------------------
# the sweep  (RealRealHighDimension.jl:587-890)
# --------------------------------------------------------------------------

def bond_step(W, LE, RE, lid, data, opts: SweepOptions, going_left, trace=None):
    """One bond of the sweep body, RealRealHighDimension.jl:733-762 / :777-801."""
    rid = lid + 1
    bt, shape4 = flatten_bt(W[lid], W[rid])                        # :733 / :777
    bt_new = apply_update(bt, LE, RE, lid, rid, data, opts, trace)  # :736 / :779
    bt5 = unflatten_bt(bt_new, shape4)                             # :753 / :796
    lsn, rsn, S = decompose_bt(bt5, lid, rid, opts.chi_max, opts.cutoff, going_left)  # :756 / :798
    update_caches(lsn, rsn, LE, RE, lid, rid, data.phi, going_left)  # :759 / :799
    W[lid], W[rid] = lsn, rsn                                      # :761-762 / :800-801
    if trace is not None:
        trace["S"] = S
        trace["bt_new_norm"] = float(np.linalg.norm(bt_new))
        trace["chi"] = len(S)
    return trace


def sweep(W, data: EncodedSet, opts: SweepOptions, LE=None, RE=None, record=None):
    """One full sweep = RealRealHighDimension.jl:727-808: backward half-sweep,
    cache rebuild (:770), forward half-sweep, cache rebuild (:804)."""
    T = len(W)
    if LE is None:
        LE, RE = construct_caches(W, data.phi, going_left=True)    # :631 / :804
    for j in range(T - 2, -1, -1):                                 # :731
        tr = {} if record is not None else None
        bond_step(W, LE, RE, j, data, opts, True, tr)
        if record is not None:
            tr.update(lid=j, going_left=True)
            record.append(tr)
    LE, RE = construct_caches(W, data.phi, going_left=False)       # :770
    for j in range(0, T - 1):                                      # :776
        tr = {} if record is not None else None
        bond_step(W, LE, RE, j, data, opts, False, tr)
        if record is not None:
            tr.update(lid=j, going_left=False)
            record.append(tr)
    LE, RE = construct_caches(W, data.phi, going_left=True)        # :804
    return LE, RE


def fit(W0, train: EncodedSet, test: EncodedSet | None, opts: SweepOptions, record=None):
    """fitMPS(W, training_states_meta, testing_states_meta, opts),
    RealRealHighDimension.jl:587-890.  Returns (W, training_information)."""
    assert np.all(np.diff(train.label_index) >= 0), "Training data must be sorted by class!"  # :624
    has_test = test is not None and test.N > 0
    if has_test:
        assert np.all(np.diff(test.label_index) >= 0), "Testing data must be sorted by class!"
    W = [t.copy() for t in W0]
    info = {k: [] for k in ("train_loss", "train_acc", "test_loss", "time_taken", "train_KL_div")}
    if has_test:
        info.update({k: [] for k in ("test_acc", "test_KL_div", "test_conf")})

    def log(time_taken):
        if opts.log_level <= 0:
            return None
        mse, kld, acc = mse_loss_acc(W, train)
        info["train_loss"].append(mse)
        info["train_acc"].append(acc)
        info["time_taken"].append(time_taken)
        info["train_KL_div"].append(kld)
        if has_test:
            tm, tk, ta, cf = mse_loss_acc(W, test, conf=True)
            info["test_loss"].append(tm)
            info["test_acc"].append(ta)
            info["test_KL_div"].append(tk)
            info["test_conf"].append(cf)
        return acc

    LE, RE = construct_caches(W, train.phi, going_left=True)       # :631
    log(0.0)                                                       # :657-689
    for its in range(opts.nsweeps):                                # :726
        start = time.time()
        rec = [] if record is not None else None
        LE, RE = sweep(W, train, opts, LE, RE, rec)
        elapsed = time.time() - start                              # :806-808
        if record is not None:
            record.append(rec)
        acc = log(elapsed)                                         # :813-845
        if opts.exit_early and acc == 1.0:                         # :847
            break
    W = normalize_mps(W)                                           # :852
    log(float("nan"))                                              # :854-885
    return W, info


# --------------------------------------------------------------------------
# inputs: preprocessing + encodings  (src/utils.jl:161-295, src/Encodings/bases.jl)
# --------------------------------------------------------------------------

def robust_sigmoid_fit(X):
    """Normalization.jl 0.7.3 RobustSigmoid fitted over the whole matrix:
    (median, iqr) (options.jl:72-77; call site utils.jl:174-176)."""
    med = float(np.median(X))
    q75, q25 = np.percentile(X, [75, 25])
    return med, float(q75 - q25)


def robust_sigmoid_apply(X, med, iqr):
    return 1.0 / (1.0 + np.exp(-(X - med) / (iqr / 1.35)))


def transform_train_data(X, sigmoid_transform=True, minmax=True, data_bounds=(0.0, 1.0), enc_range=(-1.0, 1.0)):
    """transform_train_data, utils.jl:161-199.  X is (N, T); orientation is
    irrelevant because both normalisations are fitted over the whole matrix."""
    norms = [None, None]
    Xs = np.array(X, dtype=float, copy=True)
    if sigmoid_transform:
        norms[0] = robust_sigmoid_fit(Xs)
        Xs = robust_sigmoid_apply(Xs, *norms[0])
    if minmax:
        lo, hi = float(Xs.min()), float(Xs.max())
        norms[1] = (lo, hi)
        Xs = (Xs - lo) / (hi - lo)
        lb, ub = data_bounds
        Xs = Xs * (ub - lb) + lb                                   # :186-191
    a, b = enc_range
    Xs = (b - a) * Xs + a                                          # :196-197
    return Xs, norms


def transform_test_data(X, norms, minmax=True, data_bounds=(0.0, 1.0), enc_range=(-1.0, 1.0)):
    """transform_test_data, utils.jl:202-275 (series are rows here)."""
    if X.size == 0:
        return np.array(X, dtype=float, copy=True), []
    Xs = np.array(X, dtype=float, copy=True)
    if norms[0] is not None:
        Xs = robust_sigmoid_apply(Xs, *norms[0])
    if norms[1] is not None:
        lo, hi = norms[1]
        Xs = (Xs - lo) / (hi - lo)
    if minmax:
        lb, ub = data_bounds
        Xs = Xs * (ub - lb) + lb
    oob = []
    for i in range(Xs.shape[0]):                                   # :243-266
        ts = Xs[i]
        tr = [i, 0.0, 1.0]
        lo, hi = ts.min(), ts.max()
        if lo < 0:
            ts -= lo
            hi = ts.max()
            tr[1] = lo
        if hi > 1:
            ts /= hi
            tr[2] = hi
        if tr[1:] != [0.0, 1.0]:
            oob.append(tr)
    a, b = enc_range
    Xs = (b - a) * Xs + a
    return Xs, oob


def legendre_encode(x, d, norm=False):
    """legendre_encode, bases.jl:77-92 with Pl(x,i;norm=Val(:normalized)) =
    sqrt((2i+1)/2) P_i(x) (LegendrePolynomials 0.4.5).  ``:Legendre`` ==
    ``:Legendre_No_Norm`` (options.jl:245-246) => norm=False by default."""
    x = np.asarray(x, dtype=float)
    out = np.empty(x.shape + (d,))
    for i in range(d):
        c = np.zeros(i + 1)
        c[i] = 1.0
        out[..., i] = math.sqrt((2 * i + 1) / 2.0) * np.polynomial.legendre.legval(x, c)
    if norm:
        out /= math.sqrt(math.sqrt((2 * d + 1) / 2.0) * d)         # Pl(1,d;normalized)*d  :86-89
    return out


def fourier_freqs(d):
    """get_fourier_freqs, bases.jl:27-34: 0, 1, -1, 2, -2, ..."""
    hb = math.ceil((d - 1.0) / 2.0)
    fr = [0]
    for i in range(1, hb + 1):
        fr += [i, -i]
    return fr[:d]


def fourier_encode(x, d):
    """fourier_encode, bases.jl:23-42: cispi(k x)/sqrt(d)."""
    x = np.asarray(x, dtype=float)
    k = np.asarray(fourier_freqs(d), dtype=float)
    return np.exp(1j * np.pi * x[..., None] * k) / math.sqrt(d)


def encode_dataset(X_orig, X_scaled, y, encoder, enc_range, class_keys=None):
    """encode_dataset / encode_safe_dataset(EncodeSeparate{false}),
    Encodings/encodings.jl:33-156.  X_* are (N, T).  Stable sort by label
    (:43, Julia sortperm is stable), range check (:115-119), class_keys =
    sorted classes -> slot (RealRealHighDimension.jl:485-486)."""
    y = np.asarray(y)
    if X_scaled.shape[0] == 0:
        return EncodedSet(np.zeros((0, 0, 0)), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int64))
    order = np.argsort(y, kind="stable")
    Xo, Xs, ys = X_orig[order], X_scaled[order], y[order]
    a, b = enc_range
    if not np.all((a <= Xs) & (Xs <= b)):
        raise ValueError(f"Data must be rescaled between {a} and {b} before encoding.")
    if class_keys is None:
        classes = np.unique(ys)
        class_keys = {c: i for i, c in enumerate(classes.tolist())}
    phi = encoder(Xs)
    lab = np.array([class_keys[v] for v in ys.tolist()], dtype=np.int32)
    vals, counts = np.unique(ys, return_counts=True)               # countmap + sortperm(keys) :151-152
    return EncodedSet(phi, lab, counts.astype(np.int64), labels=ys, original_data=Xo)


# --------------------------------------------------------------------------
# synthetic inputs  (docs/src/classification.md:20-43, src/Simulation/toy_data.jl:53-85)
# --------------------------------------------------------------------------

def trendy_sine_dataset(N, T, rng, sigma=0.1):
    """Two-class noisy trendy sine: x_t = sin(2 pi t/tau + psi) + m t/T + sigma n_t,
    t = 1..T (toy_data.jl:68-71); class 1 tau~U(12,15), class 2 tau~U(16,19),
    slope in {-3,0,3}, phase U(0,2pi) (classification.md:20-43)."""
    t = np.arange(1, T + 1, dtype=float)
    X = np.empty((N, T))
    y = np.empty(N, dtype=np.int64)
    half = N // 2
    for i in range(N):
        cls = 1 if i < half else 2
        tau = rng.uniform(12, 15) if cls == 1 else rng.uniform(16, 19)
        m = rng.choice([-3.0, 0.0, 3.0])
        psi = rng.uniform(0, 2 * np.pi)
        X[i] = np.sin(2 * np.pi * t / tau + psi) + m * t / T + sigma * rng.standard_normal(T)
        y[i] = cls
    return X, y


def random_mps(T, d, chi, C, rng, dtype=np.float64):
    """Stand-in for generate_startingMPS (RealRealHighDimension.jl:1-41): Gaussian
    site tensors with bond dimension chi (capped by d^j near the ends as
    ITensors.random_mps does), label index on the last site, unit norm, all
    sites left of T left-orthonormal (= orthogonalize!(W, T)).  Julia's RNG
    stream cannot be reproduced; golden tests carry their own initial MPS."""
    dims = [1]
    for j in range(1, T):
        dims.append(int(min(chi, d ** min(j, T - j, 30))))
    dims.append(1)
    W = []
    for j in range(T):
        shape = (dims[j], d, dims[j + 1]) + ((C,) if j == T - 1 else ())
        t = rng.standard_normal(shape)
        if np.issubdtype(dtype, np.complexfloating):
            t = t + 1j * rng.standard_normal(shape)
        W.append(t.astype(dtype))
    # left-canonicalise with QR sweeps (orthogonalize!(W, T))
    for j in range(T - 1):
        Dl, dd, Dr = W[j].shape
        Q, R = np.linalg.qr(W[j].reshape(Dl * dd, Dr))
        k = Q.shape[1]
        W[j] = Q.reshape(Dl, dd, k)
        if W[j + 1].ndim == 4:
            W[j + 1] = np.einsum("km,msbc->ksbc", R, W[j + 1])
        else:
            W[j + 1] = np.einsum("km,msb->ksb", R, W[j + 1])
    W[-1] = W[-1] / np.linalg.norm(W[-1])
    return W
