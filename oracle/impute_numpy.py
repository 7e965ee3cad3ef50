"""oracle/impute_numpy.py - NumPy restatement of MPSTime.jl's imputation path (src/Imputation).

TEST INFRASTRUCTURE ONLY (checker of tests/test_gpu_impute.py and tests/test_impute_oracle.py); never imported by the
package.  PARITY UNPINNED against Julia: the reference's known-answer tests for this path
(test/imputation.jl:33-34,48-55) need its downloaded ECG200 split, a BigFloat-trained MPS that is not in the tree and
Julia's Xoshiro stream for the missing-site masks.  What pins it instead: it follows the reference line by line
(citations below), and tests/test_impute_oracle.py checks it against an independent brute-force definition of the
conditional densities (full state-vector contraction) on small chains.

Conventions: a site tensor is an ndarray (Dl, d, Dr) (label site already sliced to one class, expand_label_index);
sites and grid indices are 0-based; `enc[j]` is the length-d encoded state of site j.
"""
from __future__ import annotations

import numpy as np


def expand_label_index(W):
    """src/utils.jl:356-370: one MPS per class, label index fixed, each normalised."""
    pos = [j for j, t in enumerate(W) if t.ndim == 4][0]
    out = []
    for c in range(W[pos].shape[3]):
        mps = [t.copy() for t in W]
        mps[pos] = W[pos][:, :, :, c].copy()
        nrm = mps_norm3(mps)
        mps[-1] = mps[-1] / nrm            # normalize!(mpsc): the scale is irrelevant to every conditional density
        out.append(mps)
    return out


def mps_norm3(mps):
    E = np.ones((1, 1))
    for t in mps:
        E = np.einsum("ab,asc,bsd->cd", E, t, np.conj(t))
    return float(np.sqrt(abs(E[0, 0])))


def precondition(class_mps, enc, imputation_sites):
    """MPS_methods.jl:42-99: project every known site onto its encoded state; what is left is an MPS over the missing
    sites only.  Returns the list of conditioned tensors: first one (d, r) [or (d,) when it is also the last], middle
    ones (l, d, r), last one (l, d)."""
    T = len(class_mps)
    imp = list(imputation_sites)
    known = [j for j in range(T) if j not in set(imp)]
    known_set = set(known)
    n_imp = len(imp)

    def cond_until_next(i, it):
        # condition_until_next!, :1-22: it *= (class_mps[i] * dag(timeseries_enc[i])) while i is known
        while i < T and i in known_set:
            M = np.einsum("asb,s->ab", class_mps[i], np.conj(enc[i]))
            it = M if it is None else it @ M
            i += 1
        return i, it

    cond = []
    i = 0
    idx = 0
    while i < T:
        if idx == n_imp - 1:
            it = None
            if i in known_set:
                i, it = cond_until_next(i, it)
            last = class_mps[i]
            i += 1
            i, it2 = cond_until_next(i, None)
            t = last if it is None else np.einsum("xa,asb->xsb", it, last)
            if it2 is not None:
                t = np.einsum("xsb,by->xsy", t, it2)
            cond.append(t)
        elif i in known_set:
            i, it = cond_until_next(i, None)
            cond.append(np.einsum("xa,asb->xsb", it, class_mps[i]))
        else:
            cond.append(class_mps[i].copy())
        idx += 1
        i += 1
    return cond


def _right_orthogonalize(cond):
    """orthogonalize!(mps, 1): every tensor right of the first becomes right-orthonormal (LQ sweeps from the end)."""
    t = [c.copy() for c in cond]
    for j in range(len(t) - 1, 0, -1):
        l, d, r = t[j].shape
        m = t[j].reshape(l, d * r)
        q, rr = np.linalg.qr(m.T)                     # m = rr^T q^T
        t[j] = q.T.reshape(q.shape[1], d, r)
        t[j - 1] = np.einsum("asb,bk->ask", t[j - 1], rr.T)
    return t


def _left_orthogonalize(cond):
    t = [c.copy() for c in cond]
    for j in range(len(t) - 1):
        l, d, r = t[j].shape
        q, rr = np.linalg.qr(t[j].reshape(l * d, r))
        t[j] = q.reshape(l, d, q.shape[1])
        t[j + 1] = np.einsum("kb,bsc->ksc", rr, t[j + 1])
    return t


def cumul_trapz_even(xs, p):
    """NumericalIntegration.cumul_integrate(x, y, TrapezoidalEvenFast()): evenly spaced, dx = x[2]-x[1]."""
    dx = xs[1] - xs[0]
    out = np.zeros(len(p))
    out[1:] = np.cumsum((p[:-1] + p[1:])) * (dx / 2.0)
    return out


def weighted_median(v, w):
    """StatsBase.median(v, weights): value at which the cumulative weight (in increasing order of v, stable) first
    exceeds half the total; one weight above half the total wins outright."""
    mask = w != 0
    v, w = v[mask], w[mask]
    mid = w.sum() / 2.0
    k = int(np.argmax(w))
    if w[k] > mid:
        return float(v[k])
    order = np.argsort(v, kind="stable")
    cw = np.cumsum(w[order])
    hit = int(np.searchsorted(cw, mid, side="right"))          # first index with cw > mid
    if hit > 0 and cw[hit - 1] == mid:
        return float(0.5 * (v[order[hit - 1]] + v[order[hit]]))
    return float(v[order[hit]])


def probs_from_rdm(A, grid_phi):
    """What impute_at! evaluates on every grid state (MPS_methods.jl:124-130,155-157 + sampling_utils.jl:19-50).
    `rdm .= A * A'` is allocated as a plain Matrix - the `eltype(...) isa SVector` test at :126 is never true - so
    get_conditional_probability dispatches to the (state, A::Matrix) methods with A = rdm: p = state' * rdm,
    abs(dot(p, p)).  The reference's "probability" of x is therefore  phi(x)^H rho rho^H phi(x) = |rho phi(x)|^2,
    NOT the Born value phi^H rho phi the SVector/MMatrix method (:43-48) would give.  Restated as executed."""
    rho = A @ np.conj(A).T                                      # (d, d)
    P = np.conj(grid_phi) @ rho                                 # rows phi^H rho
    return np.sum(np.abs(P) ** 2, axis=1)


def impute_at(cond, xs, grid_phi, method="median", order="forwards", get_wmad=True, u=None, encode=None,
              rejection_threshold=None, max_trials=10):
    """impute_at!, MPS_methods.jl:103-177, with get_median_from_rdm (:159-196 of sampling_utils.jl), get_mode_from_rdm
    (:98-143), get_mean_from_rdm (:66-96; `encode(x)` gives the state of the expectation value) and
    get_sample_from_rdm (:262-296) with or without rejection; the uniform numbers `u[k]` (one, or max_trials with
    rejection) for the k-th conditioned site are supplied by the caller.  Returns (x, err) per conditioned site.

    Backwards order is restated as the mirror image of forwards (orthogonality centre on the last missing site, the
    vector handed leftwards).  In the reference the re-conditioning step tests `order == :forwards` (:163) where
    `order` is ITensors' function, not `impute_order`: the test is always false and the contraction is written for the
    LAST index of the next tensor, which after orthogonalize! is the bond to the site just imputed in either order -
    i.e. exactly this recursion."""
    n = len(cond)
    if order == "forwards":
        t = _right_orthogonalize(cond)              # orthogonalize!(mps, first_idx), :115
        idxs = list(range(n))
        c = t[0]
        assert c.shape[0] == 1, "everything left of the first missing site is known: trivial left index"
        A = c[0]                                     # (d, r)
    else:
        t = _left_orthogonalize(cond)               # :120
        idxs = list(range(n - 1, -1, -1))
        c = t[-1]
        assert c.shape[2] == 1, "everything right of the last missing site is known: trivial right index"
        A = c[:, :, 0].T                             # (d, l)
    xout, eout = np.zeros(n), np.zeros(n)
    for ii, i in enumerate(idxs):
        p = probs_from_rdm(A, grid_phi)
        if method == "mode":
            k = int(np.argmax(p))
            ms, xk = grid_phi[k], xs[k]
            err = 0.0
        elif method == "mean":
            dxs = xs[1] - xs[0]
            Z = dxs * (np.sum(p) - 0.5 * (p[0] + p[-1]))        # integrate(x, y, TrapezoidalEvenFast())
            dx = np.mean(np.abs(np.diff(xs)))                    # impute_mean, MPS_methods.jl:250
            xk = float(np.sum(xs * p) * dx / Z)
            ms = np.asarray(encode(xk)) / np.sqrt(Z)
            err = float(np.sqrt(np.sum((xs - xk) ** 2 * p) * dx / Z)) if get_wmad else 0.0
        else:
            cdf = cumul_trapz_even(xs, p)
            Z = cdf[-1]
            cdf = cdf / Z
            pn = p / Z
            if method == "median":
                k = int(np.argmin(np.abs(cdf - 0.5)))
                err = weighted_median(np.abs(xs - xs[k]), pn) if get_wmad else 0.0
            elif rejection_threshold is None:                    # :none
                k = int(np.argmin(np.abs(cdf - float(np.ravel(u[ii])[0]))))
                err = 0.0
            else:
                km = int(np.argmin(np.abs(cdf - 0.5)))
                err = weighted_median(np.abs(xs - xs[km]), pn)
                k = km
                for trial in range(max_trials):
                    k = int(np.argmin(np.abs(cdf - float(u[ii][trial]))))
                    if abs(xs[k] - xs[km]) < rejection_threshold * err:
                        break
            ms, xk = grid_phi[k] / np.sqrt(Z), xs[k]
        xout[i], eout[i] = xk, err
        if ii != n - 1:
            Am = np.conj(ms) @ A                               # ms' * A
            nxt = t[idxs[ii + 1]]
            if order == "forwards":
                A = np.einsum("a,asb->sb", Am, nxt)
            else:
                A = np.einsum("asb,b->sa", nxt, Am)
            A = A / np.max(np.abs(A))                          # scale-free from here on (the reference runs norm=false)
    return xout, eout


def impute(class_mps, enc, imputation_sites, xs, grid_phi, method="median", order="forwards", get_wmad=True, u=None, **kw):
    """impute_median / impute_mean / impute_mode / impute_ITS (MPS_methods.jl:198-345) for one series.  `enc` (T, d): encoded states of
    the KNOWN values (rows of missing sites are ignored).  Returns (x_imputed, err) at the imputation sites."""
    imp = sorted(int(j) for j in imputation_sites)
    cond = precondition(class_mps, enc, imp)
    return impute_at(cond, xs, grid_phi, method, order, get_wmad, u, **kw)


def brute_force_conditional(class_mps, enc, known_mask, site, fixed, grid_phi):
    """Independent definition for the tests: p(x_site | known values, already imputed values `fixed` {site: state}),
    marginalising every other missing site, by contracting the full chain with density-matrix environments."""
    T = len(class_mps)
    # left environment (density form) up to `site`
    L = np.ones((1, 1))
    for j in range(site):
        W = class_mps[j]
        if known_mask[j] or j in fixed:
            v = enc[j] if known_mask[j] else fixed[j]
            M = np.einsum("asb,s->ab", W, np.conj(v))
            L = M.T @ L @ np.conj(M)
        else:
            L = np.einsum("ab,asc,bsd->cd", L, W, np.conj(W))
    R = np.ones((1, 1))
    for j in range(T - 1, site, -1):
        W = class_mps[j]
        if known_mask[j] or j in fixed:
            v = enc[j] if known_mask[j] else fixed[j]
            M = np.einsum("asb,s->ab", W, np.conj(v))
            R = M @ R @ np.conj(M).T
        else:
            R = np.einsum("asc,cd,bsd->ab", W, R, np.conj(W))
    W = class_mps[site]
    rho = np.einsum("ab,asc,cd,btd->st", L, W, R, np.conj(W))          # (d, d)
    P = np.conj(grid_phi) @ rho
    return np.sum(np.abs(P) ** 2, axis=1)                               # the reference's |rho phi|^2 (see probs_from_rdm)
