"""SweepEngine: thin Python mirror of the C ABI (one context = one GPU).

Arrays cross the boundary in the layouts of include/mpstime_hip.h.  On the
Python side a site tensor is an ndarray (Dl, d, Dr) or (Dl, d, Dr, C) - the
boundary layout is its Fortran-order (d, Dl, Dr[, C]) permutation.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _site_to_abi(t, dtype=np.float64):
    t = np.asarray(t, dtype=dtype)
    if t.ndim == 3:
        a = np.transpose(t, (1, 0, 2))          # (d, Dl, Dr)
    else:
        a = np.transpose(t, (1, 0, 2, 3))       # (d, Dl, Dr, C)
    return np.asfortranarray(a)


def _site_from_abi(buf, Dl, d, Dr, Cj):
    shape = (d, Dl, Dr) + ((Cj,) if Cj else ())
    a = np.reshape(buf, shape, order="F")
    return np.ascontiguousarray(np.transpose(a, (1, 0, 2) + ((3,) if Cj else ())))


DTYPES = {np.dtype(np.float64): L.F64, np.dtype(np.float32): L.F32, np.dtype(np.complex128): L.C128, np.dtype(np.complex64): L.C64}


def _dtype_of(x):
    """Element type of an array-like as the engine sees it (opts.dtype): float64 unless it already is one of the four."""
    dt = np.asarray(x).dtype
    return dt if dt in DTYPES else np.dtype(np.float64)


class SweepEngine:
    def __init__(self, device: int = 0):
        self.lib = L.load()
        self.ctx = C.c_void_p()
        rc = self.lib.mpst_create(C.byref(self.ctx), device)
        if rc:
            raise L.MPSTError(rc, (self.lib.mpst_last_error(None) or b"").decode())
        self.T = self.d = self.C = 0
        self.N = [0, 0]
        self.dtype = np.dtype(np.float64)      # element type of the data sets and the MPS (mpst_set_dataset's dtype)

    # -- plumbing ---------------------------------------------------------------------
    def _chk(self, rc):
        if rc:
            msg = (self.lib.mpst_last_error(self.ctx) or b"").decode()
            raise (L.SVDError if rc == L.MPST_ERR_SVD else L.MPSTError)(rc, msg)

    def close(self):
        if getattr(self, "ctx", None) and self.ctx.value:
            self.lib.mpst_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration ----------------------------------------------------------------
    def set_options(self, chi_max, eta=0.01, cutoff=1e-10, update_iters=1, loss="KLD", bbopt="TSGO",
                    rescale=(False, True), train_classes_separately=False, svd_alg=0, rebuild_caches=False, track_cost=False):
        if str(loss).upper() not in L.LOSS:
            raise L.MPSTError(L.MPST_ERR_UNSUPPORTED, f"loss {loss!r} unsupported by the array sweep")
        if str(bbopt).upper() not in L.OPT:
            raise L.MPSTError(L.MPST_ERR_UNSUPPORTED,
                              "Optim/OptimKit based solvers currently unimplemented for this version, "
                              "set 'use_legacy_ITensor=true' in MPSOptions to enable")
        o = L.mpst_options(int(chi_max), int(update_iters), L.LOSS[str(loss).upper()], L.OPT[str(bbopt).upper()],
                           int(bool(rescale[0])), int(bool(rescale[1])), int(bool(train_classes_separately)),
                           int(svd_alg), int(bool(rebuild_caches)), int(bool(track_cost)), float(eta), float(cutoff))
        self._iters = int(update_iters)
        self._chk(self.lib.mpst_set_options(self.ctx, C.byref(o)))

    def set_batch_hint(self, K):
        """This fit will be advanced in batches of about K (``sweep_batch``): fewer, longer gradient shares (mpst_set_batch_hint)."""
        self._chk(self.lib.mpst_set_batch_hint(self.ctx, int(K)))

    def set_dtype(self, dtype):
        """opts.dtype for data sets that are encoded on the device (mpst_set_dtype); set_dataset takes it from its array."""
        dt = np.dtype(dtype)
        self._chk(self.lib.mpst_set_dtype(self.ctx, DTYPES[dt]))
        self.dtype = dt
        self._dtype_fixed = True

    def set_dataset(self, which, phi, label_index, C_classes, global_counts=None, dtype=None):
        """``dtype``: element type the engine trains in (float64 / float32 / complex128 / complex64 = opts.dtype); default
        the array's own type."""
        dt = np.dtype(dtype) if dtype is not None else _dtype_of(phi)
        phi = np.ascontiguousarray(phi, dtype=dt)
        lab = np.ascontiguousarray(label_index, dtype=np.int32)
        N, T, d = phi.shape if phi.ndim == 3 and phi.size else (0, self.T, self.d)
        gc = None
        if global_counts is not None:
            gc = np.ascontiguousarray(global_counts, dtype=np.int64)
        self._chk(self.lib.mpst_set_dataset(
            self.ctx, which, phi.ctypes.data_as(C.c_void_p), lab.ctypes.data_as(C.POINTER(C.c_int32)), N, T, d,
            int(C_classes), DTYPES[dt], gc.ctypes.data_as(C.POINTER(C.c_int64)) if gc is not None else None))
        self.T, self.d, self.C = T, d, int(C_classes)
        self.N[which] = N
        self.dtype = dt

    def encode_dataset(self, which, X_sorted, label_index, C_classes, basis="Legendre_No_Norm", d=None, sigmoid_transform=True,
                       minmax=True, data_bounds=(0.0, 1.0), enc_range=(-1.0, 1.0), norms=None, rescale_out_of_bounds=True,
                       global_counts=None, sigmoid_fit=None):
        """Preprocess + encode the raw (N, T) matrix on the device (mpst_encode_dataset).  ``X_sorted`` must already be
        sorted by class.  ``norms=None`` fits a training set (median/IQR on the host, min/max on the device) and
        returns ``(norms, seconds)``; with ``norms`` from the training fit the data is treated as a test set and
        ``(oob, seconds)`` comes back, ``oob`` in the format of transform_test_data (utils.jl:243-266)."""
        from .encodings import Norms, model_encoding
        try:
            basis = model_encoding(basis).name           # canonical name; :Legendre is :Legendre_No_Norm (options.jl:245-246)
        except Exception:
            pass
        if basis not in L.BASIS:
            raise L.MPSTError(L.MPST_ERR_UNSUPPORTED, f"device-side encoding implements the closed-form bases {sorted(L.BASIS)}, not {basis!r}")
        X = np.ascontiguousarray(X_sorted, dtype=np.float64)
        lab = np.ascontiguousarray(label_index, dtype=np.int32)
        N, T = X.shape
        d = int(d if d is not None else self.d)
        eo = L.mpst_encode_opts()
        eo.basis, eo.sigmoid_transform, eo.minmax = L.BASIS[basis], int(bool(sigmoid_transform)), int(bool(minmax))
        eo.is_test, eo.rescale_out_of_bounds = int(norms is not None), int(bool(rescale_out_of_bounds))
        eo.data_lb, eo.data_ub = map(float, data_bounds)
        eo.range_a, eo.range_b = map(float, enc_range)
        if norms is None:
            if sigmoid_transform:
                if sigmoid_fit is not None:             # (median, iqr) fitted elsewhere, e.g. over all shards
                    eo.median, eo.iqr = map(float, sigmoid_fit)
                else:
                    eo.fit_sigmoid = 1                  # median / quartiles from a device sort of the values
        else:
            if norms.sigmoid is not None:
                eo.median, eo.iqr = norms.sigmoid
            eo.sigmoid_transform = int(norms.sigmoid is not None)
            if norms.minmax is not None:
                eo.lo, eo.hi = norms.minmax
        gc = np.ascontiguousarray(global_counts, dtype=np.int64) if global_counts is not None else None
        fix = np.zeros((N, 2)) if norms is not None else None
        sec = C.c_double()
        dp = C.POINTER(C.c_double)
        self._chk(self.lib.mpst_encode_dataset(
            self.ctx, which, X.ctypes.data_as(dp), lab.ctypes.data_as(C.POINTER(C.c_int32)), N, T, d, int(C_classes), C.byref(eo),
            gc.ctypes.data_as(C.POINTER(C.c_int64)) if gc is not None else None,
            fix.ctypes.data_as(dp) if fix is not None else None, C.byref(sec)))
        self.T, self.d, self.C = T, d, int(C_classes)
        self.N[which] = N
        self.dtype = self._ctx_dtype(basis)
        if norms is None:
            out = Norms(sigmoid=(eo.median, eo.iqr) if sigmoid_transform else None, minmax=(eo.lo, eo.hi) if minmax else None)
            return out, sec.value
        oob = [[i, float(fix[i, 0]), float(fix[i, 1])] for i in range(N) if fix[i, 0] != 0.0 or fix[i, 1] != 1.0]
        return oob, sec.value

    def encode_values(self, X, basis="Legendre_No_Norm", d=4, sigmoid_transform=False, minmax=False, data_bounds=(0.0, 1.0),
                      enc_range=None, norms=None, rescale_out_of_bounds=False):
        """mpst_encode_values: the device preprocessing + encoding kernels on a raw (N, T) matrix, states back to the host -
        (N, T, d) float64 for the Legendre / Uniform bases, complex128 for "Fourier", "Stoudenmire", "Sahand".  Defaults: X is already in the encoding's
        domain (no transforms, identity range map); with transforms the [0, 1] data is mapped onto ``enc_range``
        (default: the basis' own range, (-1, 1) for Legendre / Fourier, (0, 1) for Stoudenmire / Sahand / Uniform).  ``norms`` as for encode_dataset (a test set) or None."""
        from .encodings import model_encoding
        basis_range = (-1.0, 1.0)
        try:
            enc_ = model_encoding(basis)
            basis, basis_range = enc_.name, (enc_.range or basis_range)
        except Exception:
            pass
        if basis not in L.BASIS:
            raise L.MPSTError(L.MPST_ERR_UNSUPPORTED, f"device-side encoding implements the closed-form bases {sorted(L.BASIS)}, not {basis!r}")
        X = np.ascontiguousarray(X, dtype=np.float64)
        N, T = X.shape
        eo = L.mpst_encode_opts()
        eo.basis, eo.sigmoid_transform, eo.minmax = L.BASIS[basis], int(bool(sigmoid_transform)), int(bool(minmax))
        eo.is_test, eo.rescale_out_of_bounds = int(norms is not None), int(bool(rescale_out_of_bounds))
        eo.data_lb, eo.data_ub = map(float, data_bounds)
        if enc_range is None:
            enc_range = basis_range if (sigmoid_transform or minmax or norms is not None) else (0.0, 1.0)
        eo.range_a, eo.range_b = map(float, enc_range)
        if norms is None:
            eo.fit_sigmoid = int(bool(sigmoid_transform))
        else:
            if norms.sigmoid is not None:
                eo.median, eo.iqr = norms.sigmoid
            eo.sigmoid_transform = int(norms.sigmoid is not None)
            if norms.minmax is not None:
                eo.lo, eo.hi = norms.minmax
        cx = basis in ("Fourier", "Stoudenmire", "Sahand")
        out = np.zeros((N, T, int(d)), dtype=np.complex128 if cx else np.float64)
        sec = C.c_double()
        self._chk(self.lib.mpst_encode_values(self.ctx, X.ctypes.data_as(C.POINTER(C.c_double)), N, T, int(d), C.byref(eo),
                                              out.ctypes.data_as(C.c_void_p), None, C.byref(sec)))
        return out, sec.value

    def _ctx_dtype(self, basis):
        """element type after a device-side encoding: what set_dtype fixed, else the basis' own (float64 / complex128)"""
        cx = basis in ("Fourier", "Stoudenmire", "Sahand")
        if getattr(self, "_dtype_fixed", False):
            return self.dtype
        return np.dtype(np.complex128 if cx else np.float64)

    def get_encoded(self, which=0):
        phi = np.zeros((self.N[which], self.T, self.d), dtype=self.dtype)
        self._chk(self.lib.mpst_get_encoded(self.ctx, which, phi.ctypes.data_as(C.POINTER(C.c_double))))
        return phi

    def set_mps(self, W, label_site=None):
        T = len(W)
        if label_site is None:
            label_site = [j for j, t in enumerate(W) if np.ndim(t) == 4]
            assert len(label_site) == 1, "exactly one site must carry the label index"
            label_site = label_site[0]
        chi = np.array([W[0].shape[0]] + [t.shape[2] for t in W], dtype=np.int32)
        bufs = [_site_to_abi(t, self.dtype) for t in W]       # in the context's element type
        ptrs = (C.c_void_p * T)(*[b.ctypes.data for b in bufs])
        self._chk(self.lib.mpst_set_mps(self.ctx, ptrs, chi.ctypes.data_as(C.POINTER(C.c_int32)), T, int(label_site)))

    def get_chi(self):
        chi = np.zeros(self.T + 1, dtype=np.int32)
        ls = C.c_int32()
        self._chk(self.lib.mpst_get_chi(self.ctx, chi.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ls)))
        return chi, ls.value

    def get_mps(self):
        chi, ls = self.get_chi()
        bufs = []
        for j in range(self.T):
            n = self.d * chi[j] * chi[j + 1] * (self.C if j == ls else 1)
            bufs.append(np.zeros(int(n), dtype=self.dtype))
        ptrs = (C.c_void_p * self.T)(*[b.ctypes.data for b in bufs])
        self._chk(self.lib.mpst_get_mps(self.ctx, ptrs))
        return [_site_from_abi(bufs[j], int(chi[j]), self.d, int(chi[j + 1]), self.C if j == ls else 0)
                for j in range(self.T)]

    # -- the path ---------------------------------------------------------------------
    def build_caches(self):
        self._chk(self.lib.mpst_build_caches(self.ctx))

    def sweep(self):
        st = L.mpst_sweep_stats()
        self._chk(self.lib.mpst_sweep(self.ctx, C.byref(st)))
        return {"seconds": st.seconds, "max_chi": st.max_chi, "eig_sweeps_total": st.eig_sweeps_total,
                "eig_fallbacks": st.eig_fallbacks}

    def loss_trace(self):
        """(2(T-1), update_iters + 1): losses before every optimiser step and at the updated bond tensor (track_cost)."""
        out = np.zeros((2 * (self.T - 1), self._iters + 1))
        self._chk(self.lib.mpst_get_loss_trace(self.ctx, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def bond_step(self, lid, going_left):
        dbg = L.mpst_bond_debug()
        self._chk(self.lib.mpst_bond_step(self.ctx, int(lid), int(bool(going_left)), C.byref(dbg)))
        return {"loss": dbg.loss, "grad_norm": dbg.grad_norm, "bt_new_norm": dbg.bt_norm, "chi": dbg.chi_new,
                "eig_sweeps": dbg.eig_sweeps, "S": np.array(dbg.spectrum[:dbg.n_spectrum])}

    def eval(self, which=0):
        mse, kld, acc = C.c_double(), C.c_double(), C.c_double()
        conf = np.zeros((self.C, self.C), dtype=np.int64)
        self._chk(self.lib.mpst_eval(self.ctx, which, C.byref(mse), C.byref(kld), C.byref(acc),
                                     conf.ctypes.data_as(C.POINTER(C.c_int64))))
        return mse.value, kld.value, acc.value, conf

    def classify(self, which=1, return_overlaps=False):
        N = self.N[which]
        pred = np.zeros(N, dtype=np.int32)
        yh = np.zeros((N, self.C), dtype=np.complex128 if self.dtype.kind == "c" else np.float64)     # overlaps are fp64 (pairs)
        self._chk(self.lib.mpst_classify(self.ctx, which, pred.ctypes.data_as(C.POINTER(C.c_int32)),
                                         yh.ctypes.data_as(C.POINTER(C.c_double))))
        return (pred, yh) if return_overlaps else pred

    def impute(self, which, missing, grid_x, grid_phi, method=0, get_wmad=True, u=None, order=0, max_trials=1,
               rejection_threshold=0.0, mean_basis=1):
        """mpst_impute: (x, err, seconds); x / err are (N, T) with the imputed value / its uncertainty at every missing
        site.  method 0 median, 1 mode, 2 quantile of u (N, T), 3 mean, 4 inverse-transform sampling with rejection
        (u (N, T, max_trials)); order 0 forwards, 1 backwards."""
        m = np.ascontiguousarray(missing, dtype=np.uint8)
        N, T = m.shape
        gx = np.ascontiguousarray(grid_x, dtype=np.float64)
        gp = np.ascontiguousarray(grid_phi, dtype=np.complex128 if self.dtype.kind == "c" else np.float64)   # grid states: fp64 (pairs)
        assert gp.shape == (len(gx), self.d) and N == self.N[which] and T == self.T
        uu = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
        if uu is not None:
            assert uu.size == N * T * (int(max_trials) if int(method) == 4 else 1)
        x = np.zeros((N, T))
        err = np.zeros((N, T))
        sec = C.c_double()
        dp = C.POINTER(C.c_double)
        o = L.ImputeOpts(int(method), int(order), int(bool(get_wmad)), int(max_trials), int(mean_basis), 0, float(rejection_threshold))
        self._chk(self.lib.mpst_impute(self.ctx, which, m.ctypes.data_as(C.POINTER(C.c_uint8)), gx.ctypes.data_as(dp),
                                       C.cast(gp.ctypes.data, dp), len(gx), C.byref(o),
                                       uu.ctypes.data_as(dp) if uu is not None else None, x.ctypes.data_as(dp),
                                       err.ctypes.data_as(dp), C.byref(sec)))
        return x, err, sec.value

    def impute_model(self, W, phi, label_index, missing, grid_x, grid_phi, method=0, get_wmad=True, u=None, order=0, max_trials=1,
                     rejection_threshold=0.0, mean_basis=None, compute="f64", label_site=None):
        """mpst_impute_model_run: the imputation engine on a model handed over in one call.  ``W``: site tensors
        (Dl, d, Dr), the label site (Dl, d, Dr, C); ``phi`` (N, T, d) encoded known values; real or complex (then
        ``grid_phi`` is complex too).  ``compute`` "f64" or "f32" (fp32 chain contractions, fp64 densities).
        Returns (x, err, seconds)."""
        cx = any(np.iscomplexobj(t) for t in W) or np.iscomplexobj(phi) or np.iscomplexobj(grid_phi)
        dt = np.complex128 if cx else np.float64
        T = len(W)
        if label_site is None:
            label_site = [j for j, t in enumerate(W) if np.ndim(t) == 4]
            assert len(label_site) == 1, "exactly one site must carry the label index"
            label_site = label_site[0]
        Cn = int(W[label_site].shape[3])
        d = int(W[0].shape[1])
        chi = np.array([W[0].shape[0]] + [t.shape[2] for t in W], dtype=np.int32)
        bufs = [_site_to_abi(t, dt) for t in W]
        ptrs = (C.c_void_p * T)(*[b.ctypes.data for b in bufs])
        ph = np.ascontiguousarray(phi, dtype=dt)
        lab = np.ascontiguousarray(label_index, dtype=np.int32)
        m = np.ascontiguousarray(missing, dtype=np.uint8)
        N = ph.shape[0]
        assert ph.shape == (N, T, d) and m.shape == (N, T) and lab.shape == (N,)
        gx = np.ascontiguousarray(grid_x, dtype=np.float64)
        gp = np.ascontiguousarray(grid_phi, dtype=dt)
        assert gp.shape == (len(gx), d)
        uu = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
        if uu is not None:
            assert uu.size == N * T * (int(max_trials) if int(method) == 4 else 1)
        if mean_basis is None:
            mean_basis = 2 if cx else 1
        model = L.ImputeModel(N, T, d, Cn, int(label_site), 1 if cx else 0, {"f64": 0, "f32": 1}[compute],
                              C.cast(ptrs, C.POINTER(C.c_void_p)), chi.ctypes.data_as(C.POINTER(C.c_int32)),
                              ph.ctypes.data_as(C.c_void_p), lab.ctypes.data_as(C.POINTER(C.c_int32)))
        o = L.ImputeOpts(int(method), int(order), int(bool(get_wmad)), int(max_trials), int(mean_basis), 0, float(rejection_threshold))
        x = np.zeros((N, T))
        err = np.zeros((N, T))
        sec = C.c_double()
        dp = C.POINTER(C.c_double)
        self._chk(self.lib.mpst_impute_model_run(self.ctx, C.byref(model), m.ctypes.data_as(C.POINTER(C.c_uint8)), gx.ctypes.data_as(dp),
                                                 gp.ctypes.data_as(C.c_void_p), len(gx), C.byref(o),
                                                 uu.ctypes.data_as(dp) if uu is not None else None, x.ctypes.data_as(dp),
                                                 err.ctypes.data_as(dp), C.byref(sec)))
        return x, err, sec.value

    def impute_phases(self):
        """(environment pass, density sweep) device seconds of the last imputation call."""
        out = np.zeros(2)
        self._chk(self.lib.mpst_get_impute_phases(self.ctx, out.ctypes.data_as(C.POINTER(C.c_double))))
        return float(out[0]), float(out[1])

    def impute_info(self):
        """how the last imputation call ran: {"closed_form_densities": bool (Fourier / Legendre grid states on a uniform grid),
        "batched_sweep": bool (sixteen instances per workgroup)}"""
        out = (C.c_int32 * 2)()
        self._chk(self.lib.mpst_get_impute_info(self.ctx, out, 2))
        return {"closed_form_densities": bool(out[0]), "batched_sweep": bool(out[1])}

    def normalize(self):
        self._chk(self.lib.mpst_normalize(self.ctx))

    # -- diagnostics ------------------------------------------------------------------
    def set_profile(self, mask):
        self._chk(self.lib.mpst_set_profile(self.ctx, int(mask)))

    def get_profile(self):
        us = np.zeros(16)
        cnt = np.zeros(16, dtype=np.int64)
        self._chk(self.lib.mpst_get_profile(self.ctx, us.ctypes.data_as(C.POINTER(C.c_double)),
                                            cnt.ctypes.data_as(C.POINTER(C.c_int64))))
        return {k: (us[i], int(cnt[i])) for i, k in enumerate(L.KERNEL_CLASSES)}

    def info(self):
        out = (C.c_int32 * 20)()
        self._chk(self.lib.mpst_get_info_n(self.ctx, out, 20))
        return {"four_launch_chain": bool(out[18]), "tail_redos": out[19],
                "subspace_attempted": out[16], "subspace_accepted": out[17], "fused": bool(out[0]), "large_bond": bool(out[1]), "nparts": out[2], "nchunks": out[3], "cap": out[4],
                "ranks": out[5], "graph": bool(out[6]), "library_eig_fallbacks": out[7], "persistent_tridiag_aborts": out[8],
                "xcd_local_misplaced": out[9], "sliced_bond_gemms": bool(out[10]), "grad_shares": out[11],
                "eig_merged": bool(out[12]), "large_bond_sweep_redos": out[13], "large_bond_verdict_per_sweep": bool(out[14]),
                "typed_kernels": bool(out[15]), "dtype": (out[15] - 1) if out[15] else L.F64}

    def eig_phases(self):
        us = np.zeros(6)
        self._chk(self.lib.mpst_get_eig_phases(self.ctx, us.ctypes.data_as(C.POINTER(C.c_double))))
        return dict(zip(("tridiag", "bisect", "eigvec", "backtransform", "verify", "tridiag_cycles"), us.tolist()))

    def tail_phases(self):
        """Stamps (us since its first tile workgroup started) of the last stamped k_bond_tail launch: that tile workgroup, the first
        workgroup of the next bond's tensor, the first back-split workgroup."""
        us = np.zeros(55)
        self._chk(self.lib.mpst_get_tail_phases(self.ctx, us.ctypes.data_as(C.POINTER(C.c_double))))
        tile = ("start", "candidates_requested", "factors_requested", "bond_dims_known", "all_requested", "factors_in_lds", "truncation",
                "candidates_in_lds", "polished", "overlap_product_issued", "role_requested", "b1_passed", "s_tile_formed", "env_rows", "z_rowdot", "tile_done", "role_done",
                "stores_drained")
        role = tile
        pick = lambda names, x: {k: round(float(v), 2) for k, v in zip(names, x) if v >= 0 or k == "start"}
        # workgroup 0 (hosts a job of the next bond's tensor when the sweep goes on), the first back-split host, the last workgroup (no role)
        return {"host_of_a_chain_job": pick(tile, us[:16]), "plain_tile": pick(role, us[16:32]),
                "bonds_by_candidate_orthogonality": dict(zip(("below_1e-13", "below_1e-8", "below_3e-5", "above"), [int(x) for x in us[48:52]])),
                "all_workgroups": {"first_start": round(float(us[52]), 2), "last_start": round(float(us[54]), 2), "last_end": round(float(us[53]), 2)}}

    def selftest_mfma(self, A, B):
        A = np.ascontiguousarray(A, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        K = A.shape[1]
        out = np.zeros((16, 16))
        dp = C.POINTER(C.c_double)
        self._chk(self.lib.mpst_selftest_mfma(self.ctx, A.ctypes.data_as(dp), B.ctypes.data_as(dp), K,
                                              out.ctypes.data_as(dp)))
        return out

    def selftest_eig(self, G, alg=0):
        G = np.ascontiguousarray(G, dtype=np.float64)
        n = G.shape[0]
        lam = np.zeros(n)
        E = np.zeros((n, n))
        sw = C.c_int32()
        dp = C.POINTER(C.c_double)
        self._chk(self.lib.mpst_selftest_eig(self.ctx, G.ctypes.data_as(dp), n, alg, lam.ctypes.data_as(dp),
                                             E.ctypes.data_as(dp), C.byref(sw)))
        return lam, E, sw.value


def sweep_batch(engines):
    """mpst_sweep_batch: one sweep of K independent fits of the same shape in ONE launch chain (every launch carries all K
    fits).  ``engines``: SweepEngines prepared like for ``sweep()`` (options, data, MPS, build_caches).  Returns one stats dict per
    engine; ``seconds`` is the device time of the whole batch.  Raises MPSTError(MPST_ERR_UNSUPPORTED) for fits outside the
    headline chain or of different shapes - drive those with ``sweep()`` one by one."""
    lib = L.load()
    K = len(engines)
    arr = (C.c_void_p * K)(*[e.ctx.value for e in engines])
    st = (L.mpst_sweep_stats * K)()
    rc = lib.mpst_sweep_batch(arr, K, st)
    if rc:
        engines[0]._chk(rc)
    return [{"seconds": s.seconds, "max_chi": s.max_chi, "eig_sweeps_total": s.eig_sweeps_total, "eig_fallbacks": s.eig_fallbacks} for s in st]


def sweep_batch_multi(engines, groups=None):
    """mpst_sweep_batch_multi: one sweep of K independent fits dealt over several devices - ``groups[k]`` names the group of engine k
    (default: its device); every group is one ``sweep_batch`` launch chain on its own host thread, all groups run concurrently, no
    collective.  Returns one stats dict per engine (``seconds`` = device time of its group)."""
    lib = L.load()
    K = len(engines)
    arr = (C.c_void_p * K)(*[e.ctx.value for e in engines])
    g = None if groups is None else (C.c_int32 * K)(*[int(x) for x in groups])
    st = (L.mpst_sweep_stats * K)()
    rc = lib.mpst_sweep_batch_multi(arr, K, g, st)
    if rc:
        engines[0]._chk(rc)
    return [{"seconds": s.seconds, "max_chi": s.max_chi, "eig_sweeps_total": s.eig_sweeps_total, "eig_fallbacks": s.eig_fallbacks} for s in st]


def comm_library():
    """Which librccl the HIP library has bound (it is loaded at run time, never linked): path + how it was found, the
    library's ncclGetVersion code and the NCCL_VERSION_CODE the HIP library was compiled against."""
    lib = L.load()
    buf = C.create_string_buffer(1024)
    ver, built = C.c_int32(0), C.c_int32(0)
    rc = lib.mpst_comm_library(buf, 1024, C.byref(ver), C.byref(built))
    return {"ok": rc == 0, "library": buf.value.decode(), "version": ver.value, "built_against": built.value}
