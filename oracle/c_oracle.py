"""ctypes driver of oracle/mps_oracle.c (TEST INFRASTRUCTURE ONLY; sweep parity unpinned - see the C header).

LAPACK ``dgesdd`` is taken from SciPy's bundled LAPACK through the C function pointer
exported by ``scipy.linalg.cython_lapack`` - the same routine Julia's DivideAndConquer SVD calls.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "mps_oracle.c")
_libs = {}


class orc_opts(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("T", "d", "C", "N", "chi_max", "cap", "update_iters", "loss", "opt",
                                        "rescale_before", "rescale_after", "train_sep", "rebuild_caches")] + \
               [("eta", C.c_double), ("cutoff", C.c_double)]


def _gesdd_pointer():
    import scipy.linalg.cython_lapack as cl
    cap = cl.__pyx_capi__["dgesdd"]
    C.pythonapi.PyCapsule_GetName.restype = C.c_char_p
    C.pythonapi.PyCapsule_GetName.argtypes = [C.py_object]
    C.pythonapi.PyCapsule_GetPointer.restype = C.c_void_p
    C.pythonapi.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
    return C.pythonapi.PyCapsule_GetPointer(cap, C.pythonapi.PyCapsule_GetName(cap))


def build(native=False, out_dir=None):
    """Compile mps_oracle.c.  native=True tunes for the CPU it is run on (cpu_baseline timing)."""
    out_dir = out_dir or _HERE
    name = "libmps_oracle_native.so" if native else "libmps_oracle.so"
    path = os.path.join(out_dir, name)
    # the parity checker never gets -ffast-math; the timed cpu_baseline copy does (like the reference's @fastmath loops)
    flags = ["-march=native", "-ffast-math"] if native else ["-march=x86-64-v3"]
    subprocess.check_call(["gcc", "-O3", *flags, "-fPIC", "-std=gnu11", "-shared", _SRC, "-o", path, "-lm"])
    return path


def load(native=False):
    key = bool(native)
    if key in _libs:
        return _libs[key]
    if native:
        path = build(native=True, out_dir=tempfile.mkdtemp(prefix="mps_oracle_"))
    else:
        path = os.path.join(_HERE, "libmps_oracle.so")
        if not os.path.exists(path):
            path = build()
    lib = C.CDLL(path)
    lib.orc_run.restype = C.c_int
    _libs[key] = lib
    return lib


class COracle:
    """State of one training run in the C oracle (site tensors, caches, bond dimensions)."""

    def __init__(self, W, phi, label_index, class_distribution, chi_max, eta=0.01, cutoff=1e-10, update_iters=1,
                 loss="KLD", bbopt="TSGO", rescale=(False, True), train_classes_separately=False,
                 rebuild_caches=True, native=False):
        self.lib = load(native)
        self.phi = np.ascontiguousarray(phi, dtype=np.float64)
        N, T, d = self.phi.shape
        Cn = len(class_distribution)
        self.label = np.ascontiguousarray(label_index, dtype=np.int32)
        self.counts = np.ascontiguousarray(class_distribution, dtype=np.int64)
        cap = max([chi_max] + [t.shape[2] for t in W])
        self.o = orc_opts(T, d, Cn, N, int(chi_max), cap, int(update_iters), {"KLD": 0, "MSE": 1}[loss],
                          {"TSGO": 0, "GD": 1}[bbopt], int(rescale[0]), int(rescale[1]),
                          int(train_classes_separately), int(rebuild_caches), float(eta), float(cutoff))
        self.T, self.d, self.C, self.N, self.cap = T, d, Cn, N, cap
        self.slot = cap * d * cap * Cn
        self.sites = np.zeros((T, self.slot))
        self.chi = np.array([W[0].shape[0]] + [t.shape[2] for t in W], dtype=np.int32)
        for j, t in enumerate(W):
            self.sites[j, :t.size] = np.ascontiguousarray(t, dtype=np.float64).reshape(-1)
        ls = [j for j, t in enumerate(W) if t.ndim == 4]
        self.label_site = C.c_int(ls[0])
        self.LE = np.zeros((T, N, cap))
        self.RE = np.zeros((T, N, cap))
        self.gesdd = _gesdd_pointer()

    def _run(self, mode, max_bonds=0, record=False, first_bond=0):
        nb = 2 * (self.T - 1)
        stride = 5 + self.d * self.cap * max(self.C, 1) + 8
        dbg = np.zeros((nb, stride)) if record else None
        secs = np.zeros(2)
        dp = C.POINTER(C.c_double)
        rc = self.lib.orc_run(C.byref(self.o), self.phi.ctypes.data_as(dp), self.label.ctypes.data_as(C.POINTER(C.c_int)),
                              self.counts.ctypes.data_as(C.POINTER(C.c_long)), self.sites.ctypes.data_as(dp),
                              C.c_long(self.slot), self.chi.ctypes.data_as(C.POINTER(C.c_int)), C.byref(self.label_site),
                              self.LE.ctypes.data_as(dp), self.RE.ctypes.data_as(dp), C.c_void_p(self.gesdd),
                              C.c_int(mode), C.c_int(first_bond), C.c_int(max_bonds), dbg.ctypes.data_as(dp) if record else None,
                              C.c_int(stride), secs.ctypes.data_as(dp))
        if rc:
            raise RuntimeError(f"LAPACK dgesdd failed in the C oracle: info={rc}")
        out = {"seconds": secs[0], "bonds": int(secs[1])}
        if record:
            out["bonds_rec"] = [{"loss": r[0], "grad_norm": r[1], "bt_new_norm": r[2], "chi": int(r[3]),
                                 "S": r[5:5 + int(r[4])].copy()} for r in dbg[:int(secs[1])]]
        return out

    def build_caches(self, around_label=False):
        """construct_caches(going_left=true); around_label=True: the environments on both sides of the label site."""
        return self._run(2 if around_label else 1)

    def sweep(self, max_bonds=0, record=False, first_bond=0):
        """Bonds [first_bond, first_bond+max_bonds) of one sweep (all of it by default)."""
        return self._run(0, max_bonds, record, first_bond)

    def get_mps(self):
        W = []
        ls = self.label_site.value
        for j in range(self.T):
            shape = (int(self.chi[j]), self.d, int(self.chi[j + 1])) + ((self.C,) if j == ls else ())
            W.append(self.sites[j, :int(np.prod(shape))].reshape(shape).copy())
        return W
