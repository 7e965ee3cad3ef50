"""The C-ABI shared library loads on a machine without a GPU, exports every symbol that
include/mpstime_hip.h declares, and the ctypes mirror of its structs has the C layout."""
import ctypes
import os
import re
import subprocess
import tempfile

import pytest

import mpstime_jl_amd as mt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mpstime_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mpst_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = mt._lib.load()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in mt._lib.SYMBOLS, f"{n} has no ctypes binding"
    assert sorted(mt._lib.SYMBOLS) == names
    assert lib.mpst_version() == mt._lib.ABI_VERSION == 2        # MPST_ABI_VERSION of the header


def test_struct_layouts_match_the_header():
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "mpstime_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(mpst_options), offsetof(mpst_options, eta), offsetof(mpst_options, cutoff),
         sizeof(mpst_sweep_stats), sizeof(mpst_bond_debug), offsetof(mpst_bond_debug, spectrum), offsetof(mpst_bond_debug, chi_new));
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(mpst_impute_opts), offsetof(mpst_impute_opts, rejection_threshold),
         sizeof(mpst_impute_model), offsetof(mpst_impute_model, dtype), offsetof(mpst_impute_model, site),
         offsetof(mpst_impute_model, label_idx), sizeof(mpst_encode_opts));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        vals = [int(x) for x in subprocess.check_output([exe]).split()]
    L = mt._lib
    assert vals == [ctypes.sizeof(L.mpst_options), L.mpst_options.eta.offset, L.mpst_options.cutoff.offset,
                    ctypes.sizeof(L.mpst_sweep_stats), ctypes.sizeof(L.mpst_bond_debug),
                    L.mpst_bond_debug.spectrum.offset, L.mpst_bond_debug.chi_new.offset,
                    ctypes.sizeof(L.ImputeOpts), L.ImputeOpts.rejection_threshold.offset, ctypes.sizeof(L.ImputeModel),
                    L.ImputeModel.dtype.offset, L.ImputeModel.site.offset, L.ImputeModel.label_idx.offset,
                    ctypes.sizeof(L.mpst_encode_opts)]


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mt.MPSTError, match="no HIP device"):
        mt.SweepEngine(0)


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "mpstime.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".jl")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("# oracle", ""), f"{f} refers to the oracle"


def test_collective_library_is_bound_at_run_time_not_linked():
    """RCCL is opened with dlopen when first needed (the copy already in the process wins), so the shared library
    carries no DT_NEEDED entry for it; the probe reports which file was bound and both version numbers."""
    so = os.path.join(ROOT, "mpstime.jl_amd", "csrc", "libmpstime_hip.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "librccl" not in needed
    info = mt.comm_library()
    assert info["ok"], info
    assert "librccl" in info["library"]
    assert info["version"] // 10000 == info["built_against"] // 10000      # same major: the ABI the loader accepts
