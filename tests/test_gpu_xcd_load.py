"""The XCD-local, barrier-free tridiagonalisation (k_bt_coop<512, SC_XCD>, mpst_eig_blocked.hip) under UNEVEN load: while a
d = 8, chi = 64 sweep (d*chi = 512) repeats 50 times, two other contexts stream bond GEMMs at N = 32768 from their own
threads, so the 32 cooperating workgroups start at different times, share their CUs with other kernels and find the L2
and the memory system busy.  Every repetition must give the bits of the launch-per-step path (MPST_BT_NO_COOP=1, run in a
child process): the exchange's self-validating entries, the placement roll call and the fall-backs may cost time, never a bit.
MI355X_MICROARCH.md, hand-off rules: 'test every hand-off under uneven load, checking every word'."""
import hashlib
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import mpstime_jl_amd as mt
from oracle import ref_numpy as R
from tests.helpers import load_engine, make_problem

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = (48, 6, 8, 24, 64)            # N, T, d, chi_init, chi_max: bonds up to 8 * 64 = 512


def _digest_of_sweeps(reps):
    N, T, d, chi0, chimax = SHAPE
    ds, W0 = make_problem(N, T, d, chi0, 1, seed=11)
    opts = R.SweepOptions(nsweeps=1, chi_max=chimax, eta=0.05, loss_grad="KLD", bbopt="TSGO")
    eng = mt.SweepEngine(0)
    digests = []
    try:
        load_engine(eng, ds, W0, opts)
        for _ in range(reps):
            eng.set_mps(W0)
            eng.build_caches()
            eng.sweep()
            h = hashlib.sha256()
            for t in eng.get_mps():
                h.update(np.ascontiguousarray(t).tobytes())
            digests.append(h.hexdigest())
        info = eng.info()
    finally:
        eng.close()
    return digests, info


def test_xcd_local_tridiagonalisation_under_uneven_load():
    # reference bits: the launch-per-step path, alone on the GPU, in a child process (the switch is read once per process)
    env = dict(os.environ, MPST_BT_NO_COOP="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    code = ("import sys; sys.path.insert(0, %r); from tests.test_gpu_xcd_load import _digest_of_sweeps; "
            "d, i = _digest_of_sweeps(2); assert d[0] == d[1]; print('DIGEST', d[0])" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    ref = [ln.split()[1] for ln in out.stdout.splitlines() if ln.startswith("DIGEST")][0]

    # load generators: two contexts streaming the bond GEMMs of N = 32768 series (chi = 32, d = 4)
    import bench
    full = bench.make_inputs(32768, 12, 4)
    stop = threading.Event()
    gens = []
    for g in range(2):
        e = mt.SweepEngine(0)
        e.set_options(chi_max=32, eta=0.01)
        e.set_dataset(0, full.phi, full.label_index, 2)
        e.set_mps(mt.generate_startingMPS(4, 12, 4, 2, 100 + g))
        e.build_caches()
        e.sweep()
        gens.append(e)
    counts = [0, 0]

    def run(k):
        while not stop.is_set():
            gens[k].sweep()
            counts[k] += 1

    threads = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    try:
        digests, info = _digest_of_sweeps(50)
    finally:
        stop.set()
        for t in threads:
            t.join()
        for e in gens:
            e.close()
    print(f"50 sweeps under load ({counts} background sweeps): persistent_tridiag_aborts={info['persistent_tridiag_aborts']} "
          f"xcd_local_misplaced={info['xcd_local_misplaced']} library_eig_fallbacks={info['library_eig_fallbacks']}")
    assert info["large_bond"]
    assert min(counts) >= 1                          # the load really ran beside the solves
    bad = [i for i, dg in enumerate(digests) if dg != ref]
    assert not bad, (bad, info)
