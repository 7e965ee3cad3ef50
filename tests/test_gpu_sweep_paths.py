"""The path `bench.py` times - mpst_sweep: one hipGraph replay per sweep, bond k+1's tensor formed inside bond k's last
launch (k_env_split's chained blocks) - against the path every full-size oracle comparison goes through: one mpst_bond_step per
bond (plain stream, tensor assembled from the site tensors by k_bt_assemble).  RealRealHighDimension.jl:727-808."""
import numpy as np
import pytest

import bench
import mpstime_jl_amd as mt
from oracle import ref_numpy as R
from tests.helpers import bond_of

pytestmark = pytest.mark.gpu


def _engine(full, chi, W):
    eng = mt.SweepEngine(0)
    eng.set_options(chi_max=chi, eta=0.01)
    eng.set_dataset(0, full.phi, full.label_index, 2)
    eng.set_mps(W)
    eng.build_caches()
    return eng


def test_graph_sweep_equals_bond_steps_at_the_benchmarked_shape():
    """N = 4096, T = 100, chi = 32, d = 4 (BASELINE configs[2]).  From a common steady-state MPS: two mpst_sweep calls
    against 2 x 198 mpst_bond_step calls - IDENTICAL BITS.  The two paths used to differ in one piece of arithmetic (the
    chained path forms T = bt_new E with the contraction split over four waves and feeds it on from the accumulators, the
    back-split stored the same product from one sequential accumulation); bonds therefore started from states differing in
    the last bits, and the sweep's own dynamics - a truncated SVD is discontinuous where sigma_chi ~ sigma_chi+1 - decided
    how far that grew: 3e-13 ... 6e-3 after ONE sweep depending on the starting MPS (lab/path_dev.py).  The back-split
    now adds its partial products in the chained path's order (gemm_tile_g4), the stored site tensor IS the chained T, and
    the replayed graph equals the stepped sweep bit for bit - for every start tried."""
    N, T, d, chi = 4096, 100, 4, 32
    full = bench.make_inputs(N, T, d)
    W0 = mt.generate_startingMPS(4, T, d, 2, 1234)
    eng = _engine(full, chi, W0)
    try:
        for _ in range(3):                       # steady state: all bulk bonds at chi_max
            eng.sweep()
        Ws = eng.get_mps()
        sub = slice(0, N, 37)
        res = {}
        for path in ("sweep", "steps"):
            eng.set_mps(Ws)
            eng.build_caches()
            per = []
            for s in range(2):
                if path == "sweep":
                    eng.sweep()
                else:
                    for q in range(2 * (T - 1)):
                        eng.bond_step(*bond_of(q, T))
                per.append((eng.get_mps(), eng.eval(0), eng.get_chi()[0].copy()))
            res[path] = per
    finally:
        eng.close()
    for s in (0, 1):
        (Wa, (msa, kla, aca, cfa), chia), (Wb, (msb, klb, acb, cfb), chib) = res["sweep"][s], res["steps"][s]
        assert np.array_equal(chia, chib)
        ya, yb = R.contract_mps(Wa, full.phi[sub]), R.contract_mps(Wb, full.phi[sub])
        dev = np.abs(ya - yb).max() / np.abs(yb).max()
        same = all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(Wa, Wb))
        print(f"sweep {s + 1}: overlaps differ by {dev:.2e} (relative), KLD {kla:.12f} vs {klb:.12f}, bitwise equal: {same}")
        assert same, (s, dev)
        assert kla == klb and msa == msb and aca == acb and np.array_equal(cfa, cfb)


@pytest.mark.parametrize("K", [4, 12, 20])
def test_batched_sweep_of_independent_fits_is_bit_identical(K):
    """mpst_sweep_batch: K fits of one shape (different data, starting MPS, eta and cutoff) advanced by one launch chain give the
    bits of K separate mpst_sweep calls, sweep after sweep; a fit of another shape is refused.  K = 12 and 20: more eigensolver
    workgroups than CUs, so every workgroup takes 2 (3, unevenly) eigenpairs (k_eig_trivec_bm)."""
    import mpstime_jl_amd as mt
    from tests.helpers import make_problem
    probs = [make_problem(256, 12, 4, 4, 2, seed=50 + k) for k in range(K)]
    etas = [[0.05, 0.02, 0.05, 0.1][k % 4] for k in range(K)]

    def fresh(k, chi=12):
        e = mt.SweepEngine(0)
        e.set_batch_hint(K)                 # the share count of the gradient blocks belongs to the context: solo and batched sweeps agree bit for bit
        e.set_options(chi_max=chi, eta=etas[k], cutoff=1e-10 if k != 2 else 1e-8)
        ds, W = probs[k]
        e.set_dataset(0, ds.phi, ds.label_index, 2)
        e.set_mps(W)
        e.build_caches()
        return e

    solo = [fresh(k) for k in range(K)]
    bat = [fresh(k) for k in range(K)]
    try:
        for sweep in range(3):
            for e in solo:
                e.sweep()
            st = mt.sweep_batch(bat)
            assert len(st) == K and all(s["eig_fallbacks"] == 0 for s in st)
            for a, b in zip(solo, bat):
                for ta, tb in zip(a.get_mps(), b.get_mps()):
                    assert np.array_equal(ta, tb)
                assert a.eval(0)[:3] == b.eval(0)[:3]
        # a single sweep() on a batched context continues from the same state
        solo[1].sweep()
        bat[1].sweep()
        assert all(np.array_equal(x, y) for x, y in zip(solo[1].get_mps(), bat[1].get_mps()))
        other = fresh(0, chi=8)
        try:
            with pytest.raises(mt.MPSTError, match="differs in shape"):
                mt.sweep_batch([bat[0], other])
        finally:
            other.close()
    finally:
        for e in solo + bat:
            e.close()


def test_fits_dealt_over_groups_are_bit_identical_and_destroyed_members_do_not_alias():
    """mpst_sweep_batch_multi: K fits dealt over G groups (on a node: one group per device; here two groups on the one GPU), every
    group one launch chain on its own host thread, no collective: the bits of K separate mpst_sweep calls.  A group may hold fits
    of another shape than its neighbour.  Then the graph-key regression: a member is destroyed and recreated (the allocator may
    hand back its address) - the lead must not replay the graph captured for the dead context."""
    import mpstime_jl_amd as mt
    from tests.helpers import make_problem
    K = 6
    shapes = [(256, 12, 12)] * 3 + [(192, 10, 8)] * 3          # group 0 / group 1: different N, T, chi_max
    probs = [make_problem(shapes[k][0], shapes[k][1], 4, 4, 2, seed=70 + k) for k in range(K)]

    def fresh(k):
        e = mt.SweepEngine(0)
        e.set_batch_hint(3)
        e.set_options(chi_max=shapes[k][2], eta=0.05)
        ds, W = probs[k]
        e.set_dataset(0, ds.phi, ds.label_index, 2)
        e.set_mps(W)
        e.build_caches()
        return e

    solo = [fresh(k) for k in range(K)]
    bat = [fresh(k) for k in range(K)]
    groups = [0, 0, 0, 1, 1, 1]
    try:
        for sweep in range(2):
            for e in solo:
                e.sweep()
            st = mt.sweep_batch_multi(bat, groups)
            assert len(st) == K and all(s["eig_fallbacks"] == 0 for s in st)
            for a, b in zip(solo, bat):
                assert all(np.array_equal(x, y) for x, y in zip(a.get_mps(), b.get_mps()))
        with pytest.raises(mt.MPSTError, match="differs in shape"):
            mt.sweep_batch_multi(bat, [0, 0, 0, 0, 1, 1])       # a group mixes the two shapes
        # destroy and recreate a non-lead member of group 0 with the same call sequence, then batch again
        for _ in range(3):
            bat[1].close()
            solo[1].close()
            bat[1], solo[1] = fresh(1), fresh(1)
            solo[0].sweep(); solo[1].sweep(); solo[2].sweep()
            mt.sweep_batch(bat[:3])
            for a, b in zip(solo[:3], bat[:3]):
                assert all(np.array_equal(x, y) for x, y in zip(a.get_mps(), b.get_mps()))
    finally:
        for e in solo + bat:
            e.close()


def test_three_groups_of_unequal_size_with_a_member_recreated_between_calls():
    """mpst_sweep_batch_multi with three groups of 4, 3 and 2 fits (three shapes), every group a launch chain on its own host thread
    (their graphs captured one after the other before any thread runs): the bits of separate sweeps; a member of the middle group
    destroyed and recreated between two calls; a context named in two groups is refused."""
    import mpstime_jl_amd as mt
    from tests.helpers import make_problem
    sizes = [4, 3, 2]
    shapes = [(256, 12, 12), (192, 10, 8), (160, 9, 16)]
    groups, shp = [], []
    for g, k in enumerate(sizes):
        groups += [g] * k
        shp += [shapes[g]] * k
    K = len(groups)
    probs = [make_problem(shp[k][0], shp[k][1], 4, 4, 2, seed=90 + k) for k in range(K)]

    def fresh(k):
        e = mt.SweepEngine(0)
        e.set_batch_hint(sizes[groups[k]])
        e.set_options(chi_max=shp[k][2], eta=[0.05, 0.02, 0.1][k % 3])
        ds, W = probs[k]
        e.set_dataset(0, ds.phi, ds.label_index, 2)
        e.set_mps(W)
        e.build_caches()
        return e

    solo = [fresh(k) for k in range(K)]
    bat = [fresh(k) for k in range(K)]
    try:
        for call in range(3):
            if call == 1:                       # member 5 (the middle group's second fit) goes and comes back: same data, same start
                bat[5].close()
                solo[5].close()
                bat[5], solo[5] = fresh(5), fresh(5)
                for e in (bat[5], solo[5]):     # ... brought to where its group is
                    e.sweep()
            for e in solo:
                e.sweep()
            st = mt.sweep_batch_multi(bat, groups)
            assert len(st) == K and all(s["eig_fallbacks"] == 0 for s in st)
            for k, (a, b) in enumerate(zip(solo, bat)):
                assert all(np.array_equal(x, y) for x, y in zip(a.get_mps(), b.get_mps())), (call, k)
        with pytest.raises(mt.MPSTError, match="appears twice"):
            mt.sweep_batch_multi(bat[:4] + [bat[0]] + bat[5:], groups)
    finally:
        for e in solo + bat:
            e.close()
