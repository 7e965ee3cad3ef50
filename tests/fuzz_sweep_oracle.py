"""Random shapes through the engine's bond update against the NumPy oracle, teacher forced (every bond from the oracle's state), both launch
chains; not collected by pytest; run by hand on a GPU box: python tests/fuzz_sweep_oracle.py [seed] [cases]."""
import os
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import mpstime_jl_amd as mt                             # noqa: E402
from oracle import ref_numpy as R                       # noqa: E402
from tests.helpers import load_engine, make_problem    # noqa: E402


def one(N, T, d, chi, C, loss, sep, chain4):
    ds, W0 = make_problem(N, T, d, min(4, chi), C, seed=7 + N + T, balanced=False)
    opts = R.SweepOptions(nsweeps=1, chi_max=chi, eta=0.05, loss_grad=loss, bbopt="TSGO" if loss == "KLD" else "GD", train_classes_separately=sep)
    os.environ["MPST_CHAIN4"] = "1" if chain4 else "0"
    eng = mt.SweepEngine(0)
    try:
        load_engine(eng, ds, W0, opts)
        W = [t.copy() for t in W0]
        LE, RE = R.construct_caches(W, ds.phi, True)
        worst = [0.0, 0.0, 0.0, 0.0]
        for going_left, order in ((True, range(T - 2, -1, -1)), (False, range(0, T - 1))):
            if not going_left:
                LE, RE = R.construct_caches(W, ds.phi, False)
            for lid in order:
                eng.set_mps(W)
                eng.build_caches()
                tr = eng.bond_step(lid, going_left)
                tro = {}
                R.bond_step(W, LE, RE, lid, ds, opts, going_left, tro)
                assert tr["chi"] == tro["chi"], ("chi", lid, tr["chi"], tro["chi"])
                worst[0] = max(worst[0], abs(tr["loss"] - tro["loss"]) / max(1.0, abs(tro["loss"])))
                worst[1] = max(worst[1], abs(tr["grad_norm"] - tro["grad_norm"]) / tro["grad_norm"])
                worst[2] = max(worst[2], np.abs(tr["S"][:tr["chi"]] - tro["S"]).max() / tro["S"][0])
                yo, yg = R.contract_mps(W, ds.phi), R.contract_mps(eng.get_mps(), ds.phi)
                worst[3] = max(worst[3], np.abs(yo - yg).max() / np.abs(yo).max())
        assert worst[0] < 1e-10 and worst[1] < 1e-7 and worst[2] < 1e-8 and worst[3] < 1e-7, worst
    finally:
        eng.close()
        os.environ.pop("MPST_CHAIN4", None)


def main(seed=0, cases=16):
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(cases):
        d = int(rng.choice([2, 3, 4, 5, 8, 11, 16]))
        chi = int(rng.choice([c for c in (2, 3, 5, 8, 12, 17, 24, 32) if d * c <= 128]))
        N = int(rng.choice([20, 65, 130, 300]))
        T = int(rng.integers(3, 9))
        C = int(rng.integers(1, 4))
        loss = str(rng.choice(["KLD", "KLD", "MSE"]))
        sep = bool(rng.integers(0, 2)) and loss == "KLD"
        chain4 = bool(rng.integers(0, 2))
        try:
            one(N, T, d, chi, C, loss, sep, chain4)
            print("ok  ", N, T, d, chi, C, loss, sep, chain4, flush=True)
        except BaseException as e:
            bad += 1
            print("FAIL", N, T, d, chi, C, loss, sep, chain4, repr(e)[:300], flush=True)
    print("failures:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(x) for x in sys.argv[1:3])) else 0)
