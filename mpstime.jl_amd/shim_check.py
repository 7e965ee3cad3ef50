"""Companion of julia/roundtrip_check.jl --shim: the fit the Julia shim is asked to reproduce.

    python -m mpstime_jl_amd.shim_check          (on a GPU box)

writes julia/shim_inputs.npz (raw series, labels, options, the starting MPS in (left bond, site, right bond[, label]) order)
and julia/shim_expected_digest.txt (content digest of the MPS this Python mirror trains from them).  A maintainer with Julia
then runs `julia --project=<MPSTime.jl> julia/roundtrip_check.jl <MPSTime.jl> --shim`: equal digests mean the ccall marshalling
of MPSTimeHIP.jl agrees with the tested ctypes binding bit for bit (same library, same inputs, same launch sequence).
The encoding on the Julia side is MPSTime.jl's own; it is pinned to 3e-15 against this package's (tests/test_reference_fixture.py),
so the digests can differ in the last bits if the two encoders round differently - the script prints both for that reason.
"""
import os

import numpy as np


def main():
    import mpstime_jl_amd as mt
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "julia")
    rng = np.random.default_rng(7)
    T, N, d, chi_max, chi_init, nsweeps, eta = 24, 200, 4, 12, 4, 2, 0.05
    X1, _ = mt.trendy_sine(T, N // 2, period=(12.0, 15.0), slope=[-3.0, 0.0, 3.0], sigma=0.1, rng=rng)
    X2, _ = mt.trendy_sine(T, N // 2, period=(16.0, 19.0), slope=[-3.0, 0.0, 3.0], sigma=0.1, rng=rng)
    X = np.concatenate([X1, X2])
    y = np.concatenate([np.ones(N // 2, dtype=np.int64), 2 * np.ones(N // 2, dtype=np.int64)])
    opts = mt.MPSOptions(d=d, chi_max=chi_max, nsweeps=nsweeps, eta=eta, chi_init=chi_init, verbosity=-1, encoding="Legendre_No_Norm")
    W0 = mt.generate_startingMPS(chi_init, T, d, 2, 1234)
    trained, info, _ = mt.fitMPS(X, y, opts=opts, W=W0)
    digest = mt.mps_content_digest(trained.mps)
    arrs = {f"mps0_{j}": np.ascontiguousarray(t) for j, t in enumerate(W0)}
    np.savez(os.path.join(here, "shim_inputs.npz"), X_train=X, y_train=y, d=d, chi_max=chi_max, nsweeps=nsweeps, eta=eta, chi_init=chi_init, **arrs)
    with open(os.path.join(here, "shim_expected_digest.txt"), "w") as f:
        f.write(digest)
    print("train KLD per sweep:", info["train_KL_div"])
    print("digest:", digest)


if __name__ == "__main__":
    main()
