"""Sensitivity study (test infrastructure): the C oracle against ITSELF after a 1e-13 / 1e-10 relative
perturbation of the initial label tensor, BASELINE.json configs[2] shape.  Output recorded in DESIGN.md:
the first sweep amplifies 1e-13 to O(1) differences in per-bond loss, i.e. free-running trajectories of
any two fp64 implementations (including the reference on two machines, docs/src/classification.md:57-60)
decorrelate within one sweep at this size; parity there is checked bond by bond from a common state."""
import sys; sys.path.insert(0,'.')
import numpy as np, time
import mpstime_jl_amd as mt
from oracle.c_oracle import COracle
from oracle import ref_numpy as R
import bench
N=int(sys.argv[1]) if len(sys.argv)>1 else 4096
full = bench.make_inputs(N, 100, 4); W0 = mt.generate_startingMPS(4, 100, 4, 2, 1234)
ds = R.EncodedSet(full.phi, full.label_index, full.class_distribution)
res=[]
for eps in (0.0, 1e-13, 1e-10):
    W=[t.copy() for t in W0]
    rng=np.random.default_rng(5)
    W[-1] = W[-1]*(1+eps*rng.standard_normal(W[-1].shape))
    co = COracle(W, full.phi, full.label_index, full.class_distribution, 32, eta=0.01, rebuild_caches=False)
    co.build_caches()
    t0=time.time(); rec = co.sweep(record=True)["bonds_rec"]; 
    kld = R.mse_loss_acc(co.get_mps(), ds)[1]
    res.append((eps, [b["loss"] for b in rec], [b["chi"] for b in rec], kld))
    print("eps", eps, "time", round(time.time()-t0,1), "KLD after sweep", kld, flush=True)
base=res[0]
for eps, loss, chi, kld in res[1:]:
    d=np.abs(np.array(loss)-np.array(base[1]))/np.maximum(1,np.abs(base[1]))
    print("eps",eps,"bond-loss rel diff at bonds 10,50,98,150,197:", d[[10,50,98,150,197]], "max", d.max(), "chi diffs", int(np.sum(np.array(chi)!=np.array(base[2]))), "KLD diff", kld-base[3])
