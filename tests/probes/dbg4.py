import sys; sys.path.insert(0,'.')
import numpy as np, time
import mpstime_jl_amd as mt
from oracle.c_oracle import COracle
from oracle import ref_numpy as R
import bench
full = bench.make_inputs(4096, 100, 4); W0 = mt.generate_startingMPS(4, 100, 4, 2, 1234)
ds = R.EncodedSet(full.phi, full.label_index, full.class_distribution)
co = COracle(W0, full.phi, full.label_index, full.class_distribution, 32, eta=0.01, rebuild_caches=False)
co.build_caches()
eng = mt.SweepEngine(0); eng.set_options(chi_max=32, eta=0.01); eng.set_dataset(0, full.phi, full.label_index, 2); eng.set_mps(W0); eng.build_caches()
sub = slice(0,4096,8)
dsub = R.EncodedSet(full.phi[sub], full.label_index[sub], None)
for sw in range(3):
    t0=time.time(); ref = co.sweep(record=True)["bonds_rec"]; t1=time.time()
    tr = [eng.bond_step(l, True) for l in range(98,-1,-1)] + [eng.bond_step(l, False) for l in range(0,99)]
    dl = max(abs(a["loss"]-b["loss"])/max(1,abs(b["loss"])) for a,b in zip(tr,ref))
    dchi = [a["chi"]-b["chi"] for a,b in zip(tr,ref)]
    mo = R.mse_loss_acc(co.get_mps(), ds); mg = eng.eval(0)
    Wg = eng.get_mps(); Wc = co.get_mps()
    yo, yg = R.contract_mps(Wc, full.phi[sub]), R.contract_mps(Wg, full.phi[sub])
    print("sweep", sw, "oracle s", round(t1-t0,1), "max rel bond-loss diff", dl, "chi diffs nonzero", sum(1 for x in dchi if x), "max", max(map(abs,dchi)),
          "KLD oracle", mo[1], "engine", mg[1], "rel", abs(mo[1]-mg[1])/abs(mo[1]), "acc", mo[2], mg[2], "overlap rel diff", np.abs(yo-yg).max()/np.abs(yo).max(),
          "min P kept (oracle)", min(b["S"][b["chi"]-1]**2 for b in ref[5:-5]))
