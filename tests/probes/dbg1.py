import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import ref_numpy as R
from tests.helpers import load_engine, make_problem
import mpstime_jl_amd as mt
N, T, d, chi0, chimax, C = 96, 8, 3, 3, 9, 2
ds, W0 = make_problem(N, T, d, chi0, C, seed=5)
opts = R.SweepOptions(nsweeps=3, chi_max=chimax, eta=0.03)
rec=[]
Wo, info = R.fit(W0, ds, None, opts, record=rec)
print("oracle", info["train_KL_div"])
e = mt.SweepEngine(0)
load_engine(e, ds, W0, opts)
e.build_caches()
print("gpu init", e.eval(0)[:3])
for s in range(3):
    st=e.sweep(); print("gpu sweep", s, e.eval(0)[:3], st, e.get_chi())
# bond by bond variant
e2 = mt.SweepEngine(0)
load_engine(e2, ds, W0, opts)
e2.build_caches()
for s in range(3):
    k=0
    for gl, order in ((True, range(T-2,-1,-1)), (False, range(0,T-1))):
        for lid in order:
            tr=e2.bond_step(lid, gl)
            o=rec[s][k]; k+=1
            if abs(tr['loss']-o['loss'])>1e-9 or tr['chi']!=o['chi']:
                print("MISMATCH sweep",s,"lid",lid,gl,tr['loss'],o['loss'],tr['chi'],o['chi'], tr['eig_sweeps'])
    print("gpu2 sweep", s, e2.eval(0)[:3])
