import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import ref_numpy as R
from tests.helpers import load_engine
from tests.test_oracle import load_golden
import mpstime_jl_amd as mt
g, ds, W0, opts = load_golden("tests/golden/config1_trendy_sine.npz")
T = ds.phi.shape[1]
for alg in (0, 1):
    eng = mt.SweepEngine(0)
    load_engine(eng, ds, W0, opts, svd_alg=alg)
    eng.build_caches()
    k = 0; out=[]
    W=[t.copy() for t in W0]; LE,RE=R.construct_caches(W,ds.phi,True)
    for sw in range(1):
        for gl, order in ((True, range(T-2,-1,-1)), (False, range(0,T-1))):
            if not gl: LE,RE=R.construct_caches(W,ds.phi,False)
            for lid in order:
                tr = eng.bond_step(lid, gl)
                tro={}; R.bond_step(W,LE,RE,lid,ds,opts,gl,tro)
                So=tro["S"]; 
                e1=abs(tr["loss"]-g["bond_loss"][k])/max(1,abs(g["bond_loss"][k])); e2=abs(tr["grad_norm"]-g["bond_grad_norm"][k])/g["bond_grad_norm"][k]
                # full spectrum gap at truncation boundary from oracle: need all S -> recompute
                out.append((k,lid,gl,e1,e2,tr["chi"],int(g["bond_chi"][k]), tr["eig_sweeps"]))
                k+=1
    print("alg",alg)
    for o in out:
        if o[0]%6==0 or o[4]>1e-7: print(o)
    eng.close()
