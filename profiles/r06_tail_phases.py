#!/usr/bin/env python3
"""In-kernel phase stamps of the four-launch chain's last launch (k_bond_tail) and of the eigensolver at the headline shape, from the
replayed sweep graph (the middle bond of the backward half-sweep; MPST_TAIL_STAMP="lid,going_left" picks another).
From the repo root on the GPU box: python profiles/r06_tail_phases.py [N] [chi]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpstime_jl_amd as mt
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T, d, C = 100, 4, 2
full = bench.make_inputs(N, T, d)
W0 = mt.generate_startingMPS(4, T, d, C, 1234)
eng = mt.SweepEngine(0)
eng.set_options(chi_max=chi, eta=0.01, cutoff=1e-10, update_iters=1, loss="KLD", bbopt="TSGO", rescale=(False, True))
eng.set_dataset(0, full.phi, full.label_index, C)
eng.set_mps(W0)
eng.build_caches()
secs = []
for _ in range(5):
    secs.append(eng.sweep()["seconds"])
if eng.info().get("four_launch_chain"):
    eng.tail_phases()                       # clears the all-workgroups span
secs.append(eng.sweep()["seconds"])
out = {"N": N, "chi": chi, "sweep_ms": [round(1e3 * s, 3) for s in secs], "info": eng.info(), "eig_phases_us": eng.eig_phases()}
if out["info"].get("four_launch_chain"):
    ph = eng.tail_phases()
    out["tail_wg0_chain_host_us"] = ph["host_of_a_chain_job"]
    out["tail_last_wg_plain_tile_us"] = ph["plain_tile"]
    out["bonds_by_candidate_orthogonality"] = ph["bonds_by_candidate_orthogonality"]
    out["tail_all_workgroups_us"] = ph["all_workgroups"]
eng.set_profile(0x7FF)
eng.sweep()
out["event_profile_us_per_launch"] = {k: (round(v[0] / max(v[1], 1), 2), v[1]) for k, v in eng.get_profile().items() if v[1]}
print(json.dumps(out))
