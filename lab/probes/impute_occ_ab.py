"""Same-box A/B of k_imp_left's two builds: one workgroup per CU (256 VGPRs, no spill) against two (128 VGPRs each, 70-183
spilled).  MPST_IMP_OCC is read once per process, so every arm runs in a child process.
    python lab/probes/impute_occ_ab.py > profiles/r03_impute_occupancy_ab.txt"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import json, os, sys, time
sys.path.insert(0, %r)
import numpy as np
import mpstime_jl_amd as mt
sys.argv = ["x", "--quick"]
import importlib.util
spec = importlib.util.spec_from_file_location("ib", os.path.join(%r, "lab", "probes", "impute_bench.py"))
src = open(spec.origin).read().split("res = []")[0]      # helpers only (random_chain, block_mask, problem)
ns = {"__file__": spec.origin, "__name__": "impute_bench_helpers"}
exec(compile(src, spec.origin, "exec"), ns)
eng = mt.SweepEngine(0)
for (N, T, d, chi, cx, compute) in [(4096, 100, 4, 32, False, "f64"), (4096, 100, 4, 32, False, "f32"), (4096, 100, 4, 32, True, "f64"),
                                      (4096, 100, 4, 32, True, "f32"), (1024, 200, 8, 64, False, "f64"), (1024, 200, 8, 64, True, "f32")]:
    W, xs, gphi, phi, m = ns["problem"](N, T, d, chi, cx)
    lab = np.zeros(N, dtype=np.int32)
    eng.impute_model(W, phi[:8], lab[:8], m[:8], xs, gphi, 0, True, compute=compute)
    best = min(eng.impute_model(W, phi, lab, m, xs, gphi, 0, True, compute=compute)[2] for _ in range(3))
    print(json.dumps(dict(N=N, T=T, d=d, chi=chi, complex=cx, compute=compute, device_ms=round(best * 1e3, 3))), flush=True)
""" % (ROOT, ROOT)
rows = {}
for occ in ("1", "2"):
    out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MPST_IMP_OCC=occ), capture_output=True, text=True, timeout=1200)
    if out.returncode != 0:
        print("arm", occ, "failed:", out.stderr[-1500:])
        sys.exit(1)
    rows[occ] = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
print("k_imp_left: one workgroup per CU (no spill) vs two per CU (spilling build), best of 3 passes, device ms (environment pass + density sweep)")
for a, b in zip(rows["1"], rows["2"]):
    tag = f"N={a['N']} T={a['T']} d={a['d']} chi={a['chi']} {'complex' if a['complex'] else 'real'} {a['compute']}"
    print(f"{tag:48s} occ1 {a['device_ms']:9.3f}   occ2 {b['device_ms']:9.3f}   occ2/occ1 {b['device_ms'] / a['device_ms']:.3f}")
