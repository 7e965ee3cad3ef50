#!/bin/bash
# Same-box A/B of the four-launch chain (round 6): bench lines with MPST_CHAIN4=0 (six launches per bond: k_yhat_s, k_grad_s, k_gram_upd,
# k_eig_trivec, k_eig_fin, k_env_split) / 1 (k_grad_s, k_gram_upd, k_eig_trivec, k_bond_tail), with the reference's two cache rebuilds per
# sweep (the headline `value`) and without them; and the cache rebuild in one launch (k_env_walk) against one launch per site.
# From the repo root on the GPU box: profiles/r06_ab_chain4.sh
out=gpurun_out/r06/ab_chain4
mkdir -p $out
for rb in "" "--no-rebuild-caches"; do
  tag=${rb:+_no_rebuild}
  MPST_CHAIN4=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --concurrent 1 $rb > $out/chain6$tag.json 2> $out/chain6$tag.err
  MPST_CHAIN4=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --concurrent 1 $rb > $out/chain4$tag.json 2> $out/chain4$tag.err
done
MPST_ENV_WALK=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --concurrent 1 > $out/chain4_rebuild_per_site_launches.json 2> $out/chain4_persite.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r06/ab_chain4/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get("kernels", {})
        print(os.path.basename(f), "value", round(d["value"], 3), "ms", round(d["ms_per_step"], 3), "rebuild_caches", d["config"]["rebuild_caches"], "four", d["launch_chain"].get("four_launch_chain"),
              "redos", d["launch_chain"].get("tail_redos"), "KLD", d["train_KL_div_after"], "acc", d["train_acc_after"], {n: v["avg_us"] for n, v in k.items()})
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-2000:])
PY
