#!/bin/bash
# Same-box A/B of the four-launch chain (round 6): bench lines with MPST_CHAIN4=0 (six launches per bond) / 1, both register budgets of
# k_bond_tail.  From the repo root on the GPU box: profiles/r06_ab_chain4.sh
out=gpurun_out/r06/ab_chain4
mkdir -p $out
for rb in "" "--rebuild-caches"; do
  tag=${rb:+_rebuild}
  MPST_CHAIN4=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --concurrent 1 $rb > $out/chain6$tag.json 2> $out/chain6$tag.err
  MPST_CHAIN4=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --concurrent 1 $rb > $out/chain4$tag.json 2> $out/chain4$tag.err
  MPST_CHAIN4=1 MPST_TAIL_WIDE=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --concurrent 1 $rb > $out/chain4_wide$tag.json 2> $out/chain4_wide$tag.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r06/ab_chain4/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get("kernels", {})
        print(os.path.basename(f), "value", round(d["value"], 3), "ms", round(d["ms_per_step"], 3), "four", d["launch_chain"].get("four_launch_chain"), "redos", d["launch_chain"].get("tail_redos"),
              "KLD", d["train_KL_div_after"], "acc", d["train_acc_after"], {n: v["avg_us"] for n, v in k.items()})
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-2000:])
PY
