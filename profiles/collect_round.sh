#!/bin/bash
# On the GPU box, from the repo root:  profiles/collect_round.sh [round-tag]   (default r06)
# Re-takes every profile and bench line profiles/README.md lists for the round on the sources of this tree and leaves them under
# gpurun_out/<round>/final/ with the names they carry in profiles/ (copy them over and commit).  The counter file is taken first:
# bench.py quotes it (roofline.traffic) only while its csrc digest matches the tree.
set -euo pipefail
R=${1:-r06}
export ROUND=$R
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=gpurun_out/$R/final
mkdir -p $out
profiles/collect_profiles.sh headline 1 bench.py --steps 3 --warmup 2 --no-cpu-baseline --concurrent 1 > $out/collect_headline.log 2>&1
cp gpurun_out/$R/headline/kernel_stats.csv $out/${R}_rocprofv3_kernel_stats.csv
cp gpurun_out/$R/headline/pmc_counters.json $out/${R}_pmc_counters.json
grep -v "^$" gpurun_out/$R/headline/stdout_under_rocprof.txt | tail -1 > $out/${R}_bench_n1_under_rocprofv3.json
# the bench line quotes the counters taken a moment ago on this very tree (nothing is written into profiles/ from here)
python bench.py --steps 20 --warmup 3 --pmc-file $out/${R}_pmc_counters.json > $out/${R}_bench_n1.json 2> $out/bench_n1.err
# same-box A/B at N = 4096: one persistent launch for yhat + gradient (k_bond_fused + k_fused_reduce, MPST_B2=0) against the sliced pair
MPST_B2=0 profiles/collect_profiles.sh headline_fused 0 bench.py --steps 3 --warmup 2 --no-cpu-baseline --concurrent 1 > $out/collect_fused.log 2>&1
cp gpurun_out/$R/headline_fused/kernel_stats.csv $out/${R}_rocprofv3_kernel_stats_bond_fused_ab.csv
grep -v "^$" gpurun_out/$R/headline_fused/stdout_under_rocprof.txt | tail -1 > $out/${R}_bench_n1_bond_fused_ab_under_rocprofv3.json
# same-box A/B of the four-launch chain (k_bond_tail) against the six-launch chain it replaces, with and without the cache rebuilds
profiles/r06_ab_chain4.sh > $out/${R}_ab_chain4.txt 2>&1 || true
python profiles/r06_tail_phases.py > $out/${R}_tail_phases.json 2> $out/tail_phases.err || true
# the same kernels when the chip is fed: 8 and 32 independent fits per launch (mpst_sweep_batch) - per-kernel durations, HBM bytes and MFMA
# utilisation of the _b variants, each batch size on its own (profiles/r06_batched_probe.py runs nothing but the batched chain)
for K in 8 32; do
  profiles/collect_profiles.sh batched$K 1 profiles/r06_batched_probe.py $K 4096 2 > $out/collect_batched$K.log 2>&1
  cp gpurun_out/$R/batched$K/kernel_stats.csv $out/${R}_rocprofv3_kernel_stats_batched$K.csv
  cp gpurun_out/$R/batched$K/pmc_counters.json $out/${R}_pmc_counters_batched$K.json
  python profiles/r06_batched_probe.py $K 4096 5 > $out/${R}_batched_rate_$K.json 2> $out/batched_rate_$K.err
done
if [ -f profiles/ubench/b2dbg/libmpstime_hip_b2dbg.so ]; then
  for K in 8 32; do python profiles/ubench/b2_stamps.py $K > $out/${R}_b2_stamps_k$K.json 2> $out/b2_stamps_$K.err || true; done
fi
# BASELINE configs[3]'s size on one GPU (the persistent k_bond_fused pair): kernel stats, counters, bench line; and the sliced pair at that size
profiles/collect_profiles.sh n32768 1 bench.py --N 32768 --steps 3 --warmup 2 --no-cpu-baseline --concurrent 1 > $out/collect_n32768.log 2>&1
cp gpurun_out/$R/n32768/kernel_stats.csv $out/${R}_rocprofv3_kernel_stats_n32768.csv
cp gpurun_out/$R/n32768/pmc_counters.json $out/${R}_pmc_counters_n32768.json
MPST_B2=1 MPST_CHAIN4=0 python bench.py --N 32768 --steps 5 --warmup 2 --no-cpu-baseline --concurrent 1 > $out/${R}_bench_n32768_sliced_ab.json 2> $out/bench_n32768_sliced.err || true
# two ranks sharing the one GPU of the box (one-shot all-reduce; RCCL refuses two ranks on a device): the N > 1 line with its multi_gpu keys
MPST_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --fits-per-gpu 8 > $out/${R}_bench_gpus2_shared_gpu.json 2> $out/bench_gpus2.err || true
profiles/collect_profiles.sh impute 1 bench.py --workload impute --steps 1 --warmup 1 --no-cpu-baseline > $out/collect_impute.log 2>&1
cp gpurun_out/$R/impute/kernel_stats.csv $out/${R}_rocprofv3_kernel_stats_impute.csv
cp gpurun_out/$R/impute/pmc_counters.json $out/${R}_pmc_counters_impute.json
python bench.py --workload impute > $out/${R}_bench_impute_configs4.json 2> $out/bench_impute.err
for t in c64 f32; do
  profiles/collect_profiles.sh typed_$t 1 bench.py --dtype $t --classes 1 --steps 1 --warmup 2 --study-bonds 2 > $out/collect_$t.log 2>&1
  cp gpurun_out/$R/typed_$t/kernel_stats.csv $out/${R}_rocprofv3_kernel_stats_typed_$t.csv
  cp gpurun_out/$R/typed_$t/pmc_counters.json $out/${R}_pmc_counters_typed_$t.json
  grep -v "^$" gpurun_out/$R/typed_$t/stdout_under_rocprof.txt | tail -1 > $out/${R}_bench_typed_${t}_under_rocprofv3.json
  python bench.py --dtype $t --classes 1 > $out/${R}_bench_typed_${t}_N8192_T200_chi64_d8.json 2> $out/bench_$t.err
done
python bench.py --dtype f32 --classes 2 > $out/${R}_bench_typed_f32_two_classes.json 2> $out/bench_f32c2.err
python bench.py --N 8192 --T 200 --chi 64 --d 8 --steps 3 --warmup 2 --no-cpu-baseline --concurrent 1 > $out/${R}_bench_chi64_d8_N8192_T200.json 2> $out/bench_big.err
python bench.py --N 32768 --steps 5 --warmup 2 --no-cpu-baseline --concurrent 1 > $out/${R}_bench_n32768.json 2> $out/bench_n32768.err
for t in c64 f32; do python bench.py --dtype $t --study > $out/${R}_tolerance_study_$t.json 2> $out/study_$t.err; done
python profiles/r06_determinism.py 10 2>/dev/null | grep -v amdgpu.ids > $out/${R}_determinism_soak.log || true
ls -la $out
