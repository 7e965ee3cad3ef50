#!/bin/bash
# On the GPU box, from the repo root:  profiles/collect_round.sh [round-tag]   (default r04)
# Re-takes every profile and bench line profiles/README.md lists for the round on the sources of this tree and leaves them under
# gpurun_out/<round>/final/ with the names they carry in profiles/ (copy them over and commit).  The counter file is taken first:
# bench.py quotes it (roofline.traffic) only while its csrc digest matches the tree.
R=${1:-r04}
export ROUND=$R
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=gpurun_out/$R/final
mkdir -p $out
profiles/collect_profiles.sh headline 1 bench.py --steps 3 --warmup 2 --no-cpu-baseline --concurrent 1 > /dev/null 2>&1
cp gpurun_out/$R/headline/kernel_stats.csv $out/${R}_rocprofv3_kernel_stats.csv
cp gpurun_out/$R/headline/pmc_counters.json $out/${R}_pmc_counters.json
grep -v "^$" gpurun_out/$R/headline/stdout_under_rocprof.txt | tail -1 > $out/${R}_bench_n1_under_rocprofv3.json
cp $out/${R}_pmc_counters.json profiles/${R}_pmc_counters.json          # so that the bench line below can quote it
python bench.py --steps 20 --warmup 3 --cpu-full-sweep > $out/${R}_bench_n1.json 2> $out/bench_n1.err
profiles/collect_profiles.sh impute 1 bench.py --workload impute --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cp gpurun_out/$R/impute/kernel_stats.csv $out/${R}_rocprofv3_kernel_stats_impute.csv
cp gpurun_out/$R/impute/pmc_counters.json $out/${R}_pmc_counters_impute.json
python bench.py --workload impute > $out/${R}_bench_impute_configs4.json 2> $out/bench_impute.err
for t in c64 f32; do
  profiles/collect_profiles.sh typed_$t 1 bench.py --dtype $t --steps 1 --warmup 1 --study-bonds 2 > /dev/null 2>&1
  cp gpurun_out/$R/typed_$t/kernel_stats.csv $out/${R}_rocprofv3_kernel_stats_typed_$t.csv
  cp gpurun_out/$R/typed_$t/pmc_counters.json $out/${R}_pmc_counters_typed_$t.json
  grep -v "^$" gpurun_out/$R/typed_$t/stdout_under_rocprof.txt | tail -1 > $out/${R}_bench_typed_${t}_under_rocprofv3.json
  python bench.py --dtype $t > $out/${R}_bench_typed_${t}_N8192_T200_chi64_d8.json 2> $out/bench_$t.err
done
ls -la $out
