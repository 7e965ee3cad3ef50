#!/bin/bash
# On the GPU box, from the repo root:  profiles/collect_profiles.sh <tag> <pmc: 0|1> <program + args ...>
# Writes gpurun_out/${ROUND:-r06}/<tag>/: kernel_stats.csv (rocprofv3 --kernel-trace --stats of the command), the command's own
# output, and - with pmc=1 - three separate counter passes aggregated by profiles/aggregate_pmc.py into pmc_counters.json.
set -euo pipefail
tag=$1; pmc=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
prog=$root/$1; shift      # the program is given relative to the repo root; the profiler runs from /tmp
out=$root/gpurun_out/${ROUND:-r06}/$tag
rm -rf $out            # never leave an earlier run's kernel_stats.csv / pmc_counters.json behind a failed pass
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python3 $prog "$@" > $out/stdout_under_rocprof.txt 2> $out/kt.err
db=$(ls $out/kt/*results.db $out/kt/*/*results.db 2>/dev/null | head -1 || true)
test -n "$db" || { echo "collect_profiles: no rocprofv3 results for $tag" >&2; exit 1; }
python3 $root/profiles/rocpd_stats.py $db $out/kernel_stats.csv > /dev/null
test -s $out/kernel_stats.csv
if [ "$pmc" = "1" ]; then
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc$i -o p -- python3 $prog "$@" > $out/pmc$i.stdout 2> $out/pmc$i.err
  done
  f1=$(ls $out/pmc1/*counter_collection.csv $out/pmc1/*/*counter_collection.csv 2>/dev/null | head -1 || true)
  f2=$(ls $out/pmc2/*counter_collection.csv $out/pmc2/*/*counter_collection.csv 2>/dev/null | head -1 || true)
  f3=$(ls $out/pmc3/*counter_collection.csv $out/pmc3/*/*counter_collection.csv 2>/dev/null | head -1 || true)
  cd $root
  test -n "$f1" -a -n "$f2" -a -n "$f3" || { echo "collect_profiles: a counter pass of $tag left no csv" >&2; exit 1; }
  python3 profiles/aggregate_pmc.py $f1 $f2 $f3 $out/pmc_counters.json " ($tag: $(basename $prog) $*)" > $out/pmc_summary.txt 2>&1
  test -s $out/pmc_counters.json
  rm -rf $out/pmc1 $out/pmc2 $out/pmc3
fi
rm -rf $out/kt
ls -la $out
