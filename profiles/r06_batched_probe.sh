#!/bin/bash
# profiles/r06_batched_probe.sh <tag> K [N]   (GPU box, repo root): aggregate rate + rocprofv3 kernel stats of the batched chain
set -euo pipefail
tag=$1; K=$2; N=${3:-4096}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r06/$tag
rm -rf $out; mkdir -p $out
python3 $root/profiles/r06_batched_probe.py $K $N 5 > $out/rate.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python3 $root/profiles/r06_batched_probe.py $K $N 2 > $out/under_rocprof.json 2> $out/kt.err
db=$(ls $out/kt/*results.db $out/kt/*/*results.db 2>/dev/null | head -1 || true)
python3 $root/profiles/rocpd_stats.py $db $out/kernel_stats.csv > /dev/null
rm -rf $out/kt
cat $out/rate.json; grep "_b" $out/kernel_stats.csv | cut -c1-120
