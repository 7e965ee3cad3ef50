set -e
export ROUND=r06
profiles/collect_profiles.sh batched8 1 profiles/r06_batched_probe.py 8 4096 1 > gpurun_out/r06_c8.log 2>&1 || tail -5 gpurun_out/r06_c8.log
profiles/collect_profiles.sh batched32 1 profiles/r06_batched_probe.py 32 4096 1 > gpurun_out/r06_c32.log 2>&1 || tail -5 gpurun_out/r06_c32.log
root=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $root/gpurun_out/r06/tcc8 -o p -- python3 $root/profiles/r06_batched_probe.py 8 4096 1 > /dev/null 2> $root/gpurun_out/r06/tcc8.err || tail -3 $root/gpurun_out/r06/tcc8.err
cd $root
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/r06/tcc8/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'][:40]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
for k,v in acc.items():
    if '_b' in k: print(k, {c:round(x/n[(k,c)]) for c,x in v.items()})
PY
rm -rf gpurun_out/r06/tcc8
python3 - <<'PY'
import json
for t in ('batched8','batched32'):
    j=json.load(open(f'gpurun_out/r06/{t}/pmc_counters.json'))
    for k,v in j['kernels'].items():
        if '_b' in k: print(t,k[:40],v.get('launches'),v.get('hbm_bytes_per_launch_corrected'),v.get('mfma_util'))
PY
