"""Sum rocprofv3 --pmc counters per kernel (lab): python pmc_sum.py <dir> [kernel substring]"""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if sub in k:
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k].add(row["Dispatch_Id"])
for k in acc:
    print(k, "dispatches", len(cnt[k]))
    for c, v in sorted(acc[k].items()):
        print("   %-32s %.4g  (per dispatch %.4g)" % (c, v, v / max(1, len(cnt[k]))))
