"""Summarise rocprofv3 --pmc passes: mean counter value per dispatch, per kernel.
usage: pmc_sum.py out.csv dir1 dir2 ..."""
import csv, glob, os, sys, re, json
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            k = re.sub(r"^icm::", "", k)
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
names = sorted({c for k in acc for c in acc[k]})
with open(out, "w") as fo:
    fo.write("kernel,dispatches," + ",".join(names) + "\n")
    for k in sorted(acc):
        n = max(v[1] for v in acc[k].values())
        fo.write('"%s",%d,' % (k, n) + ",".join("%g" % (acc[k][c][0] / acc[k][c][1]) if c in acc[k] else "" for c in names) + "\n")
print(open(out).read())
