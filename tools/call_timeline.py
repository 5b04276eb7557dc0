"""Kernels AND copies of one steady-state sweep / drop-in call on every queue, from a rocprofv3 --kernel-trace [--memory-copy-trace]
directory: start, duration, end (us from the sweep's k_assoc_group) and queue.  python tools/call_timeline.py OUT [index of the sweep, default -3]"""
import csv, sys, glob
d = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void icm::", "").replace("icm::", "")[:44], "q" + r.get("Queue_Id", "?")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", ""))[:30], "copy"))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_assoc_group<false, false")]
i0, i1 = idx[which], idx[which + 1]
t0 = rows[i0][0]
print("period %.1f us" % ((rows[i1][0] - t0) / 1e3))
for s, e, n, q in rows[i0:i1 + 1]:
    print("%-46s start %7.1f  dur %6.1f  end %7.1f  %s" % (n, (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, q))
