"""Start, duration, stream and the gap to the previous launch on the same stream of every kernel of ONE steady-state
sweep, from a rocprofv3 kernel trace ("main queue"):

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --workload S2 --steps 20 --warmup 5 --no-extras --no-roofline
    python tools/sweep_timeline.py OUT
"""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_assoc_runs<false" in n or "k_assoc_group<false, false" in n]
i0, i1 = idx[40], idx[41]
t0 = int(rows[i0]["Start_Timestamp"])
print("sweep period: %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
prev_end = {}
for r in rows[i0:i1 + 1]:
    s, e, st = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"]
    n = r["Kernel_Name"].split("(")[0].replace("void icm::", "").replace("icm::", "")[:40]
    print("%-42s start %7.1f  dur %6.1f  stream %s  gap %s" % (n, (s - t0) / 1e3, (e - s) / 1e3, st, "" if st not in prev_end else "%.1f" % ((s - prev_end[st]) / 1e3)))
    prev_end[st] = e
