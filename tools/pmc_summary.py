"""Summarise rocprofv3 --pmc passes: mean counter value per dispatch, per kernel.
usage: pmc_summary.py out.csv dir1 dir2 ...
       pmc_summary.py --traffic profiles/traffic.json WORKLOAD traffic_summary.csv [BUILD_ID]
           -> HBM bytes per launch by bench.py's kernel names (what bench.py reports as roofline.traffic):
              (2 x FETCH_SIZE + WRITE_SIZE) KB, FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950"""
import csv, glob, os, sys, re, json

BENCH_NAMES = (("k_assoc_runs<false", "k_assoc_runs"), ("k_assoc_group<false, false", "k_assoc_group"), ("k_chunk_l1", "k_chunk_l1"), ("k_chunk_l2", "k_chunk_l2"),
               ("k_lm_l3", "k_lm_l3"), ("k_rec_push", "k_rec_push"), ("k_pose_moments", "k_pose_moments"),
               ("k_solve_m_fused", "k_solve"), ("k_scan_", "k_scan"),
               ("k_neigh_table", "k_neigh_table"), ("k_fl_", "k_filtrar"), ("k_pose_rot", "k_pose_rot"))

if len(sys.argv) > 1 and sys.argv[1] == "--traffic":
    dst, workload, src = sys.argv[2:5]
    build = sys.argv[5] if len(sys.argv) > 5 else None    # icm_build_id() of the library that was profiled
    tj = json.load(open(dst)) if os.path.exists(dst) else {}
    per = {}
    for r in csv.DictReader(open(src)):
        for pat, name in BENCH_NAMES:
            if r["kernel"].startswith(pat) and r["FETCH_SIZE"] and r["WRITE_SIZE"]:
                per[name] = per.get(name, 0) + int(round((2 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024))
                break
    tj[workload] = per
    tj.setdefault("_source", {})[workload] = os.path.join("profiles", os.path.basename(src))
    tj.setdefault("_build_id", {})[workload] = build
    tj["_note"] = ("HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, KB units; FETCH doubled per "
                   "MI355X_MICROARCH.md: gfx950 counts 128-B reads at 64 B; the doubling is calibrated for wide streaming reads and "
                   "over-counts narrow gathers); kernels launched several times per sweep (k_scan, k_filtrar) are summed "
                   "over their per-launch averages.  Written by tools/pmc_summary.py --traffic from the file named in _source; _build_id = "
                   "icm_build_id() of the library that was profiled: bench.py reports the numbers only for that very build.")
    json.dump(tj, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps(per, indent=1))
    sys.exit(0)

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
