"""Summarise rocprofv3 csv output (kernel stats + per-kernel PMC means) into one text file."""
import csv, glob, os, sys, collections
root = sys.argv[1]
print("# rocprofv3 summary for", root)
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("## kernel stats:", os.path.relpath(f, root))
    for i, row in enumerate(csv.reader(open(f))):
        if i < 8: print(",".join(row))
for f in sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("## kernel durations (us): name, calls, mean, min, max; plus VGPR/SGPR/LDS/scratch of the first call")
    for k, v in by.items():
        r0 = next(r for r in rows if r["Kernel_Name"] == k)
        extra = {c: r0.get(c) for c in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
        print(f"{k[:90]}, {len(v)}, {sum(v)/len(v):.1f}, {min(v):.1f}, {max(v):.1f}, {extra}")
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        acc = collections.defaultdict(list)
        for r in rows:
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        print("## pmc:", os.path.basename(d))
        for (k, c), v in sorted(acc.items()):
            if "lqmpc" in k:
                print(f"{k[:60]}, {c}, n={len(v)}, mean={sum(v)/len(v):.6g}")
