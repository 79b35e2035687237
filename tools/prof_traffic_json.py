"""Dev tool: fold one tools/prof.sh output directory into profiles/pmc_traffic.json (the HBM bytes per bench step bench.py quotes).

usage: python tools/prof_traffic_json.py gpurun_out/prof_<tag> <key> [steps]
  key    "<last_kernel>:<mode>:C<config>:Bsz<per-GPU batch>:T<T>:<mix>", as bench.py builds it
  steps  launches of the bench in each rocprofv3 pass (prof.sh: --steps 5 --warmup 1 -> 6)
Per step: sum over every lqmpc kernel of (mean FETCH_SIZE + mean WRITE_SIZE) x calls / steps, KB -> bytes; FETCH_SIZE is NOT doubled
(the guide's x2 is calibrated for 16 B/lane streaming reads; these kernels read 8 B/lane), and the same sum of mean durations as
kernel_ms_per_launch, which bench.py compares with its live HIP-event time before it quotes the traffic."""
import csv, glob, json, os, sys, collections
root, key = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE_SQ_INSTS_VALU_MFMA_F64"):
    for f in glob.glob(os.path.join(root, "pmc_" + name + "*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
detail, total, ms, mfma = {}, 0.0, 0.0, 0.0
for k, v in dur.items():
    if "lqmpc" not in k:
        continue
    short = k.split("(")[0].replace("void ", "").replace("lqmpc::", "")
    per_step = len(v) / steps
    ms += sum(v) / len(v) * per_step
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = cnt[c].get(k)
        if vals:
            kb = sum(vals) / len(vals) * per_step
            detail[f"{short}:{c}_KB_per_step"] = round(kb, 1)
            total += kb * 1024
    vals = cnt["SQ_INSTS_VALU_MFMA_F64"].get(k)
    if vals:
        mfma += sum(vals) / len(vals) * per_step
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")
d = json.load(open(out)) if os.path.exists(out) else {}
d[key] = {"hbm_bytes_per_launch": int(total), "kernel_ms_per_launch": round(ms, 4),
          "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB, FETCH not doubled: 8 B/lane loads) of `bench.py --steps 5 --warmup 1`, {os.path.basename(root)}",
          "detail": detail}
if mfma:
    d[key]["mfma_f64_insts_per_launch"] = int(mfma)
json.dump(d, open(out, "w"), indent=1)
print(key, d[key]["hbm_bytes_per_launch"], d[key]["kernel_ms_per_launch"], d[key].get("mfma_f64_insts_per_launch"))
