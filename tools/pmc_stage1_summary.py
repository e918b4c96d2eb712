"""Means per dispatch of the counters in one rocprofv3 --pmc output directory for the packed stage-1
kernel (half_tri_kernel / half_tri_reg_kernel, full-batch launches); appends to gpurun_out/s1_pmc_summary.json."""
import csv, glob, json, os, sys
d = sys.argv[1]
out_path = os.path.join(os.path.dirname(d.rstrip("/")), "s1_pmc_summary.json")
acc = {}
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as fh:
        rows = [r for r in csv.DictReader(fh) if "half_tri_kernel" in r["Kernel_Name"] or "half_tri_reg_kernel" in r["Kernel_Name"]]
    if not rows:
        continue
    top = max(int(r["Grid_Size"]) for r in rows)
    for r in rows:
        if int(r["Grid_Size"]) == top:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
summary = {}
if os.path.exists(out_path):
    with open(out_path) as fh:
        summary = json.load(fh)
for k, v in acc.items():
    v = v[len(v) // 10:]
    summary[k] = {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
with open(out_path, "w") as fh:
    json.dump(summary, fh, indent=1)
print({k: round(v["mean_per_dispatch"], 1) for k, v in summary.items()})
