"""Keep the library's own kernels of a rocprofv3 *_kernel_stats.csv (drops the rocSOLVER / rocBLAS / torch
kernels of the synthetic-geometry set-up, which are outside bench.py's timed region) and recompute the shares.
usage: python tools/filter_kernel_stats.py <in.csv> <out.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "anonymous namespace" in r["Name"] or "oovqe" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in keep)
for r in keep:
    r["Percentage"] = f"{100.0 * float(r['TotalDurationNs']) / tot:.4f}"
w = csv.DictWriter(open(sys.argv[2], "w", newline=""), fieldnames=rows[0].keys())
w.writeheader()
w.writerows(keep)
print(f"kept {len(keep)} of {len(rows)} kernels")
