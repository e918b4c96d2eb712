"""Per-kernel means of every counter in a rocprofv3 --pmc output directory (tools only)."""
import csv, glob, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n =", len(next(iter(cs.values()))))
