"""List the kernel launches of ONE call from a rocprofv3 kernel trace (csv): the launches between two
consecutive occurrences of a marker kernel.  usage: python tools/trace_one_call.py <trace_kernel_trace.csv> <last kernel substring> [occurrence]"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2]
occ = int(sys.argv[3]) if len(sys.argv) > 3 else -3
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
i1 = idx[occ]
i0 = idx[occ - 1] + 1
t0 = int(rows[i0]["Start_Timestamp"])
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:60]
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))
    wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))) * int(r.get("Workgroup_Size_Y", 1)) * int(r.get("Workgroup_Size_Z", 1))
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f}  wg {gx // max(wx, 1):6d} x {wx:>4}  {short(r['Kernel_Name'])}")
print(f"span {(int(rows[i1]['End_Timestamp']) - t0) / 1e3:.1f} us, {i1 - i0 + 1} launches")
