"""Summarise the rocprofv3 passes of tools/pmc_k1.sh (N = 200 four-index transform, K1
contract_kernel) into profiles/k1_n200_<tag>.json: per-dispatch means of the MFMA / LDS / wait
counters, HBM bytes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) and the kernel-trace durations."""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
N = 200
out = {"workload": f"N = {N} four-index transform, 4 quarter steps = 4 dispatches of contract_kernel<13,.,5> per transform",
       "counters_per_dispatch": {}, "source": "tools/pmc_k1.sh (separate rocprofv3 --pmc passes, python program directly after --)"}
per_kind = {}
for d in "abcde":
    path = os.path.join(ROOT, "gpurun_out", f"k1_pmc_{d}", "p_counter_collection.csv")
    if not os.path.exists(path):
        continue
    acc = {}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if "contract_kernel<13" not in r["Kernel_Name"] and "contract_pair_kernel<13" not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            kind = "pair" if "contract_pair_kernel" in r["Kernel_Name"] else "single"
            per_kind.setdefault(kind, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out["counters_per_dispatch"][k] = {"mean": sum(v) / len(v), "dispatches": len(v)}
c = out["counters_per_dispatch"]
stats = os.path.join(ROOT, "gpurun_out", "k1_trace", "p_kernel_stats.csv")
if os.path.exists(stats):
    with open(stats) as fh:
        for r in csv.DictReader(fh):
            if "contract_kernel<13" in r["Name"] or "contract_pair_kernel<13" in r["Name"]:
                out.setdefault("kernel_trace", []).append({"kernel": r["Name"][:80], "calls": int(r["Calls"]),
                                                           "avg_us": float(r["AverageNs"]) / 1e3})
flops = 2.0 * N ** 5
if "kernel_trace" in out:
    avg = sum(k["avg_us"] * k["calls"] for k in out["kernel_trace"]) / sum(k["calls"] for k in out["kernel_trace"])
    out["derived"] = {"avg_quarter_step_us": avg, "tflops": flops / (avg * 1e-6) / 1e12,
                      "frac_of_78.6": flops / (avg * 1e-6) / 1e12 / 78.6}
if "SQ_INSTS_MFMA" in c:
    out.setdefault("derived", {})["mfma_instructions_per_quarter_step"] = c["SQ_INSTS_MFMA"]["mean"]
    out["derived"]["algorithmic_mfma_per_quarter_step"] = flops / 2048.0
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
    # SQ_* counters are summed over the SEs/XCDs; GRBM_GUI_ACTIVE likewise (8 XCDs): ratio per SIMD
    out["derived"]["mfma_busy_over_gui_active"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / c["GRBM_GUI_ACTIVE"]["mean"]
    out["derived"]["note_mfma_busy"] = ("SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE as collected (both summed over "
                                        "the 8 XCDs); divide by the 4 SIMDs x 32 CUs per XCD a cycle can count for "
                                        "when comparing with 1.0")
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    hbm = (2.0 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024.0
    out["derived"]["hbm_bytes_per_quarter_step"] = hbm
    out["derived"]["algorithmic_bytes_per_quarter_step"] = 2.0 * 8.0 * N ** 4
    out["derived"]["hbm_over_algorithmic"] = hbm / (2.0 * 8.0 * N ** 4)
out["counters_per_dispatch_by_kernel"] = {kind: {k: sum(v) / len(v) for k, v in d.items()} for kind, d in per_kind.items()}
for kind, d in out["counters_per_dispatch_by_kernel"].items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        d["mfma_busy_fraction"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["GRBM_GUI_ACTIVE"] / 128.0
out["workload"] = (f"N = {N} four-index transform: quarter steps 1, 2 on contract_pair_kernel<13,5> (two strips per wave), "
                   "step 3 (B = 200, 32-wide strips would pad) and step 4 (LAST) on contract_kernel<13,.,5>")
with open(os.path.join(ROOT, "profiles", f"k1_n200_{tag}.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out, indent=1))
