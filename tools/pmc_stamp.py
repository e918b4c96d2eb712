"""profiles/pmc_half_transform.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_stage1_issue.sh
(gpurun_out/s1_issue_counters.json), stamped with the SHA-256 of auto_oo_amd/csrc/cas.hip: bench.py reports
`roofline.traffic` only when the library it runs was built from the same stage-1 source.
usage: python tools/pmc_stamp.py <s1_issue_counters.json> <tag> <kernel label>"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, M, G = 43, 9, 256
src, tag, label = sys.argv[1:4]
c = json.load(open(src))
fk, wk = c["FETCH_SIZE"]["mean_per_dispatch"], c["WRITE_SIZE"]["mean_per_dispatch"]
hbm = (2.0 * fk + wk) * 1024.0
tri = N * (N + 1) // 2
slab = sum(N - (r & ~1) for r in range(N))
alg = G * (8.0 * slab * tri + 8.0 * tri * (M * (M + 1) // 2))
sha = hashlib.sha256(open(os.path.join(ROOT, "auto_oo_amd", "csrc", "cas.hip"), "rb").read()).hexdigest()
old = {}
p = os.path.join(ROOT, "profiles", "pmc_half_transform.json")
if os.path.exists(p):
    o = json.load(open(p))
    old = {"FETCH_SIZE_KB_raw": o.get("FETCH_SIZE_KB_raw"), "WRITE_SIZE_KB_raw": o.get("WRITE_SIZE_KB_raw"),
           "hbm_over_algorithmic": o.get("hbm_over_algorithmic"), "kernel": o.get("kernel")}
out = {"kernel": f"{label}, batched launch over {G} geometries (N={N}, M={M})", "pq_symmetric": True,
       "geometries_per_launch": G, "cas_hip_sha256": sha,
       "source": (f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_stage1_issue.sh) on "
                  f"bench.py --steps 40 --warmup 5; means over the full-batch stage-1 dispatches; profiles/{tag}_stage1_issue_counters.json"),
       "FETCH_SIZE_KB_raw": fk, "WRITE_SIZE_KB_raw": wk,
       "correction": "gfx950 FETCH_SIZE counts 128-B requests at 64 B for wide coalesced streaming reads: x2 (MI355X_MICROARCH.md, HBM section)",
       "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "hbm_over_algorithmic": hbm / alg,
       "launches_averaged": [c["FETCH_SIZE"]["dispatches"], c["WRITE_SIZE"]["dispatches"]], "previous": old}
json.dump(out, open(p, "w"), indent=1)
print(json.dumps(out, indent=1))
