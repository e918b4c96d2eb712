"""profiles/<tag>_source_hashes.json: SHA-256 of every kernel source + which profiles of the round were taken from
the tree with exactly these sources.   usage: python tools/stamp_sources.py <tag> <profile file> ..."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, files = sys.argv[1], sys.argv[2:]
src = os.path.join(ROOT, "auto_oo_amd", "csrc")
out = {"note": "SHA-256 of the kernel sources behind the listed profiles (taken from the tree with exactly these "
               "sources); bench.py stamps roofline.traffic the same way (pmc_half_transform.json carries cas.hip's hash)",
       "sources": {f: hashlib.sha256(open(os.path.join(src, f), "rb").read()).hexdigest()
                   for f in sorted(os.listdir(src)) if f.endswith((".hip", ".h"))},
       "profiles": files}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_source_hashes.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:600])
