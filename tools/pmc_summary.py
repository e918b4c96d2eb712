"""Turn the two rocprofv3 PMC passes of tools/profile_bench.sh (FETCH_SIZE, WRITE_SIZE; separate runs)
into profiles/pmc_half_transform.json: HBM bytes per launch of the dominant kernel, with the gfx950
FETCH_SIZE correction of MI355X_MICROARCH.md (x2 for wide coalesced streaming reads).
usage: python tools/pmc_summary.py <fetch_csv> <write_csv> <kernel substring> <tag> [--general]"""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, M, G = 43, 9, 256


def rows(path, sub, counter):
    """counter values of the full-batch launches (the largest grid) of the kernel"""
    recs = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if sub in r["Kernel_Name"] and r["Counter_Name"] == counter:
                recs.append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    top = max(g for g, _ in recs)
    return [v for g, v in recs if g == top]


def main():
    fetch_csv, write_csv, sub, tag = sys.argv[1:5]
    general = "--general" in sys.argv
    f = rows(fetch_csv, sub, "FETCH_SIZE")
    w = rows(write_csv, sub, "WRITE_SIZE")
    # the first launches include cold-cache effects of the set-up phase: skip a tenth
    f, w = f[len(f) // 10:], w[len(w) // 10:]
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    hbm = (2.0 * fk + wk) * 1024.0
    tri = N * (N + 1) // 2
    if general:
        alg = G * (8.0 * N ** 4 + 8.0 * N * M ** 3)
    else:
        # packed copy: slabs p <= q, upper triangle of each slab (even row starts); columns y <= z written
        slab = sum(N - (r & ~1) for r in range(N))
        alg = G * (8.0 * slab * tri + 8.0 * tri * (M * (M + 1) // 2))
    out = {
        "kernel": f"{sub}, batched launch over {G} geometries (N={N}, M={M})",
        "pq_symmetric": not general,
        "geometries_per_launch": G,
        "source": ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_bench.sh "
                   f"{tag}) on bench.py --steps 200 --warmup 20 (a step = one batched call over 256 geometries); profiles/{os.path.basename(fetch_csv)}, "
                   f"{os.path.basename(write_csv)}"),
        "FETCH_SIZE_KB_raw": fk,
        "WRITE_SIZE_KB_raw": wk,
        "correction": ("gfx950 FETCH_SIZE counts 128-B requests at 64 B for wide coalesced streaming reads: x2 "
                       "(MI355X_MICROARCH.md, HBM section)"),
        "hbm_bytes_per_launch": hbm,
        "algorithmic_bytes_per_launch": alg,
        "hbm_over_algorithmic": hbm / alg,
        "launches_averaged": [len(f), len(w)],
    }
    with open(os.path.join(ROOT, "profiles", "pmc_half_transform.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
