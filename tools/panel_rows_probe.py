"""cas_panel_kernel: general indices per workgroup (option panel_rows) against the time of a batched evaluation."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auto_oo_amd import _lib, ops
import bench
G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pqc, batch, single, thetas = bench.build_geometries(list(range(G)))
ref = None
for rows in (0, 2, 3, 4, 5, 6, 8, 11, 16):
    with _lib.debug_options(panel_rows=rows):
        for _ in range(20):
            out = batch.energy_and_gradient(thetas)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            out = batch.energy_and_gradient(thetas)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 200 * 1e6
        ops.profile_begin(detail=True)
        for _ in range(50):
            batch.energy_and_gradient(thetas)
        torch.cuda.synchronize()
        ms, cnt, by = ops.profile_end()
    if ref is None:
        ref = out.clone()
    print(f"G={G} panel_rows={rows}: {wall:.1f} us per call, launches (us) "
          f"{ {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in by.items()} }, bitwise {torch.equal(out, ref)}", flush=True)
