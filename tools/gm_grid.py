"""sym_gm_kernel: XCD-aware 1-D grid against the plain (tile, geometry) grid (debug option gm_plain_grid);
per-launch HIP-event times of the labelled kernels and the results compared bit for bit."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from auto_oo_amd import _lib, ops
G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pqc, batch, single, thetas = bench.build_geometries([g % 16 for g in range(G)])
lib = _lib.load()
ref = None
for plain in (0, 1, 0, 1):
    lib.oovqe_debug_set_option(b"gm_plain_grid", plain)
    out = batch.energy_and_gradient(thetas).clone()
    torch.cuda.synchronize()
    if ref is None:
        ref = out
    same = torch.equal(out, ref)
    for _ in range(10):
        batch.energy_and_gradient(thetas)
    torch.cuda.synchronize()
    ops.profile_begin(detail=True)
    t0 = time.perf_counter()
    for _ in range(40):
        batch.energy_and_gradient(thetas)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    _, _, by = ops.profile_end()
    us = {k: round(v[0] / v[1] * 1e3, 1) for k, v in by.items() if v[1]}
    print(f"gm_plain_grid={plain}: identical results {same}; call {dt * 1e6:.1f} us ({G / dt:.0f} evals/s); per launch (us) {us}", flush=True)
