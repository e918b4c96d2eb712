"""Batched evaluation time vs batch size beyond 64 geometries (symmetric-integral path): the same
few geometries repeated, since only the shapes matter for timing."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from auto_oo_amd import ops
out = []
for G in (32, 64, 96, 128, 192, 256):
    pqc, batch, single, thetas = bench.build_geometries([g % 8 for g in range(G)])
    for _ in range(20):
        batch.energy_and_gradient(thetas)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        batch.energy_and_gradient(thetas)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    ops.profile_begin(detail=True)
    for _ in range(20):
        batch.energy_and_gradient(thetas)
    torch.cuda.synchronize()
    by = ops.profile_end()[2]
    out.append((G, round(dt * 1e6, 1), round(G / dt), {k: round(v[0] / v[1] * 1e3, 1) for k, v in by.items() if v[1]}))
    print(out[-1], flush=True)
    del batch, single
    torch.cuda.empty_cache()
