"""half_tri workgroup split for batch sizes that do not divide the CU count: time per batched call
with the cost-model split and with the plain W = n_cu / batch (debug option tri_plain_w)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from auto_oo_amd import _lib, ops
for G in (50, 96, 100, 192, 300):
    pqc, batch, single, thetas = bench.build_geometries([g % 8 for g in range(G)])
    row = [G]
    for plain in (True, False):
        _lib.load().oovqe_debug_set_option(b"tri_plain_w", int(plain))
        for _ in range(20): batch.energy_and_gradient(thetas)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): batch.energy_and_gradient(thetas)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        row.append(("plain" if plain else "model", round(dt * 1e6, 1), round(G / dt)))
    print(row, flush=True)
    del batch, single; torch.cuda.empty_cache()
