"""Batched evaluation time vs batch size (fused path, or the T2 path with `--unfused`: debug option cas_unfused)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from auto_oo_amd import _lib
unfused = "--unfused" in sys.argv
_lib.load().oovqe_debug_set_option(b"cas_unfused", int(unfused))
out = []
for G in (1, 2, 3, 4, 6, 8, 16, 32, 64):
    pqc, batch, single, thetas = bench.build_geometries(list(range(G)))
    for _ in range(30):
        batch.energy_and_gradient(thetas)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        batch.energy_and_gradient(thetas)
    torch.cuda.synchronize()
    out.append((G, round((time.perf_counter() - t0) / 300 * 1e6, 1)))
print("unfused" if unfused else "fused", out)
