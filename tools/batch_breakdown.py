"""Per-launch breakdown of a batched evaluation for the batch sizes given on the command line."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auto_oo_amd import ops
import bench
for G in [int(a) for a in sys.argv[1:]]:
    pqc, batch, single, thetas = bench.build_geometries(list(range(G)))
    for _ in range(30):
        batch.energy_and_gradient(thetas)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        batch.energy_and_gradient(thetas)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 300 * 1e6
    ops.profile_begin(detail=True)
    for _ in range(100):
        batch.energy_and_gradient(thetas)
    torch.cuda.synchronize()
    ms, cnt, by = ops.profile_end()
    print(G, round(wall, 1), {k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in by.items()}, flush=True)
