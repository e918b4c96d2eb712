"""Per-launch breakdown of ONE un-batched evaluation (OO_pqc.energy_and_gradient), HIP events."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auto_oo_amd import ops
import bench
pqc, batch, single, thetas = bench.build_geometries(list(range(4)))
th0 = thetas[0].contiguous()
for _ in range(50):
    single.energy_and_gradient(th0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    single.energy_and_gradient(th0)
torch.cuda.synchronize()
print("single eval us:", (time.perf_counter() - t0) / 500 * 1e6)
ops.profile_begin(detail=True)
for _ in range(200):
    single.energy_and_gradient(th0)
torch.cuda.synchronize()
ms, cnt, by = ops.profile_end()
print({k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in by.items()})
