"""Throughput of the batched evaluation over wall time inside ONE process (first-process-on-a-
fresh-box diagnosis): prints calls/s for every ~1 s window."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
pqc, batch, single, thetas = bench.build_geometries(list(range(64)))
res = torch.zeros((64, 1 + batch.n_theta + batch.n_kappa), dtype=torch.float64, device="cuda")
from auto_oo_amd import ops
if len(sys.argv) > 2 and sys.argv[2] == 'prof':
    ops.profile_begin()
t_start = time.perf_counter()
while time.perf_counter() - t_start < secs:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 1.0:
        for _ in range(50):
            res[:] = batch.energy_and_gradient(thetas)
        n += 50
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"t={time.perf_counter()-t_start:5.1f}s  {dt/n*1e6:7.1f} us per batched call", flush=True)
