import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pqc, batch, single, thetas = bench.build_geometries(list(range(64)))
res = torch.zeros((64, 1 + batch.n_theta + batch.n_kappa), dtype=torch.float64, device="cuda")
for rep in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        res[:] = batch.energy_and_gradient(thetas)
    torch.cuda.synchronize()
    print(rep, f"{(time.perf_counter()-t0)/20*1e6:.1f} us per batched call", flush=True)
