import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from auto_oo_amd import ops
from auto_oo_amd.parallel import gather_results
pqc, batch, single, thetas = bench.build_geometries(list(range(256)))
G = 256
n_out = 1 + batch.n_theta + batch.n_kappa
results = torch.zeros((G, n_out), dtype=torch.float64, device="cuda")
geoms = list(range(256))
def run(n):
    last = None
    for _ in range(n):
        last = batch.energy_and_gradient(thetas)
    results.copy_(last)
for _ in range(30):
    run(8); torch.cuda.synchronize()
gather_results(results, geoms, 256, None)
def timed(K, events, gather):
    run(5); torch.cuda.synchronize()
    if events: ops.profile_begin()
    t0 = time.perf_counter()
    run(K)
    if gather: gather_results(results, geoms, 256, None)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if events: ops.profile_end()
    return el / K * 1e6
for K in (20, 200):
    for ev in (True, False):
        for ga in (True, False):
            v = sorted(timed(K, ev, ga) for _ in range(5))
            print(f"K={K} events={ev} gather={ga}: median {v[2]:.1f} us/step, min {v[0]:.1f}", flush=True)
