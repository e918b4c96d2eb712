import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for G in [int(a) for a in sys.argv[1:]]:
    pqc, batch, single, thetas = bench.build_geometries(list(range(G)))
    for _ in range(30):
        batch.energy_and_gradient(thetas)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        batch.energy_and_gradient(thetas)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(G, "submit us/call", round((t1 - t0) / 300 * 1e6, 1), "wall us/call", round((t2 - t0) / 300 * 1e6, 1), flush=True)
