"""Batched calls issued alternately on two HIP streams (workspace slots 0/1) against one stream:
do the small tail launches of one call fill the gaps of the other's N^4 pass?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pqc, batch, single, thetas = bench.build_geometries([g % 8 for g in range(G)])
streams = [torch.cuda.Stream() for _ in range(4)]
def run(n, ns):
    for i in range(n):
        if ns > 1:
            with torch.cuda.stream(streams[i % ns]):
                batch.energy_and_gradient(thetas, slot=i % ns)
        else:
            batch.energy_and_gradient(thetas)
for ns in (1, 2, 3, 4, 1, 2):
    run(24, ns); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(240, ns); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 240
    print(f"G={G} streams={ns}: {dt*1e6:.1f} us per call -> {G/dt:.0f} evals/s", flush=True)
