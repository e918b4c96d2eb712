#!/usr/bin/env python3
"""Host-side cost of one batched / single evaluation call (scratch tool)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

pqc, batch, single, thetas = bench.build_geometries(list(range(64)))
for _ in range(5):
    batch.energy_and_gradient(thetas)
torch.cuda.synchronize()
for n in (1, 8, 64):
    t0 = time.perf_counter()
    for _ in range(20):
        out = batch.energy_and_gradient(thetas, count=n)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"batch={n}: host submit {1e6*(t1-t0)/20:.1f} us/call, total {1e6*(t2-t0)/20:.1f} us/call", flush=True)
th0 = thetas[0].contiguous()
for _ in range(20):
    single.energy_and_gradient(th0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    single.energy_and_gradient(th0)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"single: host submit {1e6*(t1-t0)/200:.1f} us/call, total {1e6*(t2-t0)/200:.1f} us/call")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    batch.energy_and_gradient(thetas, count=1)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
