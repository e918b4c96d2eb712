"""Does replaying one evaluation as a HIP graph shorten the launch chain?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pqc, batch, single, thetas = bench.build_geometries(list(range(4)))
th0 = thetas[0].contiguous()
def timeit(f, n=2000):
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print("eager  :", round(timeit(lambda: single.energy_and_gradient(th0)), 1), "us")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        single.energy_and_gradient(th0)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = single.energy_and_gradient(th0)
    print("graph  :", round(timeit(g.replay), 1), "us")
    E, grad = single.energy_and_gradient(th0)
    g.replay(); torch.cuda.synchronize()
    print("match  :", float((out[1] - grad).abs().max()), float(abs(out[0] - E)))
except Exception as e:
    print("capture failed:", repr(e)[:300])
