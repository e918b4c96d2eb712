"""The circuit + RDM workgroups riding on the launch in front of the panel kernel (sym_gm_kernel / K1) against the
circuit as a launch of its own (option no_ride = 2 / 1: forced either way), by batch size.   python tools/ride_probe.py [G ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from auto_oo_amd import _lib, ops
sizes = [int(a) for a in sys.argv[1:]] or [1, 8, 16, 32, 64, 128, 256]
pqc, batch, single, thetas = bench.build_geometries(list(range(max(sizes))))
def run(G, n=200):
    for _ in range(20): batch.energy_and_gradient(thetas[:G], count=G)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = batch.energy_and_gradient(thetas[:G], count=G)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6, out.clone()
for G in sizes:
    res = {}
    for nr in (2, 1, 2, 1):
        with _lib.debug_options(no_ride=nr):
            t, o = run(G)
        res.setdefault(nr, []).append(t)
        if nr == 2: ref = o
        else: same = torch.equal(o, ref)
    print(f"G={G:4d}: riding {min(res[2]):7.1f} us per call, own launch {min(res[1]):7.1f} us, same bits: {same}", flush=True)
