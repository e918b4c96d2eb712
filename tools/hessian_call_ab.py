"""The batched Hessian call (E + gradient + full Hessian of a stack, one library call) under the library's
kernel-selection switches, by stack size.   python tools/hessian_call_ab.py [G ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from auto_oo_amd import _lib
sizes = [int(a) for a in sys.argv[1:]] or [8, 64]
def T(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6, r
for G in sizes:
    pqc, batch, single, thetas = bench.build_geometries(list(range(G)))
    ref = None
    for opts in ({}, {"k1_force_nt": 1}, {"k1_force_nt": 2}, {"no_ride": 1}, {"k1_no_pair": 1}, {"gm_one_per_cu": 1},
                 {"gm_two_per_cu": 1}, {"panel_rows": 4}, {"panel_rows": 16}, {"hess_vk_pass": 1}, {}):
        with _lib.debug_options(**opts):
            t, (E, g, H) = T(lambda: batch.energy_gradient_hessian(thetas))
        if ref is None: ref = H.clone()
        print(f"G={G:3d} {str(opts):22s} {t:8.1f} us per call, max |dH| {(H - ref).abs().max().item():.1e}", flush=True)
    del batch
