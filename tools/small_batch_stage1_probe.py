"""Energy-only evaluation of a small stack (the line search's trial) under the realisations of stage 1.
    python tools/small_batch_stage1_probe.py [G ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from auto_oo_amd import _lib, ops
sizes = [int(a) for a in sys.argv[1:]] or [1, 8, 16, 32]
pqc, batch, single, thetas = bench.build_geometries(list(range(max(sizes))))
def T(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6, r
for G in sizes:
    ref = None
    for opts in ({}, {"sym_simple": 1}, {"tri_mode": 1}, {"tri_mode": 2}, {"tri_mode": 4}, {"tri_plain_w": 1}, {"sym_mirror": 1}, {"cas_unfused": 1}):
        with _lib.debug_options(**opts):
            t, out = T(lambda: batch.evaluate(thetas[:G], derivatives=False, count=G))
            ops.profile_begin(detail=True)
            for _ in range(8): batch.evaluate(thetas[:G], derivatives=False, count=G)
            torch.cuda.synchronize(); by = ops.profile_end()[2]
            k = _lib.load().oovqe_last_stage1_kernel().decode()
        e = out[:, 1].clone()
        if ref is None: ref = e
        print(f"G={G:3d} {str(opts):22s} {t:7.1f} us  stage1 {by['half_transform'][0] / max(by['half_transform'][1], 1) * 1e3:6.1f} us  {k:32s} max|dE| {(e - ref).abs().max().item():.1e}", flush=True)
