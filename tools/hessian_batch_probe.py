"""Time of OO_pqc_batch.energy_gradient_hessian for the geometry counts on the command line
(under rocprofv3 --kernel-trace --stats: the kernels of the batched Hessian call), and the
round-3 K-type path (quarter transform from stage 1) against the round-2 one (its own pass)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auto_oo_amd import _lib
import bench
for G in [int(a) for a in sys.argv[1:]] or [64]:
    pqc, batch, single, thetas = bench.build_geometries(list(range(G)))
    def run():
        for _ in range(3):
            out = batch.energy_gradient_hessian(thetas)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = batch.energy_gradient_hessian(thetas)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 20 * 1e3, out
    t, (E, g, H) = run()
    with _lib.debug_options(hess_vk_pass=1, hess_own_stage1=1):
        t_old, (E0, g0, H0) = run()
    with _lib.debug_options(hess_own_stage1=1):
        t_own, (E1, g1, H1) = run()
    print(f"G={G}: energy_gradient_hessian {t:.3f} ms (evaluation with its own stage 1: {t_own:.3f} ms; and the K-type "
          f"quarter transform as its own pass: {t_old:.3f} ms), vs own stage 1: max |dH| = {(H - H1).abs().max().item():.2e}, "
          f"|dE| = {(E - E1).abs().max().item():.1e}, |dg| = {(g - g1).abs().max().item():.1e}; vs own pass: max |dH| = "
          f"{(H - H0).abs().max().item():.2e} (|H| max {H0.abs().max().item():.2e})", flush=True)
