"""Host-side profile of the lockstep Newton step (where the Python time of a step goes)."""
import cProfile
import contextlib
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import auto_oo_amd as aoo                               # noqa: E402
from auto_oo_amd.synthetic import synthetic_loop        # noqa: E402
import bench                                            # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pqc = aoo.Parameterized_circuit(bench.NCAS, bench.NELECAS, None, ansatz="ucc")
base, loop = synthetic_loop(bench.NAO, 20263, G, eps=0.01)
bmol = aoo.Moldata(base["int1e_ao"], base["int2e_ao"], base["overlap"], base["nuc"], bench.NELEC)
boo = aoo.OO_pqc(pqc, bmol, bench.NCAS, bench.NELECAS, oao_mo_coeff=base["oao_mo_coeff"], freeze_active=True)
with contextlib.redirect_stdout(sys.stderr):
    e_l, th_l, _, _, _ = boo.full_optimization(torch.full((pqc.theta_shape,), 0.1, dtype=torch.float64, device="cuda"),
                                               max_iterations=80, conv_tol=1e-11, verbose=None)
theta0, c_star = th_l[-1], boo.oao_mo_coeff
mols = [aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], bench.NELEC) for P in loop]
batch = aoo.OO_pqc_batch(pqc, mols, bench.NCAS, bench.NELECAS, oao_mo_coeffs=[c_star] * G, freeze_active=True)
thetas0 = theta0.reshape(1, -1).repeat(G, 1).contiguous()
bopt = aoo.BatchedNewtonStep(verbose=0)
pend = []
for _ in range(5):
    pend.append(batch.damped_newton_step(thetas0, bopt, defer_lowest=True)[2])
torch.cuda.synchronize()
# host time of a call when the device is NOT the bottleneck: time to return, device drained between calls
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th, en, p = batch.damped_newton_step(thetas0, bopt, defer_lowest=True)
    ts.append(time.perf_counter() - t0)
    pend.append(p)
print(f"G={G}: damped_newton_step returns after {sorted(ts)[10] * 1e6:.0f} us (median of 20)")
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    th, en, p = batch.damped_newton_step(thetas0, bopt, defer_lowest=True)
    pend.append(p)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
