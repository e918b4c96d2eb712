"""Ten lockstep Newton steps over G geometries (tracking regime) for rocprofv3 --kernel-trace; tools/trace_one_call.py
with marker linesearch_update lists the launches of one step with their gaps.
    python tools/lockstep_trace.py [G]"""
import contextlib
import os
import sys

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
c_saved = batch.oao_mo_coeff.clone()
bopt = aoo.BatchedNewtonStep(verbose=0)
for _ in range(10):
    th, en, pend = batch.damped_newton_step(thetas0, bopt, defer_lowest=True)
    pend.result()
    torch.cuda.synchronize()
    batch.oao_mo_coeff.copy_(c_saved)
    batch.refresh_mo_coeff()
    torch.cuda.synchronize()
