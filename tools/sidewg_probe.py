"""Chained lockstep steps (tracking regime) with the eigenvalue route on 1 ... 32 workgroups per problem:
what paces a loop that collects the lowest Hessian eigenvalues.   python tools/sidewg_probe.py [G ...]"""
import os, sys, time, contextlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import auto_oo_amd as aoo, bench
from auto_oo_amd.synthetic import synthetic_loop
sizes = [int(a) for a in sys.argv[1:]] or [8, 16, 64]
pqc = aoo.Parameterized_circuit(bench.NCAS, bench.NELECAS, None, ansatz="ucc")
base, loop = synthetic_loop(bench.NAO, 20263, max(sizes), eps=0.01)
bmol = aoo.Moldata(base["int1e_ao"], base["int2e_ao"], base["overlap"], base["nuc"], bench.NELEC)
boo = aoo.OO_pqc(pqc, bmol, bench.NCAS, bench.NELECAS, oao_mo_coeff=base["oao_mo_coeff"], freeze_active=True)
with contextlib.redirect_stdout(sys.stderr):
    e_l, th_l, _, _, _ = boo.full_optimization(torch.full((pqc.theta_shape,), 0.1, dtype=torch.float64, device="cuda"), max_iterations=80, conv_tol=1e-11, verbose=None)
theta0, c_star = th_l[-1], boo.oao_mo_coeff
mols = [aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], bench.NELEC) for P in loop]
bopt = aoo.BatchedNewtonStep(verbose=0)
for Gs in sizes:
    batch = aoo.OO_pqc_batch(pqc, mols[:Gs], bench.NCAS, bench.NELECAS, oao_mo_coeffs=[c_star] * Gs, freeze_active=True)
    thetas0 = theta0.reshape(1, -1).repeat(Gs, 1).contiguous()
    c_saved = batch.oao_mo_coeff.clone()
    b = batch._step_block()[0]
    default = b.side_wg
    for wg in (default, 1, 2, 4, 8, 16, 32):
        b.side_wg = wg
        res = []
        for rep in range(4):
            batch.oao_mo_coeff.copy_(c_saved); batch.refresh_mo_coeff(); torch.cuda.synchronize()
            t0 = time.perf_counter(); th = thetas0; pend = []
            for _ in range(6):
                th, en, p = batch.damped_newton_step(thetas0, bopt, defer_lowest=True); pend.append(p)
                batch.oao_mo_coeff.copy_(c_saved); batch.refresh_mo_coeff()
            t1 = time.perf_counter()
            for p in pend: p.result()
            torch.cuda.synchronize(); res.append(((time.perf_counter() - t0) / 6 * 1e3, (t1 - t0) / 6 * 1e3))
        best = min(res)
        print(f"G={Gs} side_wg={wg}{' (default)' if wg == default else ''}: chained step {best[0]:.3f} ms (host loop {best[1]:.3f} ms)", flush=True)
