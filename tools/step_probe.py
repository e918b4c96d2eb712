"""The lockstep Berry-loop step (tracking regime: positive definite Hessians) driven call by call and as ONE
library call (oovqe_oo_newton_step_batch), at several batch sizes; time until the step's outputs are complete.
    python tools/step_probe.py [G ...] [option=value ...]"""
import contextlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import auto_oo_amd as aoo                               # noqa: E402
from auto_oo_amd.synthetic import synthetic_loop        # noqa: E402
import bench                                            # noqa: E402

opts = dict((a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[1:] if "=" in a)   # library options name=value
sizes = [int(a) for a in sys.argv[1:] if "=" not in a] or [8, 16, 32, 64]
if opts:
    from auto_oo_amd import _lib
    _ctx = _lib.debug_options(**opts)
    _ctx.__enter__()
    print("library options:", opts)
pqc = aoo.Parameterized_circuit(bench.NCAS, bench.NELECAS, None, ansatz="ucc")
base, loop = synthetic_loop(bench.NAO, 20263, max(sizes), eps=0.01)
bmol = aoo.Moldata(base["int1e_ao"], base["int2e_ao"], base["overlap"], base["nuc"], bench.NELEC)
boo = aoo.OO_pqc(pqc, bmol, bench.NCAS, bench.NELECAS, oao_mo_coeff=base["oao_mo_coeff"], freeze_active=True)
with contextlib.redirect_stdout(sys.stderr):
    e_l, th_l, _, _, _ = boo.full_optimization(torch.full((pqc.theta_shape,), 0.1, dtype=torch.float64, device="cuda"),
                                               max_iterations=80, conv_tol=1e-11, verbose=None)
theta0, c_star = th_l[-1], boo.oao_mo_coeff
mols = [aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], bench.NELEC) for P in loop]
bopt = aoo.BatchedNewtonStep(verbose=0)
res = {}
for G in sizes:
    batch = aoo.OO_pqc_batch(pqc, mols[:G], bench.NCAS, bench.NELECAS, oao_mo_coeffs=[c_star] * G, freeze_active=True)
    thetas0 = theta0.reshape(1, -1).repeat(G, 1).contiguous()
    c_saved = batch.oao_mo_coeff.clone()
    outs = {}
    for by_calls in (True, False):
        batch.step_by_calls = by_calls
        ts = []
        for rep in range(12):
            batch.oao_mo_coeff.copy_(c_saved)
            batch.refresh_mo_coeff()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th, en, pend = batch.damped_newton_step(thetas0, bopt, defer_lowest=True)
            done = torch.cuda.Event()
            done.record()
            done.synchronize()
            ts.append(time.perf_counter() - t0)
            pend.result()
            torch.cuda.synchronize()
        outs[by_calls] = (sorted(ts[2:])[len(ts[2:]) // 2] * 1e3, th.clone(), en.clone(), batch.oao_mo_coeff.clone())
    same = all(torch.equal(a, b) for a, b in zip(outs[True][1:], outs[False][1:]))
    res[G] = (outs[True][0], outs[False][0])
    print(f"G={G:3d}: step by calls {outs[True][0]:.3f} ms, one call {outs[False][0]:.3f} ms, same bits: {same}, "
          f"speculating: {batch._all_pd_last_step}", flush=True)
g0 = max(sizes)
for G in sizes:
    print(f"t({g0}) / t({G}): by calls {res[g0][0] / res[G][0]:.2f} x, one call {res[g0][1] / res[G][1]:.2f} x")
