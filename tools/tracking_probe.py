"""How far from the base optimum do the loop geometries of synthetic_loop stay positive definite?
    python tools/tracking_probe.py"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import auto_oo_amd as aoo                      # noqa: E402
from auto_oo_amd import ops                    # noqa: E402
from auto_oo_amd.synthetic import synthetic_loop  # noqa: E402

N, NCAS, NELECAS, NELEC = 43, 3, 4, 16
pqc = aoo.Parameterized_circuit(NCAS, NELECAS, None, ansatz="ucc")
for eps in (0.05, 0.02, 0.01, 0.003):
    base, loop = synthetic_loop(N, 20263, 8, eps=eps)
    mol = aoo.Moldata(base["int1e_ao"], base["int2e_ao"], base["overlap"], base["nuc"], NELEC)
    oo = aoo.OO_pqc(pqc, mol, NCAS, NELECAS, oao_mo_coeff=base["oao_mo_coeff"], freeze_active=True)
    theta0 = torch.full((pqc.theta_shape,), 0.1, dtype=torch.float64, device="cuda")
    t0 = time.perf_counter()
    energy_l, theta_l, kappa_l, c_l, eig_l = oo.full_optimization(theta0, max_iterations=60, conv_tol=1e-10, verbose=0)
    torch.cuda.synchronize()
    print(f"eps {eps}: base optimisation {len(energy_l)} iterations in {time.perf_counter() - t0:.2f} s, "
          f"E = {energy_l[-1]:.10f}, last lowest eigenvalues {[float(e) for e in eig_l[-3:]]}", flush=True)
    theta_star, c_star = theta_l[-1], oo.oao_mo_coeff
    lows, gn = [], []
    for P in loop:
        m = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], NELEC)
        o = aoo.OO_pqc(pqc, m, NCAS, NELECAS, oao_mo_coeff=c_star, freeze_active=True)
        g = o.full_gradient(theta_star)
        H = o.full_hessian(theta_star)
        dp, low, nu, info = ops.newton_direction(H, g, want_info=True)
        lows.append((float(low), float(info)))
        gn.append(float(g.abs().max()))
    print("   loop points: (lowest eigenvalue, info)", [(round(a, 5), b) for a, b in lows], "max |grad|", max(gn), flush=True)
