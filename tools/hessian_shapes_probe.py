import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import auto_oo_amd as aoo
from auto_oo_amd import _lib
from auto_oo_amd.synthetic import synthetic_problem
for N, nelec in ((20, 28), (24, 4), (43, 30), (17, 10)):
    pqc = aoo.Parameterized_circuit(3, 4, None, ansatz="ucc")
    mols, coeffs = [], []
    for g in range(3):
        P = synthetic_problem(N, 777 + 1000 * g)
        mols.append(aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec))
        coeffs.append(P["oao_mo_coeff"])
    batch = aoo.OO_pqc_batch(pqc, mols, 3, 4, oao_mo_coeffs=coeffs)
    th = torch.tensor(np.random.default_rng(1).uniform(0, 6, (3, pqc.theta_shape))).cuda()
    print("eri_flags", batch.eri_flags, "packed", getattr(batch, "_g_packed", None) is not None, flush=True)
    E, g1, H = batch.energy_gradient_hessian(th)
    with _lib.debug_options(hess_vk_pass=1, hess_own_stage1=1):
        E0, g0, H0 = batch.energy_gradient_hessian(th)
    objs = aoo.OO_pqc(pqc, mols[0], 3, 4, oao_mo_coeff=coeffs[0])
    h1 = objs.full_hessian(th[0])
    print(N, nelec, "M =", (nelec - 4) // 2 + 3, "max|dH| new vs old path", (H - H0).abs().max().item(),
          "vs single-geometry path", (H[0] - h1).abs().max().item(), "dE", (E - E0).abs().max().item(), flush=True)
