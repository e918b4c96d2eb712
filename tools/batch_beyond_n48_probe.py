import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import auto_oo_amd as aoo
from auto_oo_amd.synthetic import synthetic_problem
N, G = 52, 2
pqc = aoo.Parameterized_circuit(3, 4, None, ansatz="ucc")
mols, coeffs, objs = [], [], []
for g in range(G):
    P = synthetic_problem(N, 4100 + g)
    mols.append(aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 16))
    coeffs.append(P["oao_mo_coeff"])
    objs.append(aoo.OO_pqc(pqc, mols[-1], 3, 4, oao_mo_coeff=P["oao_mo_coeff"]))
batch = aoo.OO_pqc_batch(pqc, mols, 3, 4, oao_mo_coeffs=coeffs)
print("flags", batch.eri_flags, "packed", None if batch._eri_packed is None else tuple(batch._eri_packed.shape))
th = torch.tensor(np.random.default_rng(2).uniform(0, 2 * np.pi, (G, 4)), device="cuda")
try:
    E, g, H = batch.energy_gradient_hessian(th)
    for i, oo in enumerate(objs):
        h1 = oo.full_hessian(th[i]); e1, g1 = oo.energy_and_gradient(th[i])
        print(i, abs(E[i].item() - e1.item()), (g[i] - g1).abs().max().item(), (H[i] - h1).abs().max().item())
except Exception as exc:
    print("energy_gradient_hessian at N = 52:", type(exc).__name__, str(exc)[:300])
new_t, e_new, low = batch.damped_newton_step(th) if True else None
print("damped step energies", e_new.tolist())
