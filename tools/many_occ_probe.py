import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import auto_oo_amd as aoo
from auto_oo_amd.synthetic import synthetic_problem
from oracle import cpu_ref as R
N, ncas, nelecas, nelec = 30, 3, 4, 44
pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="ucc")
P = synthetic_problem(N, 4242)
mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
oo = aoo.OO_pqc(pqc, mol, ncas, nelecas, oao_mo_coeff=P["oao_mo_coeff"])
theta = torch.tensor(np.random.default_rng(1).uniform(0, 2 * np.pi, pqc.theta_shape))
omol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
ooo = R.OracleOOPQC(R.OraclePQC(ncas, nelecas, "ucc"), omol, ncas, nelecas, P["oao_mo_coeff"])
for name in ("orbital_orbital_hessian", "orbital_circuit_hessian", "circuit_circuit_hessian", "full_hessian"):
    try:
        H = getattr(oo, name)(theta).cpu()
        Hr = getattr(ooo, name)(theta)
        print(name, tuple(H.shape), "max diff", float((H - Hr).abs().max()), "scale", float(Hr.abs().max()))
    except Exception as e:
        print(name, "FAILED:", repr(e)[:300])
try:
    kappa = torch.tensor(np.random.default_rng(2).standard_normal(oo.n_kappa) * 0.01)
    print("E(theta,kappa) diff", abs(oo.energy_from_parameters(theta, kappa).item() - ooo.energy_from_parameters(theta, kappa).item()))
except Exception as e:
    print("energy with kappa FAILED:", repr(e)[:300])
