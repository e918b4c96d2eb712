"""End-to-end OO_pqc evaluation at N = 100 (T2 path, streaming half-transform) against the oracle energy
and finite differences of its own energy."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import auto_oo_amd as aoo
from auto_oo_amd.synthetic import synthetic_problem
from oracle import cpu_ref as R
N, ncas, nelecas, nelec = int(sys.argv[1]) if len(sys.argv) > 1 else 100, 3, 4, 20
pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="ucc")
P = synthetic_problem(N, 777)
mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
oo = aoo.OO_pqc(pqc, mol, ncas, nelecas, oao_mo_coeff=P["oao_mo_coeff"])
theta = torch.tensor(np.random.default_rng(1).uniform(0, 2 * np.pi, pqc.theta_shape))
E, grad = oo.energy_and_gradient(theta)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): oo.energy_and_gradient(theta)
torch.cuda.synchronize(); print("eval us", (time.perf_counter() - t0) / 20 * 1e6)
omol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
ooo = R.OracleOOPQC(R.OraclePQC(ncas, nelecas, "ucc"), omol, ncas, nelecas, P["oao_mo_coeff"])
t0 = time.perf_counter(); E_ref = ooo.energy_from_parameters(theta); print("oracle energy s", time.perf_counter() - t0)
print("E diff", abs(E.item() - E_ref.item()), "E", E_ref.item())
h = 1e-5
for k in range(pqc.theta_shape):
    tp, tm = theta.clone(), theta.clone(); tp[k] += h; tm[k] -= h
    fd = (oo.energy_from_parameters(tp).item() - oo.energy_from_parameters(tm).item()) / (2 * h)
    print("dE/dtheta", k, grad[k].item(), fd)
kap = torch.zeros(oo.n_kappa, dtype=torch.float64); j = int(torch.argmax(grad[pqc.theta_shape:].abs()))
kp, km = kap.clone(), kap.clone(); kp[j] += h; km[j] -= h
fd = (oo.energy_from_parameters(theta, kp).item() - oo.energy_from_parameters(theta, km).item()) / (2 * h)
print("dE/dkappa", j, grad[pqc.theta_shape + j].item(), fd)
