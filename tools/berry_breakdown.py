"""Where does one Berry-loop step (configs[3]) spend its time?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import auto_oo_amd as aoo
from auto_oo_amd.synthetic import synthetic_problem
import bench
pqc = aoo.Parameterized_circuit(bench.NCAS, bench.NELECAS, None, ansatz="ucc")
P = synthetic_problem(bench.NAO, 20262)
mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], bench.NELEC)
oo = aoo.OO_pqc(pqc, mol, bench.NCAS, bench.NELECAS, oao_mo_coeff=P["oao_mo_coeff"])
theta0 = torch.full((pqc.theta_shape,), 0.1, dtype=torch.float64, device="cuda")
opt = aoo.NewtonStep(verbose=0)
def T(f, n=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r
kappa = torch.zeros(oo.n_kappa, dtype=torch.float64, device="cuda")
t, grad = T(lambda: oo.full_gradient(theta0)); print(f"full_gradient            {t:8.3f} ms")
t, h1 = T(lambda: oo.circuit_circuit_hessian(theta0)); print(f"circuit_circuit_hessian  {t:8.3f} ms")
t, h2 = T(lambda: oo.orbital_circuit_hessian(theta0)); print(f"orbital_circuit_hessian  {t:8.3f} ms")
t, h3 = T(lambda: oo.orbital_orbital_hessian(theta0)); print(f"orbital_orbital_hessian  {t:8.3f} ms")
t, hess = T(lambda: oo.full_hessian(theta0)); print(f"full_hessian             {t:8.3f} ms")
t, _ = T(lambda: torch.linalg.eigh(hess)); print(f"torch.linalg.eigh        {t:8.3f} ms  (n = {hess.shape[0]})")
t, _ = T(lambda: opt.damped_newton_step(oo.energy_from_parameters, (theta0, kappa), grad, hess)); print(f"damped_newton_step       {t:8.3f} ms")
