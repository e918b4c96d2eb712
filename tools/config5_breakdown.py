"""Where does one kUpCCD CAS(8e,8o) OO evaluation (E + full gradient, adjoint path) spend its time?"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import auto_oo_amd as aoo
from auto_oo_amd.synthetic import synthetic_problem
pqc = aoo.Parameterized_circuit(8, 8, None, ansatz="kupccd", k=1)
P = synthetic_problem(43, 20265)
mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 16)
oo = aoo.OO_pqc(pqc, mol, 8, 8, oao_mo_coeff=P["oao_mo_coeff"])
rng = np.random.default_rng(1)
th = torch.tensor(rng.uniform(0, 2 * np.pi, int(pqc.theta_shape)), device="cuda")
eng = pqc._sector
def T(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, r
th2 = pqc._theta2d(th)
t, psi = T(lambda: eng.state(th2)); print(f"sector state      {t:8.1f} us")
t, (g1, g2) = T(lambda: eng.rdms(psi)); print(f"sector rdms       {t:8.1f} us")
t, res = T(lambda: oo._cas_eval(oo.mo_coeff, g1, g2)); print(f"cas_eval          {t:8.1f} us")
t, _ = T(lambda: eng.adjoint(th2, psi, res["c1"], res["c2"])); print(f"sector adjoint    {t:8.1f} us")
t, _ = T(lambda: oo.energy_and_gradient(th)); print(f"energy_and_gradient {t:6.1f} us")
t, _ = T(lambda: oo.mo_coeff); print(f"mo_coeff property {t:8.1f} us")
