"""OO_pqc.full_hessian at kUpCCD CAS(8e,8o), k = 1, six times -- for rocprofv3 --kernel-trace --stats."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import auto_oo_amd as aoo
from auto_oo_amd.synthetic import synthetic_problem
import bench
P = synthetic_problem(bench.NAO, 20264)
mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 24)
pqc = aoo.Parameterized_circuit(8, 8, None, ansatz="kupccd", k=1)
oo = aoo.OO_pqc(pqc, mol, 8, 8, oao_mo_coeff=P["oao_mo_coeff"])
th = torch.tensor(np.random.default_rng(1).uniform(0, 2 * np.pi, int(pqc.theta_shape)), device="cuda")
for _ in range(3):
    H = oo.full_hessian(th)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(6):
    H = oo.full_hessian(th)
torch.cuda.synchronize()
print(f"full_hessian {(time.perf_counter() - t0) / 6 * 1e6:.1f} us, n = {H.shape[0]}")
