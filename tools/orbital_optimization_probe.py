import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import auto_oo_amd as aoo
from auto_oo_amd.synthetic import synthetic_problem
P = synthetic_problem(43, 20262)
mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 16)
pqc = aoo.Parameterized_circuit(3, 4, None, ansatz="ucc")
oo = aoo.OO_pqc(pqc, mol, 3, 4, oao_mo_coeff=P["oao_mo_coeff"])
th = torch.tensor(np.random.default_rng(1).uniform(0, 2 * np.pi, pqc.theta_shape), device="cuda")
g1, g2 = pqc.get_rdms(th)
c0 = oo.oao_mo_coeff.clone()
for rep in range(3):
    oo.oao_mo_coeff = c0.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e = oo.orbital_optimization(g1, g2, max_iterations=6, verbose=None)
    torch.cuda.synchronize()
    print(f"orbital_optimization {len(e)} iterations: {(time.perf_counter() - t0) * 1e3:.2f} ms, energies {e[0]:.6f} -> {e[-1]:.6f}")
import cProfile, pstats
oo.oao_mo_coeff = c0.clone()
pr = cProfile.Profile(); pr.enable()
oo.orbital_optimization(g1, g2, max_iterations=6, verbose=None)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
