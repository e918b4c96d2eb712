import contextlib, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import auto_oo_amd as aoo, bench
from auto_oo_amd.synthetic import synthetic_problem
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pqc = aoo.Parameterized_circuit(bench.NCAS, bench.NELECAS, None, ansatz="ucc")
Ps = [synthetic_problem(bench.NAO, 700 + g) for g in range(G)]
mols = [aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], bench.NELEC) for P in Ps]
batch = aoo.OO_pqc_batch(pqc, mols, bench.NCAS, bench.NELECAS, oao_mo_coeffs=[P["oao_mo_coeff"] for P in Ps], freeze_active=True)
th = torch.full((G, pqc.theta_shape), 0.1, dtype=torch.float64, device="cuda")
for _ in range(12):
    batch.energy_gradient_hessian(th)
torch.cuda.synchronize()
