"""full_hessian at N = 43: eager wall time, host submit time, and the same launch sequence replayed
as a captured HIP graph (tools only)."""
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
for _ in range(3):
    H = oo.full_hessian(theta0)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    H = oo.full_hessian(theta0)
t_submit = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t_wall = (time.perf_counter() - t0) / n
print(f"eager: submit {t_submit * 1e3:.3f} ms, wall {t_wall * 1e3:.3f} ms per call")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        oo.full_hessian(theta0)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        Hg = oo.full_hessian(theta0)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print("graph result matches eager:", float((Hg - H).abs().max()))
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    print(f"graph replay: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call")
except Exception as e:
    print("graph capture failed:", repr(e)[:400])
