"""Sector adjoint at CAS(8e,8o), k = 1: the string-driven lambda (round 4) against W = Ms^T V in memory (round 3,
option sector_lambda_w) over the batch size."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import auto_oo_amd as aoo
from auto_oo_amd._lib import debug_options
pqc = aoo.Parameterized_circuit(8, 8, None, ansatz="kupccd", k=1)
eng = pqc._sector
rng = np.random.default_rng(6)
c1 = torch.tensor(rng.standard_normal((8, 8)), device="cuda")
c2 = torch.tensor(rng.standard_normal((8,) * 4), device="cuda")
def T(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for B in (1, 4, 16, 32, 64, 128, 256):
    th = torch.tensor(rng.uniform(0, 2 * np.pi, (B, int(pqc.theta_shape))), device="cuda")
    psi = eng.state(th)
    out = []
    for w in (0, 1):
        with debug_options(sector_lambda_w=w):
            out.append(T(lambda: eng.adjoint(th, psi, c1, c2)))
    print(f"batch {B:4d}: string-driven {out[0]:8.1f} us   W in memory {out[1]:8.1f} us", flush=True)
