"""oovqe_newton_direction beyond n = 672 (launch sequence per panel) against torch.linalg.eigh (rocSOLVER)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from auto_oo_amd import ops


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


rng = np.random.default_rng(1)
for n in [672, 1000, 2000, 4700]:
    A = rng.standard_normal((n, n))
    H = torch.tensor(A + A.T).cuda()
    g = torch.tensor(rng.standard_normal(n)).cuda()
    t2 = timed(lambda: ops.newton_direction(H, g))
    te = timed(lambda: torch.linalg.eigh(H), reps=1)
    dp, low, nu = ops.newton_direction(H, g)
    ref = torch.linalg.eigvalsh(H)[0]
    res = ((H + nu * torch.eye(n, dtype=torch.float64, device="cuda")) @ dp + g).abs().max() / (1 + g.abs().max())
    print(f"n={n}: oovqe_newton_direction {t2:.2f} ms, torch.linalg.eigh {te:.1f} ms, "
          f"|lambda_min - eigvalsh| = {abs(float(low - ref)):.2e}, residual {float(res):.1e}", flush=True)
