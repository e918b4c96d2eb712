"""Timing of oovqe_newton_direction (two-stage multi-workgroup kernel against the one-workgroup one)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import auto_oo_amd as aoo
from auto_oo_amd import ops


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


rng = np.random.default_rng(1)
for n, G in [(331, 1), (331, 8), (331, 64), (331, 256), (100, 64), (600, 1)]:
    A = rng.standard_normal((G, n, n))
    H = torch.tensor(A + A.transpose(0, 2, 1)).cuda()
    g = torch.tensor(rng.standard_normal((G, n))).cuda()
    t2 = timed(lambda: ops.newton_direction(H, g))
    t1 = None
    if n <= 480:
        with aoo._lib.debug_options(newton_one_wg=1):
            t1 = timed(lambda: ops.newton_direction(H, g))
    dp, low, nu = ops.newton_direction(H, g)
    ref = torch.linalg.eigvalsh(H)[:, 0]
    print(f"n={n} G={G}: two-stage {t2:.0f} us, one-workgroup {t1 if t1 is None else round(t1)} us, "
          f"max |lambda_min - eigvalsh| = {float((low - ref).abs().max()):.2e}", flush=True)
