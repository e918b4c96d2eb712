"""Timing of the Cholesky fast path of the Newton direction (newton_chol.hip) beside the band route:
    python tools/newton_chol_probe.py [n]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import auto_oo_amd as aoo               # noqa: E402
from auto_oo_amd import ops, _lib       # noqa: E402
from auto_oo_amd._lib import dptr, stream_ptr, check  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 331
    lib = _lib.load()
    rng = np.random.default_rng(0)
    for G in (1, 8, 16, 32, 64):
        A = rng.standard_normal((G, n, n))
        H = torch.tensor(A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n)).cuda()
        g = torch.tensor(rng.standard_normal((G, n))).cuda()
        dp = torch.empty((G, n), dtype=torch.float64, device="cuda")
        nu = torch.empty(G, dtype=torch.float64, device="cuda")
        info = torch.empty(G, dtype=torch.float64, device="cuda")
        low = torch.empty(G, dtype=torch.float64, device="cuda")
        wpd = torch.empty(lib.oovqe_newton_direction_pd_work_size(n, G), dtype=torch.float64, device="cuda")
        wr = torch.empty(lib.oovqe_newton_direction_rest_work_size(n, G), dtype=torch.float64, device="cuda")

        def pd():
            check(lib.oovqe_newton_direction_pd(dptr(H), dptr(g), n, G, 1e-6, dptr(wpd), dptr(dp), dptr(nu),
                                                dptr(info), stream_ptr()), "pd")

        def rest(which):
            check(lib.oovqe_newton_direction_rest(dptr(H), dptr(g), n, G, 1e-6, 1e-6, 1.1, 1, dptr(info), which, 0,
                                                  dptr(wr), dptr(dp), dptr(low), dptr(nu), stream_ptr()), "rest")
        t_pd = timed(pd)
        t_noop = timed(lambda: rest(1))
        t_low = timed(lambda: rest(2))
        with _lib.debug_options(newton_no_chol=1):
            t_band = timed(lambda: ops.newton_direction(H, g))
        t_all = timed(lambda: ops.newton_direction(H, g))
        t_defer = timed(lambda: ops.newton_direction(H, g, defer_lowest=True))
        ref = -torch.linalg.solve(H, g)
        pd()
        err = float((dp - ref).abs().max())
        print(f"n={n} G={G}: pd {t_pd:.1f} us, rest(which=1, nothing to do) {t_noop:.1f} us, lowest only {t_low:.1f} us, "
              f"band route alone {t_band:.1f} us, newton_direction {t_all:.1f} us (deferred, pipelined: {t_defer:.1f}); "
              f"info {info.tolist()[:3]} max|dp - solve| {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
