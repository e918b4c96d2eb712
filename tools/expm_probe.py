"""expm and the matrix products behind it at several N and norms (tools only)."""
import time, torch
from auto_oo_amd import ops
for N in (43, 64, 100, 200, 300):
    for scale in (0.02, 0.2, 1.0, 6.0):
        gen = torch.Generator(device="cuda").manual_seed(N)
        K = torch.randn((N, N), generator=gen, dtype=torch.float64, device="cuda")
        K = K - K.T
        K = K * (scale / float(K.abs().sum(0).max()))
        def T(f, n=50):
            for _ in range(5): f()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): r = f()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e6, r
        t_e, U = T(lambda: ops.expm(K))
        ref = torch.linalg.matrix_exp(K)
        print(f"N {N} |K|_1 {scale:5.2f}: expm {t_e:7.1f} us, max err vs matrix_exp {float((U - ref).abs().max()):.1e}, "
              f"|U^T U - 1| {float((U.T @ U - torch.eye(N, device='cuda', dtype=torch.float64)).abs().max()):.1e}")
