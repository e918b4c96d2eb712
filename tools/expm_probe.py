"""expm and the matrix products behind it at N = 100, 200, 300 (tools only)."""
import time, torch
from auto_oo_amd import ops
for N in (64, 100, 200, 300):
    gen = torch.Generator(device="cuda").manual_seed(N)
    K = torch.randn((N, N), generator=gen, dtype=torch.float64, device="cuda") * (0.3 / N ** 0.5)
    K = K - K.T
    A = torch.randn((N, N), generator=gen, dtype=torch.float64, device="cuda")
    def T(f, n=50):
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6, r
    t_e, U = T(lambda: ops.expm(K))
    t_m, P = T(lambda: ops.matmul_nn(A, A))
    ref = torch.linalg.matrix_exp(K)
    print(f"N {N}: expm {t_e:7.1f} us (err {float((U - ref).abs().max()):.1e}), matmul_nn {t_m:6.1f} us (err {float((P - A @ A).abs().max() / (A @ A).abs().max()):.1e})")
