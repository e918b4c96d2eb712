"""CAS path on shapes beyond the fused kernel (N > 48): time per batched cas_eval and the HBM rate of
its N^4 sweep (T2 path: streaming half-transform + K1 + column kernel)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auto_oo_amd import ops
import auto_oo_amd.excitations as X
for N, no, ncas, G in ((64, 6, 4, 12), (96, 8, 4, 3)):
    M = no + ncas
    rng = np.random.default_rng(N)
    g = torch.tensor(rng.standard_normal((G, N, N, N, N)), device="cuda")
    h = torch.tensor(rng.standard_normal((G, N, N)), device="cuda")
    C = torch.tensor(np.stack([np.linalg.qr(rng.standard_normal((N, N)))[0] for _ in range(G)]), device="cuda")
    T2 = None
    def half():
        return [ops.cas_half_transform(g[i], C[i], M) for i in range(G)]
    for _ in range(3): half()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): half()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"N={N} M={M} G={G}: half-transform {dt/G*1e6:.1f} us per geometry -> {8.0*N**4*G/dt/1e12:.2f} TB/s", flush=True)
