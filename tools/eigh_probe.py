import time, torch
torch.manual_seed(0)
for n, G in ((331, 1), (331, 8), (331, 64)):
    A = torch.randn(G, n, n, dtype=torch.float64, device="cuda")
    H = A + A.transpose(1, 2)
    torch.linalg.eigh(H); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        v, w = torch.linalg.eigh(H)
    torch.cuda.synchronize()
    print(f"GPU eigh n={n} batch={G}: {(time.perf_counter()-t0)/3*1e3:.2f} ms total, {(time.perf_counter()-t0)/3*1e3/G:.2f} ms each")
Hc = H[0].cpu()
t0 = time.perf_counter()
for _ in range(3):
    torch.linalg.eigh(Hc)
print(f"CPU eigh n=331: {(time.perf_counter()-t0)/3*1e3:.2f} ms")
# Cholesky-based solve for comparison
P = H[0] @ H[0] + torch.eye(331, dtype=torch.float64, device="cuda")
g = torch.randn(331, 1, dtype=torch.float64, device="cuda")
torch.linalg.cholesky(P); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    L = torch.linalg.cholesky(P); x = torch.cholesky_solve(g, L)
torch.cuda.synchronize()
print(f"GPU cholesky+solve n=331: {(time.perf_counter()-t0)/10*1e3:.2f} ms")
# eigenvalues only (what a damped Newton step needs besides a linear solve)
H1 = H[0].contiguous()
torch.linalg.eigvalsh(H1); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    ev = torch.linalg.eigvalsh(H1)
torch.cuda.synchronize()
print(f"GPU eigvalsh n=331: {(time.perf_counter()-t0)/5*1e3:.2f} ms")
t0 = time.perf_counter()
for _ in range(5):
    ev = torch.linalg.eigvalsh(Hc)
print(f"CPU eigvalsh n=331: {(time.perf_counter()-t0)/5*1e3:.2f} ms")
t0 = time.perf_counter()
for _ in range(5):
    x = torch.linalg.solve(P, g)
torch.cuda.synchronize()
print(f"GPU LU solve n=331: {(time.perf_counter()-t0)/5*1e3:.2f} ms")
