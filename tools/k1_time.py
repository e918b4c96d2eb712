"""Time the N^4 transform (four chained mode contractions) and each quarter step alone.
    python tools/k1_time.py [N] [reps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auto_oo_amd import ops, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = torch.rand((N, N, N, N), dtype=torch.float64, device="cuda") - 0.5
C = torch.rand((N, N), dtype=torch.float64, device="cuda") - 0.5
o = torch.empty_like(g); w = torch.empty_like(g)
lib = _lib.load()
def T(f):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]
t = T(lambda: ops.general_4index_transform(g, C, C, C, C, out=o, work=w))
print(f"N={N}: transform {t * 1e3:.2f} ms = {8 * N ** 5 / t / 1e12:.2f} TFLOP/s = {8 * N ** 5 / t / 78.6e12:.3f} of 78.6")
# quarter steps: mode m contracted (A = N^m, B = N^(3-m))
for m in range(4):
    A, B = N ** m, N ** (3 - m)
    last = 1 if B == 1 else 0
    f = lambda: ops.check(lib.oovqe_mode_contract(ops.dptr(g), ops.dptr(C), ops.dptr(o), A, N, N, B, N, last,
                                                  ops.stream_ptr()), "mc")
    t = T(f)
    print(f"  mode {m} (A={A}, B={B}): {t * 1e3:.2f} ms = {2 * N ** 5 / t / 1e12:.2f} TFLOP/s", flush=True)
