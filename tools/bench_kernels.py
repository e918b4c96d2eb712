#!/usr/bin/env python3
"""Per-kernel timing on the GPU box (HIP events on torch's current stream).  Scratch tool used
while tuning; bench.py is the contract benchmark."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auto_oo_amd import ops  # noqa: E402


def timeit(fn, warm=3, reps=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3, ts[0] * 1e-3


def main():
    out = {}
    dev = "cuda"
    sizes = [int(a) for a in sys.argv[1:]] or [43, 96, 200]
    for N in sizes:
        g = torch.rand((N, N, N, N), dtype=torch.float64, device=dev) - 0.5
        C = torch.rand((N, N), dtype=torch.float64, device=dev) - 0.5
        o = torch.empty_like(g)
        w = torch.empty_like(g)
        med, best = timeit(lambda: ops.general_4index_transform(g, C, C, C, C, out=o, work=w),
                           warm=2, reps=5 if N >= 150 else 20)
        fl = 8.0 * N ** 5
        out[f"transform_N{N}"] = dict(ms=med * 1e3, best_ms=best * 1e3, tflops=fl / med / 1e12,
                                      best_tflops=fl / best / 1e12)
        print(N, out[f"transform_N{N}"], flush=True)
        # individual quarter steps
        n3 = N ** 3
        for name, args in (("step1", (g, C, 1, N, N, n3, False)), ("step2", (g, C, N, N, N, N * N, False)),
                           ("step3", (g, C, N * N, N, N, N, False)), ("step4", (g, C, n3, N, N, 1, True))):
            T, Cm, A, K, J, B, last = args
            med, best = timeit(lambda: ops.mode_contract(T, Cm, A, K, J, B, last, out=o),
                               warm=2, reps=5 if N >= 150 else 20)
            out[f"{name}_N{N}"] = dict(ms=med * 1e3, tflops=2.0 * N ** 5 / med / 1e12)
            print(N, name, out[f"{name}_N{N}"], flush=True)
        if N <= 128:
            M = 9 if N < 100 else 26
            T2 = torch.empty((N, N, M, M), dtype=torch.float64, device=dev)
            med, best = timeit(lambda: ops.cas_half_transform(g, C, M, out=T2), warm=3, reps=50)
            out[f"half_N{N}"] = dict(us=med * 1e6, best_us=best * 1e6, gbs=8.0 * N ** 4 / med / 1e9,
                                     best_gbs=8.0 * N ** 4 / best / 1e9)
            print(N, "half", out[f"half_N{N}"], flush=True)
        med, best = timeit(lambda: ops.expm(C - C.T, -1.0), warm=3, reps=20)
        out[f"expm_N{N}"] = dict(us=med * 1e6, best_us=best * 1e6)
        print(N, "expm", out[f"expm_N{N}"], flush=True)
        del g, o, w
        torch.cuda.empty_cache()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bench_kernels.json", "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
