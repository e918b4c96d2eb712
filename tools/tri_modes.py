"""Packed-triangle stage 1 realisations side by side (debug option tri_mode): results against
mode 1 and time per batched call; HIP-event time of the stage-1 launch alone."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from auto_oo_amd import _lib, ops
G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pqc, batch, single, thetas = bench.build_geometries([g % 16 for g in range(G)])
lib = _lib.load()
ref = None
for mode in (1, 2, 3, 4, 1):
    lib.oovqe_debug_set_option(b"tri_mode", mode)
    out = batch.energy_and_gradient(thetas)
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    diff = float((out - ref).abs().max())
    for _ in range(10):
        batch.energy_and_gradient(thetas)
    torch.cuda.synchronize()
    ops.profile_begin()
    t0 = time.perf_counter()
    for _ in range(40):
        batch.energy_and_gradient(thetas)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    ms, cnt, _ = ops.profile_end()
    print(f"tri_mode {mode}: max|diff vs mode 1| = {diff:.2e}; call {dt * 1e6:.1f} us "
          f"({G / dt:.0f} evals/s); stage 1 {ms / cnt * 1e3:.1f} us", flush=True)
