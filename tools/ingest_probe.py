"""oovqe_eri_ingest on a stack of G synthetic geometries (N = 43): time per stack, and -- under rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE (tools/pmc_ingest.sh) -- its HBM traffic against the algorithmic bytes (tensor read once + packed copy written)."""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from auto_oo_amd import ops  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pqc, batch, single, thetas = bench.build_geometries(list(range(G)))
g = batch.int2e_ao
out = batch._eri_packed
for _ in range(3):
    ops.eri_ingest(g, out=out)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    flags, _ = ops.eri_ingest(g, out=out)
    ts.append(time.perf_counter() - t0)
N = g.shape[-1]
alg = G * 8.0 * (N ** 4 + out.shape[1])
t = sorted(ts)[len(ts) // 2]
print(f"G={G}: eri_ingest {t * 1e3:.3f} ms per stack (flags read back), algorithmic {alg / 1e9:.3f} GB -> {alg / t / 1e12:.2f} TB/s; "
      f"flags {set(flags)}")
t0 = time.perf_counter()
for _ in range(10):
    ops.eri_flags(g)
    ops.eri_pack(g)
torch.cuda.synchronize()
print(f"       check (one-pass kernel without the copy) + separate pack: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
