"""What the headline's 20 deferred steps cost beyond 40 un-profiled calls: number of calls, HIP-event brackets."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from auto_oo_amd import ops  # noqa: E402

GEOMS = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pqc, batch, single, thetas = bench.build_geometries(list(range(GEOMS)))


def run(n, deferred, prof):
    torch.cuda.synchronize()
    if prof:
        ops.profile_begin()
    t = time.perf_counter()
    outs = [batch.energy_and_gradient(thetas, defer=True) if deferred else batch.energy_and_gradient(thetas)
            for _ in range(n)]
    if deferred:
        outs = [o.result() for o in outs]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    ms = cnt = 0
    if prof:
        ms, cnt, _ = ops.profile_end()
    return dt * 1e6, (ms / cnt * 1e3 if cnt else 0.0)


for deferred in (False, True):
    for prof in (False, True):
        for n in (20, 40):
            run(8, deferred, prof)
            best = min(run(n, deferred, prof) for _ in range(4))
            print(f"deferred={int(deferred)} events={int(prof)} calls={n:3d}: {best[0]:7.1f} us/call  {GEOMS / best[0] * 1e6:9.0f} evals/s"
                  f"  sweep {best[1]:6.1f} us", flush=True)
