"""Where the timed region of bench.py's 20-step headline spends its time beyond the calls themselves."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from auto_oo_amd import ops  # noqa: E402
from auto_oo_amd.parallel import gather_results  # noqa: E402

FREE_INSIDE = False
G = 256
pqc, batch, single, thetas = bench.build_geometries(list(range(G)))
my = list(range(G))
results = torch.zeros((G, 1 + batch.n_theta + batch.n_kappa), dtype=torch.float64, device="cuda")
ops.profile_begin(); ops.profile_end()


def region(steps, defer, events, tail):
    torch.cuda.synchronize()
    if events:
        ops.profile_begin()
    t0 = time.perf_counter()
    pend, last = [], None
    for _ in range(steps):
        if defer:
            pend.append(batch.energy_and_gradient(thetas, defer=True))
        else:
            last = batch.energy_and_gradient(thetas)
    t1 = time.perf_counter()
    if defer:
        for p_ in pend[-2:]:
            p_.wait()
        last = pend[-1].result()
    if tail:
        results.copy_(last)
        if FREE_INSIDE:
            del pend, last
        gather_results(results, my, G, None)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    if events:
        ops.profile_end()
    return (t3 - t0) / steps * 1e6, (t1 - t0) / steps * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6


for _ in range(3):
    region(20, True, True, True)
print("single shots, 20 steps:", [round(region(20, True, True, True)[0], 1) for _ in range(8)], flush=True)
t_w = time.perf_counter()
while time.perf_counter() - t_w < 0.5:
    region(8, True, False, False)
region(5, True, False, True)
print("after the bench's priming sequence:", round(region(20, True, True, True)[0], 1), flush=True)
import gc
gc.collect()
print("again:", [round(region(20, True, True, True)[0], 1) for _ in range(3)], flush=True)
FREE_INSIDE = True
print("result tensors freed inside the timed region:", [round(region(20, True, True, True)[0], 1) for _ in range(4)], flush=True)
print("  in-order:", [round(region(20, False, True, True)[0], 1) for _ in range(4)], flush=True)
FREE_INSIDE = False
print("  in-order, freed outside:", [round(region(20, False, True, True)[0], 1) for _ in range(4)], flush=True)
for steps in ():
    for defer in (False, True):
        for events in (False, True):
            for tail in (False, True):
                r = min(region(steps, defer, events, tail) for _ in range(3))
                print(f"steps={steps:3d} defer={int(defer)} events={int(events)} copy+gather={int(tail)}: {r[0]:7.1f} us/step "
                      f"(enqueue {r[1]:5.1f} us/step, join+tail {r[2]:6.1f} us, final sync {r[3]:7.1f} us)", flush=True)
