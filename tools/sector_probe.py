"""Where does sector_rdm_fused_kernel spend its time?  Timing-only builds of the same launch (debug option
sector_probe: 1 = chunks not rebuilt, 2 = no MFMA phase; results are wrong on purpose)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import auto_oo_amd as aoo
from auto_oo_amd._lib import debug_options
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pqc = aoo.Parameterized_circuit(8, 8, None, ansatz="kupccd", k=1)
eng = pqc._sector
rng = np.random.default_rng(6)
th = torch.tensor(rng.uniform(0, 2 * np.pi, (B, int(pqc.theta_shape))), device="cuda")
psi = eng.state(th)
def T(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print(f"batch {B}: rdms full       {T(lambda: eng.rdms(psi)):8.1f} us")
with debug_options(sector_probe=1):
    print(f"batch {B}: no chunk build  {T(lambda: eng.rdms(psi)):8.1f} us")
with debug_options(sector_probe=2):
    print(f"batch {B}: no MFMA phase   {T(lambda: eng.rdms(psi)):8.1f} us")
