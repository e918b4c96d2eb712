"""configs[4] at batch 256 (kUpCCD CAS(8e,8o), k = 1): state, RDMs and the reverse-mode gradient
as separate timed steps (and under rocprofv3: tools/profile_config5.sh)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import auto_oo_amd as aoo
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pqc = aoo.Parameterized_circuit(8, 8, None, ansatz="kupccd", k=k)
eng = pqc._sector
rng = np.random.default_rng(6)
th = torch.tensor(rng.uniform(0, 2 * np.pi, (B, int(pqc.theta_shape))), device="cuda")
c1 = torch.tensor(rng.standard_normal((8, 8)), device="cuda")
c2 = torch.tensor(rng.standard_normal((8,) * 4), device="cuda")
def T(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, r
t, psi = T(lambda: eng.state(th)); print(f"batch {B} k {k}: state   {t:9.1f} us")
t, _ = T(lambda: eng.rdms(psi)); print(f"batch {B} k {k}: rdms    {t:9.1f} us")
t, _ = T(lambda: eng.adjoint(th, psi, c1, c2)); print(f"batch {B} k {k}: adjoint {t:9.1f} us")
if len(sys.argv) > 3:
    from auto_oo_amd._lib import debug_options
    for pr in (0, 1, 2, 3):
        with debug_options(sector_probe=pr):
            t, _ = T(lambda: eng.adjoint(th, psi, c1, c2))
        print(f"batch {B} k {k}: adjoint with sector_probe = {pr}: {t:9.1f} us")
if len(sys.argv) > 3:
    for pr in (0, 1, 2):
        with debug_options(sector_probe=pr):
            t, _ = T(lambda: eng.rdms(psi))
        print(f"batch {B} k {k}: rdms with sector_probe = {pr}: {t:9.1f} us")
if len(sys.argv) > 3:
    import ctypes
    from auto_oo_amd import _lib
    lib = _lib.load()
    if hasattr(lib, "oovqe_sector_pipe_cycles"):
        with debug_options(sector_probe=9):
            eng.adjoint(th, psi, c1, c2); torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 16)()
        lib.oovqe_sector_pipe_cycles(buf)
        v = list(buf)
        print("pipe kernel, workgroup 0, cycles [work 1 | barrier 1 | work 2 | barrier 2]: multiplier wave 0", v[0:4], " helper wave 4", v[4:8])
