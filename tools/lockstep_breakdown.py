"""Where does one lockstep Berry-loop step over G geometries go (tracking regime: positive definite Hessians)?
    python tools/lockstep_breakdown.py [G]
Each piece is timed back to back (device time of the piece, host submission included)."""
import contextlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import auto_oo_amd as aoo                               # noqa: E402
from auto_oo_amd import ops                             # noqa: E402
from auto_oo_amd.synthetic import synthetic_loop        # noqa: E402
import bench                                            # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pqc = aoo.Parameterized_circuit(bench.NCAS, bench.NELECAS, None, ansatz="ucc")
base, loop = synthetic_loop(bench.NAO, 20263, G, eps=0.01)
bmol = aoo.Moldata(base["int1e_ao"], base["int2e_ao"], base["overlap"], base["nuc"], bench.NELEC)
boo = aoo.OO_pqc(pqc, bmol, bench.NCAS, bench.NELECAS, oao_mo_coeff=base["oao_mo_coeff"], freeze_active=True)
with contextlib.redirect_stdout(sys.stderr):
    e_l, th_l, _, _, _ = boo.full_optimization(torch.full((pqc.theta_shape,), 0.1, dtype=torch.float64, device="cuda"),
                                               max_iterations=80, conv_tol=1e-11, verbose=None)
theta0, c_star = th_l[-1], boo.oao_mo_coeff
mols = [aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], bench.NELEC) for P in loop]
batch = aoo.OO_pqc_batch(pqc, mols, bench.NCAS, bench.NELECAS, oao_mo_coeffs=[c_star] * G, freeze_active=True)
thetas0 = theta0.reshape(1, -1).repeat(G, 1).contiguous()
c_saved = batch.oao_mo_coeff.clone()
bopt = aoo.BatchedNewtonStep(verbose=0)
nt = batch.n_theta


def T(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, r


def restore():
    batch.oao_mo_coeff.copy_(c_saved)
    batch.refresh_mo_coeff()


t, (E, g, H) = T(lambda: batch.energy_gradient_hessian(thetas0)); print(f"G={G}: energy_gradient_hessian   {t:8.1f} us")
t, r = T(lambda: ops.newton_direction(H, g, defer_lowest=True, want_info=True)); print(f"newton_direction (deferred)        {t:8.1f} us (back to back: the side route bounds it)")
dp = r[0]
kap = (dp[:, nt:] * 1.0).contiguous()
t, _ = T(lambda: batch.rotated_mo_coeff(kap)); print(f"rotated_mo_coeff (expm + C U)      {t:8.1f} us")
t, _ = T(lambda: batch.energy(thetas0, kap)); print(f"trial energy (rotate + evaluate)   {t:8.1f} us")
t, _ = T(lambda: batch.energy(thetas0)); print(f"energy (evaluate only)             {t:8.1f} us")
t, _ = T(lambda: (batch.rotate_(kap), restore())); print(f"rotate_ + restore                  {t:8.1f} us")
t, _ = T(lambda: E.tolist()); print(f"one readback                       {t:8.1f} us")


def step():
    out = batch.damped_newton_step(thetas0, bopt, defer_lowest=True)
    restore()
    return out


t, _ = T(step, 10); print(f"damped_newton_step + restore       {t:8.1f} us")
torch.cuda.synchronize()
# host-side cost alone: the same step with nothing to wait for is not measurable; count the launches instead
from auto_oo_amd import _lib                          # noqa: E402
