"""Where does ONE sequential Berry-loop step of one geometry go (the reference's loop body, oo_pqc.py:172-196:
full_gradient, full_hessian, damped Newton step, closing energy)?  tracking regime (positive definite Hessian).
    python tools/sequential_breakdown.py"""
import contextlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import auto_oo_amd as aoo
from auto_oo_amd.synthetic import synthetic_loop
import bench

pqc = aoo.Parameterized_circuit(bench.NCAS, bench.NELECAS, None, ansatz="ucc")
base, loop = synthetic_loop(bench.NAO, 20263, 2, eps=0.01)
bmol = aoo.Moldata(base["int1e_ao"], base["int2e_ao"], base["overlap"], base["nuc"], bench.NELEC)
boo = aoo.OO_pqc(pqc, bmol, bench.NCAS, bench.NELECAS, oao_mo_coeff=base["oao_mo_coeff"], freeze_active=True)
with contextlib.redirect_stdout(sys.stderr):
    e_l, th_l, _, _, _ = boo.full_optimization(torch.full((pqc.theta_shape,), 0.1, dtype=torch.float64, device="cuda"),
                                               max_iterations=80, conv_tol=1e-11, verbose=None)
theta0, c_star = th_l[-1], boo.oao_mo_coeff
P = loop[0]
mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], bench.NELEC)
oo = aoo.OO_pqc(pqc, mol, bench.NCAS, bench.NELECAS, oao_mo_coeff=c_star, freeze_active=True)
batch1 = aoo.OO_pqc_batch(pqc, [mol], bench.NCAS, bench.NELECAS, oao_mo_coeffs=[c_star], freeze_active=True)
opt = aoo.NewtonStep(verbose=0)


def T(f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, r


kappa = torch.zeros(oo.n_kappa, dtype=torch.float64, device="cuda")
t, grad = T(lambda: oo.full_gradient(theta0)); print(f"full_gradient                      {t:8.1f} us")
t, hess = T(lambda: oo.full_hessian(theta0)); print(f"full_hessian (three blocks)        {t:8.1f} us")
t, r = T(lambda: batch1.energy_gradient_hessian(theta0.reshape(1, -1))); print(f"  the batched call with one geometry {t:6.1f} us")
print("  max |dH| between the two:", (r[2][0] - hess).abs().max().item(), " |dg|:", (r[1][0] - grad).abs().max().item())
t, (dp, low) = T(lambda: opt.newton_step(grad, hess)); print(f"newton_step (direction + readback) {t:8.1f} us, lowest eigenvalue {low:.4f}")
t, new = T(lambda: opt.damped_newton_step(oo.energy_from_parameters, (theta0, kappa), grad, hess))
print(f"damped_newton_step                 {t:8.1f} us")
t, _ = T(lambda: oo.energy_from_parameters(new[0][0], new[0][1])); print(f"energy_from_parameters(theta, kappa){t:7.1f} us")
t, _ = T(lambda: oo.energy_from_parameters(theta0)); print(f"energy_from_parameters(theta)      {t:8.1f} us")


def one():
    g = oo.full_gradient(theta0)
    h = oo.full_hessian(theta0)
    nw, eig = opt.damped_newton_step(oo.energy_from_parameters, (theta0, kappa), g, h)
    return oo.energy_from_parameters(nw[0], nw[1])


t, _ = T(one); print(f"the whole unit of work             {t:8.1f} us")


def loop(n, defer):
    pend = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g = oo.full_gradient(theta0)
        h = oo.full_hessian(theta0)
        nw, eig = opt.damped_newton_step(oo.energy_from_parameters, (theta0, kappa), g, h, defer_lowest=defer)
        pend.append(eig)
        e = oo.energy_from_parameters(nw[0], nw[1])
    ev = torch.cuda.Event(); ev.record(); ev.synchronize()
    t1 = time.perf_counter()
    vals = [float(p) for p in pend]
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


for defer in (False, True):
    loop(4, defer)
    a, b = loop(32, defer)
    print(f"32 units in a row, eigenvalues {'collected and joined at the end' if defer else 'joined per step'}: "
          f"{a:8.1f} us per unit until the main stream is done, {b:8.1f} us with the eigenvalues")
