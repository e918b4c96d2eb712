"""Row f1: the damped Newton step on the device (oovqe_newton_direction + NewtonStep /
BatchedNewtonStep) against the reference's algorithm restated on CPU (oracle.OracleNewtonStep:
two eigh calls + explicit inverse, utils/newton_raphson.py:78-211), the reference's own Newton
property tests (test/utils/test_newton_raphson.py:99-130) run on the device, and BASELINE
configs[3]'s unit of work at N = 43: full 331 x 331 Hessian (all three blocks) and one damped
Newton step against the oracle, sequential == lockstep on 64 geometries."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import auto_oo_amd as aoo            # noqa: E402
from auto_oo_amd import ops          # noqa: E402
from oracle import cpu_ref as R      # noqa: E402
from tests.test_api_gpu import _setup   # noqa: E402
from tests._oracle_cache import oracle_values, pack_sym, unpack_sym   # noqa: E402


def _reference_direction(H, g, lambda_min=1e-6, mu=1e-6, rho=1.1, aug=True):
    opt = R.OracleNewtonStep(mu=mu, rho=rho, lambda_min=lambda_min, aug=aug)
    return opt.newton_step(g, H)


def _sym(rng, n, kind):
    A = rng.standard_normal((n, n))
    if kind == "indefinite":
        S = A + A.T
    elif kind == "pd":
        S = A @ A.T / n + 0.5 * np.eye(n)
    elif kind == "near_singular":          # lowest eigenvalue in (0, lambda_min): shifted too
        Q, _ = np.linalg.qr(A)
        ev = np.linspace(0.5, 3.0, n)
        ev[0] = 3e-7
        S = (Q * ev) @ Q.T
    else:                                   # diagonal: reflectors with tau = 0
        S = np.diag(rng.standard_normal(n))
    return 0.5 * (S + S.T)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 9, 10, 11, 16, 17, 18, 33, 100, 331, 480, 601, 672])
@pytest.mark.parametrize("kind", ["indefinite", "pd", "near_singular", "diagonal"])
def test_newton_direction_vs_eigh(n, kind):
    rng = np.random.default_rng(1000 * n + len(kind))
    H = torch.tensor(_sym(rng, n, kind))
    g = torch.tensor(rng.standard_normal(n))
    dp_ref, low_ref = _reference_direction(H, g)
    dp, low, nu = ops.newton_direction(H.cuda(), g.cuda())
    scale = float(H.abs().max())
    assert abs(low.item() - low_ref) < 1e-12 * max(1.0, scale) * n
    expect_nu = 1e-6 + 1.1 * abs(low_ref) if low_ref < 1e-6 else 0.0
    assert abs(nu.item() - expect_nu) < 1e-11 * max(1.0, scale) * n
    # compare as residuals of the shifted system (the step itself is ill-conditioned by design
    # when the shifted lowest eigenvalue is ~1e-6): (H + nu) dp + g = 0
    Hs = H + nu.item() * torch.eye(n, dtype=torch.float64)
    res = (Hs @ dp.cpu() + g).abs().max() / (1.0 + g.abs().max())
    assert res < 1e-9, res
    cond = float(torch.linalg.cond(Hs))
    assert (dp.cpu() - dp_ref).abs().max() <= 1e-13 * cond * (1.0 + dp_ref.abs().max()) * n


@pytest.mark.parametrize("n,kind", [(673, "indefinite"), (700, "pd"), (1000, "near_singular"), (1544, "indefinite"),
                                    (1600, "pd")])
def test_newton_direction_beyond_one_launch(n, kind):
    """n > 672 (the operands of a panel no longer fit LDS): the host-orchestrated band reduction (four
    launches per panel of 8 columns) + the same band LDL^T solve kernel, against eigh -- no rocSOLVER
    fallback below n = 5128 (orbital spaces of N = 200)."""
    rng = np.random.default_rng(n)
    H = torch.tensor(_sym(rng, n, kind))
    g = torch.tensor(rng.standard_normal(n))
    dp_ref, low_ref = _reference_direction(H, g)
    dp, low, nu = ops.newton_direction(H.cuda(), g.cuda())
    scale = float(H.abs().max())
    assert abs(low.item() - low_ref) < 1e-12 * max(1.0, scale) * n
    Hs = H + nu.item() * torch.eye(n, dtype=torch.float64)
    res = (Hs @ dp.cpu() + g).abs().max() / (1.0 + g.abs().max())
    assert res < 1e-9, res
    cond = float(torch.linalg.cond(Hs))
    assert (dp.cpu() - dp_ref).abs().max() <= 1e-13 * cond * (1.0 + dp_ref.abs().max()) * n
    # two problems in one call
    H2 = torch.stack((H, H + 0.5 * torch.eye(n, dtype=torch.float64))).cuda()
    g2 = torch.stack((g, -g)).cuda()
    dp2, low2, _ = ops.newton_direction(H2, g2)
    assert torch.equal(dp2[0], dp) and abs(low2[1].item() - low2[0].item() - 0.5) < 1e-10 * max(1.0, scale)


def test_newton_direction_batch_and_no_augmentation():
    rng = np.random.default_rng(7)
    n, G = 58, 19
    Hs = torch.tensor(np.stack([_sym(rng, n, "indefinite" if k % 2 else "pd") for k in range(G)]))
    gs = torch.tensor(rng.standard_normal((G, n)))
    dp, low, nu = ops.newton_direction(Hs.cuda(), gs.cuda())
    dp_na, low_na, nu_na = ops.newton_direction(Hs.cuda(), gs.cuda(), aug=False)
    # (aug=False runs on the one-workgroup kernel with its pivoted solve: same eigenvalues to rounding)
    assert (low - low_na).abs().max() < 1e-12 and float(nu_na.abs().max()) == 0.0
    for k in range(G):
        d1, l1, _ = ops.newton_direction(Hs[k].cuda(), gs[k].cuda())
        # batch == one by one up to rounding (round 3: a problem alone gets more workgroups and another
        # grid of shifts than inside a batch; the same call twice gives the same bits)
        assert abs(l1.item() - low[k].item()) < 1e-13 * max(1.0, float(Hs[k].abs().max()))
        assert (d1 - dp[k]).abs().max() < 1e-9 * (1 + dp[k].abs().max())
        dr, lr = _reference_direction(Hs[k], gs[k])
        assert abs(low[k].item() - lr) < 1e-11
        assert (dp[k].cpu() - dr).abs().max() < 1e-8 * (1 + dr.abs().max())
        dr_na, _ = _reference_direction(Hs[k], gs[k], aug=False)      # plain H^-1 g, indefinite or not
        assert (dp_na[k].cpu() - dr_na).abs().max() < 1e-8 * (1 + dr_na.abs().max())
    nmax = aoo._lib.load().oovqe_newton_direction_max_n()
    with pytest.raises(aoo._lib.OovqeError):
        ops.newton_direction(torch.eye(nmax + 1, dtype=torch.float64, device="cuda"),
                             torch.ones(nmax + 1, dtype=torch.float64, device="cuda"))


@pytest.mark.parametrize("n,G", [(331, 1), (331, 5), (331, 64), (58, 19), (100, 300), (9, 3)])
def test_newton_direction_two_stage_equals_one_workgroup_kernel(n, G):
    """The default kernel (two stages, several workgroups per problem: band reduction on the matrix
    cores + band LDL^T multisection) against the round-2 one-workgroup tridiagonalisation kernel
    (debug option newton_one_wg) on the same Hessians: same lowest eigenvalues, same shifts, same
    directions (as residuals), whatever the number of workgroups a problem gets (1 ... 32)."""
    rng = np.random.default_rng(31 * n + G)
    Hs = torch.tensor(np.stack([_sym(rng, n, ("indefinite", "pd", "near_singular")[k % 3]) for k in range(G)])).cuda()
    gs = torch.tensor(rng.standard_normal((G, n))).cuda()
    dp, low, nu = ops.newton_direction(Hs, gs)
    with aoo._lib.debug_options(newton_one_wg=1):
        dp1, low1, nu1 = ops.newton_direction(Hs, gs)
    scale = float(Hs.abs().amax())
    assert (low - low1).abs().max() < 1e-12 * max(1.0, scale) * n
    assert (nu - nu1).abs().max() < 1e-11 * max(1.0, scale) * n
    eye = torch.eye(n, dtype=torch.float64, device="cuda")
    for k in range(0, G, max(1, G // 8)):
        Hk = Hs[k] + nu[k] * eye
        res = (Hk @ dp[k] + gs[k]).abs().max() / (1.0 + gs[k].abs().max())
        assert res < 1e-9, (k, float(res))
        cond = float(torch.linalg.cond(Hk))
        assert (dp[k] - dp1[k]).abs().max() <= 1e-13 * cond * (1.0 + dp1[k].abs().max()) * n
    # a problem alone gets more workgroups than inside a batch: other summation orders, same numbers
    d0, l0, _ = ops.newton_direction(Hs[0], gs[0])
    cond0 = float(torch.linalg.cond(Hs[0] + nu[0] * eye))
    assert (d0 - dp[0]).abs().max() <= 1e-13 * cond0 * (1.0 + dp[0].abs().max()) * n
    assert abs(l0.item() - low[0].item()) < 1e-12 * max(1.0, scale) * n
    # and the same call twice gives the same bits
    dp_again, low_again, _ = ops.newton_direction(Hs, gs)
    assert torch.equal(dp_again, dp) and torch.equal(low_again, low)


def _quartic_problems(G, n, seed):
    rng = np.random.default_rng(seed)
    mats, x0s = [], []
    for g in range(G):
        A = rng.standard_normal((n, n))
        S = A + A.T if g % 2 else A @ A.T + 0.1 * np.eye(n)      # odd g: indefinite at the start
        mats.append(torch.tensor(S))
        x0s.append(torch.tensor(rng.standard_normal(n)) * (3.0 if g >= G - 2 else 0.3))
    return mats, x0s


def test_damped_newton_step_vs_oracle_and_batched():
    """NewtonStep.damped_newton_step == oracle step (incl. augmented Hessians and backtracking);
    BatchedNewtonStep == NewtonStep problem by problem."""
    n, G = 12, 6
    mats, x0s = _quartic_problems(G, n, 11)

    def make(S):
        def f(a, b):
            x = torch.cat((a, b))
            Sx = S.to(x.device)
            return 0.5 * x @ Sx @ x + 0.25 * (x ** 4).sum()       # quartic term: line search matters
        return f
    fns = [make(S) for S in mats]
    grads = torch.stack([mats[g] @ x0s[g] + x0s[g] ** 3 for g in range(G)])
    hess = torch.stack([mats[g] + torch.diag(3 * x0s[g] ** 2) for g in range(G)])
    params_cpu = [(x[:5].clone(), x[5:].clone()) for x in x0s]
    params_dev = [(a.cuda(), b.cuda()) for a, b in params_cpu]
    opt, ref = aoo.NewtonStep(verbose=0), R.OracleNewtonStep()
    new_b, low_b = aoo.BatchedNewtonStep(verbose=0).damped_newton_steps(fns, params_dev, grads.cuda(),
                                                                       hess.cuda())
    backtracked = 0
    for g in range(G):
        new_s, low_s = opt.damped_newton_step(fns[g], params_dev[g], grads[g].cuda(), hess[g].cuda())
        new_r, low_r = ref.damped_newton_step(fns[g], params_cpu[g], grads[g], hess[g])
        assert isinstance(low_s, float) and abs(low_s - low_r) < 1e-11
        full = torch.cat([x.flatten() for x in new_r]) - torch.cat([x.flatten() for x in params_cpu[g]])
        dp_r, _ = ref.newton_step(grads[g], hess[g])
        backtracked += int((full - dp_r).abs().max() > 1e-6)
        for a, b, c in zip(new_s, new_r, new_b[g]):
            assert a.shape == b.shape and (a.cpu() - b).abs().max() < 1e-9 * max(1.0, float(b.abs().max()))
            assert (c - a).abs().max() < 1e-12 * max(1.0, float(b.abs().max()))
        assert abs(low_b[g].item() - low_s) < 1e-13
    assert backtracked >= 1                      # the far starts really exercise the line search
    # single-argument objective: a flat tensor comes back (newton_raphson.py:187-190)
    f1 = lambda x: fns[0](x[:5], x[5:])          # noqa: E731
    new1, _ = opt.damped_newton_step(f1, (x0s[0].cuda(),), grads[0].cuda(), hess[0].cuda())
    assert isinstance(new1, torch.Tensor) and new1.shape == (n,)
    # a non-descent direction with a rejected first trial is an assertion, as in the reference
    with pytest.raises(AssertionError):
        opt.backtracking(lambda x: (x ** 2).sum(), (torch.ones(3, dtype=torch.float64, device="cuda"),),
                         torch.ones(3, dtype=torch.float64, device="cuda"),
                         torch.ones(3, dtype=torch.float64, device="cuda") * 2)


def _newton_optimize(cost, x0, max_iterations, conv_tol, **kw):
    from torch.autograd.functional import jacobian, hessian
    opt = aoo.NewtonStep(verbose=0, **kw)
    theta = x0
    energies = [cost(theta).item()]
    for n in range(max_iterations):
        theta, _ = opt.damped_newton_step(cost, (theta,), jacobian(cost, theta), hessian(cost, theta))
        energies.append(cost(theta).item())
        if n > 1 and abs(energies[-1] - energies[-2]) < conv_tol:
            break
    return energies, theta


@pytest.mark.parametrize("dim,max_iterations,conv_tol,lambda_min,rho,mu",
                         [(2, 20, 1e-12, 1e-6, 2, 1e-4), (4, 20, 1e-12, 1e-6, 2, 1e-4),
                          (8, 50, 1e-10, 1e-6, 3, 1e-4)])
def test_reference_property_type_a_on_device(dim, max_iterations, conv_tol, lambda_min, rho, mu):
    """test/utils/test_newton_raphson.py:99-116 with the device-side step."""
    gen = torch.Generator().manual_seed(dim)
    a = torch.rand(dim, dim, generator=gen, dtype=torch.float64) - 0.5
    a = (a.T + a).cuda()
    va = torch.linalg.eigvalsh(a)

    def cost(x):
        u = torch.linalg.matrix_exp(-aoo.vector_to_skew_symmetric(x))
        return ((u.T @ a @ u - torch.diag(va)) ** 2).sum()
    x0 = (1e-5 * (torch.rand(dim * (dim - 1) // 2, generator=gen, dtype=torch.float64) - 0.5)).cuda()
    energies, x = _newton_optimize(cost, x0, max_iterations, conv_tol, aug=True,
                                   lambda_min=lambda_min, rho=rho, mu=mu)
    assert abs(energies[-1]) < 1e-8
    u = torch.linalg.matrix_exp(-aoo.vector_to_skew_symmetric(x))
    assert torch.allclose(u.T @ a @ u, torch.diag(va), atol=1e-6)


@pytest.mark.parametrize("t,max_iterations", [(4.0, 10), (3.0, 10), (0.00004, 100)])
def test_reference_property_type_b_on_device(t, max_iterations):
    """test/utils/test_newton_raphson.py:119-130: plain damped Newton with backtracking, n = 1."""
    def cost(x):
        return (-t * torch.log(torch.abs(x)) + torch.abs(x) - t + t * np.log(t)).sum()
    energies, x = _newton_optimize(cost, torch.tensor([10.0], dtype=torch.float64, device="cuda"),
                                   max_iterations, 1e-12, aug=False)
    assert abs(energies[-1]) < 1e-8


def oracle_config3_n43():
    """The oracle's side of configs[3] at N = 43 (seed 20262, theta = 0.1): full gradient, full 331 x 331 Hessian and
    one damped Newton step -- ~5 minutes of host CPU, kept as tests/golden/oracle_cache/config3_n43_s20262.npz."""
    def compute():
        P = R.synthetic_problem(43, 20262)
        omol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 16)
        ooo = R.OracleOOPQC(R.OraclePQC(3, 4, "ucc"), omol, 3, 4, P["oao_mo_coeff"])
        theta = torch.full((4,), 0.1, dtype=torch.float64)
        g_ref = ooo.full_gradient(theta)
        h_ref = ooo.full_hessian(theta)
        kappa = torch.zeros(ooo.n_kappa, dtype=torch.float64)
        new_r, low_r = R.OracleNewtonStep().damped_newton_step(ooo.energy_from_parameters, (theta, kappa), g_ref, h_ref)
        return {"gradient": g_ref.numpy(), "hessian_triu": pack_sym(h_ref.numpy()),
                "hessian_asymmetry": np.array(float((h_ref - h_ref.T).abs().max())),
                "new_theta": new_r[0].numpy(), "new_kappa": new_r[1].numpy(), "lowest_eigenvalue": np.array(low_r),
                "energy": np.array(ooo.energy_from_parameters(theta).item()),
                "new_energy": np.array(ooo.energy_from_parameters(new_r[0], new_r[1]).item())}
    return oracle_values("config3_n43_s20262", compute)


def test_config3_unit_of_work_at_cc_pvdz_shape():
    """BASELINE configs[3] at N = 43: energy + full gradient + full 331 x 331 Hessian (theta-theta,
    kappa-theta, kappa-kappa) + one damped Newton step, against the oracle (1e-8 abs, 1e-9 Ha)."""
    ooo, opqc, oo, pqc = _setup(43, 20262)
    ref = oracle_config3_n43()
    assert float(ref["hessian_asymmetry"]) < 1e-9
    theta = torch.full((pqc.theta_shape,), 0.1, dtype=torch.float64)
    grad = oo.full_gradient(theta)
    hess = oo.full_hessian(theta)
    g_ref = torch.tensor(ref["gradient"])
    h_ref = torch.tensor(unpack_sym(ref["hessian_triu"]))
    nt = pqc.theta_shape
    assert hess.shape == (331, 331) and h_ref.shape == (331, 331)
    assert abs(oo.energy_from_parameters(theta).item() - float(ref["energy"])) < 1e-9
    assert (grad.cpu() - g_ref).abs().max() < 1e-8
    assert (hess[:nt, :nt].cpu() - h_ref[:nt, :nt]).abs().max() < 1e-8      # circuit-circuit
    assert (hess[nt:, :nt].cpu() - h_ref[nt:, :nt]).abs().max() < 1e-8      # orbital-circuit
    assert (hess[:nt, nt:].cpu() - h_ref[:nt, nt:]).abs().max() < 1e-8
    assert (hess[nt:, nt:].cpu() - h_ref[nt:, nt:]).abs().max() < 1e-8      # orbital-orbital
    kappa = torch.zeros(oo.n_kappa, dtype=torch.float64)
    new, low = aoo.NewtonStep(verbose=0).damped_newton_step(
        oo.energy_from_parameters, (theta.cuda(), kappa.cuda()), grad, hess)
    assert abs(low - float(ref["lowest_eigenvalue"])) < 1e-9
    e_new = oo.energy_from_parameters(new[0], new[1]).item()
    assert abs(e_new - float(ref["new_energy"])) < 1e-9
    assert e_new < oo.energy_from_parameters(theta).item()
    assert (new[0].cpu() - torch.tensor(ref["new_theta"])).abs().max() < 1e-7
    assert (new[1].cpu() - torch.tensor(ref["new_kappa"])).abs().max() < 1e-7


def test_deferred_lowest_eigenvalue_of_single_steps():
    """NewtonStep.damped_newton_step(defer_lowest=True): the same new parameters, the eigenvalue as an
    ops.PendingLowest that float() joins -- for a positive definite Hessian (computed beside the line search) and for
    an indefinite one (known before the direction: already there); OO_pqc.full_optimization collects them that way
    and still returns plain floats (oo_pqc.py:155-207)."""
    rng = np.random.default_rng(12)
    n = 60
    x0 = torch.tensor(rng.standard_normal(n)).cuda()
    for kind in ("pd", "indefinite"):
        Hm = torch.tensor(_sym(rng, n, kind)).cuda()
        b = torch.tensor(rng.standard_normal(n)).cuda()
        fn = lambda x: 0.5 * (x * (Hm @ x)).sum() + (b * x).sum() + 0.05 * (x ** 4).sum()   # noqa: E731
        g = Hm @ x0 + b + 0.2 * x0 ** 3
        Hx = Hm + torch.diag(0.6 * x0 ** 2)
        opt = aoo.NewtonStep(verbose=0)
        new0, low0 = opt.damped_newton_step(fn, (x0,), g, Hx)
        new1, low1 = opt.damped_newton_step(fn, (x0,), g, Hx, defer_lowest=True)
        assert isinstance(low0, float) and isinstance(low1, ops.PendingLowest)
        assert torch.equal(new0, new1)
        assert float(low1) == low0 and low1.item() == low0
    ooo, opqc, oo, pqc = _setup(13, 20261)
    theta = torch.full((pqc.theta_shape,), 0.1, dtype=torch.float64)
    e_l, th_l, k_l, c_l, eig_l = oo.full_optimization(theta, max_iterations=4, verbose=None)
    assert len(eig_l) == len(e_l) and all(isinstance(e, float) for e in eig_l)


def test_full_optimization_one_call_iterations_equal_the_loop_of_calls():
    """OO_pqc.full_optimization (oo_pqc.py:155-207) runs every iteration as one library call on a stack of one
    geometry; ``optimization_by_calls`` keeps the reference's sequence of calls (gradient, Hessian, damped Newton
    step, rotation, closing energy).  Same energies, parameters, orbitals and eigenvalues to rounding, the same
    number of iterations, the reference's list quirks."""
    outs = []
    for by_calls in (True, False):
        ooo, opqc, oo, pqc = _setup(13, 20261, freeze_active=True)
        oo.optimization_by_calls = by_calls
        theta0 = torch.full((pqc.theta_shape,), 0.1, dtype=torch.float64)
        outs.append((oo.full_optimization(theta0, max_iterations=60, conv_tol=1e-11, verbose=None), oo))
        assert ("_stack1" in oo.__dict__) == (not by_calls)
    (a, oo_a), (b, oo_b) = outs
    assert len(a[0]) == len(b[0]) < 60
    assert np.abs(np.array(a[0]) - np.array(b[0])).max() < 1e-10
    assert np.abs(np.array(a[4]) - np.array(b[4])).max() < 1e-9 and all(isinstance(e, float) for e in b[4])
    for ta, tb, ca, cb in zip(a[1], b[1], a[3], b[3]):
        assert (ta - tb).abs().max() < 1e-8 and (ca - cb).abs().max() < 1e-8
        assert tb.shape == ta.shape
    assert b[2][-1] is b[1][-1]
    assert (oo_a.oao_mo_coeff - oo_b.oao_mo_coeff).abs().max() < 1e-8
    assert abs(oo_b.energy_from_parameters(b[1][-1]).item() - b[0][-1]) < 1e-11


def test_lockstep_newton_equals_sequential_on_64_geometries():
    """configs[3]: 64 geometries stepped one by one (NewtonStep) and in lockstep (BatchedNewtonStep)."""
    from auto_oo_amd.synthetic import synthetic_problem
    N, G = 43, 64
    pqc = aoo.Parameterized_circuit(3, 4, None, ansatz="ucc")
    objs = []
    for g in range(G):
        P = synthetic_problem(N, 20262 + 1000 * g)
        mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 16)
        objs.append(aoo.OO_pqc(pqc, mol, 3, 4, oao_mo_coeff=P["oao_mo_coeff"]))
    theta0 = torch.full((pqc.theta_shape,), 0.1, dtype=torch.float64, device="cuda")
    kap = [torch.zeros(oo.n_kappa, dtype=torch.float64, device="cuda") for oo in objs]
    grads = torch.stack([oo.full_gradient(theta0) for oo in objs])
    hess = torch.stack([oo.full_hessian(theta0) for oo in objs])
    new_b, low_b = aoo.BatchedNewtonStep(verbose=0).damped_newton_steps(
        [oo.energy_from_parameters for oo in objs], [(theta0, k) for k in kap], grads, hess)
    opt = aoo.NewtonStep(verbose=0)
    for g in range(0, G, 7):
        new_s, low_s = opt.damped_newton_step(objs[g].energy_from_parameters, (theta0, kap[g]),
                                              grads[g], hess[g])
        assert abs(low_b[g].item() - low_s) < 1e-13
        e_b = objs[g].energy_from_parameters(*new_b[g]).item()
        e_s = objs[g].energy_from_parameters(*new_s).item()
        assert abs(e_b - e_s) < 1e-10
        assert (new_b[g][1] - new_s[1]).abs().max() < 1e-10


def _batch_of(N, G, seed0=20262, freeze_active=False, nelec=16):
    from auto_oo_amd.synthetic import synthetic_problem
    pqc = aoo.Parameterized_circuit(3, 4, None, ansatz="ucc")
    mols, coeffs, objs, probs = [], [], [], []
    for g in range(G):
        P = synthetic_problem(N, seed0 + 1000 * g)
        mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
        mols.append(mol)
        coeffs.append(P["oao_mo_coeff"])
        probs.append(P)
        objs.append(aoo.OO_pqc(pqc, mol, 3, 4, oao_mo_coeff=P["oao_mo_coeff"], freeze_active=freeze_active))
    batch = aoo.OO_pqc_batch(pqc, mols, 3, 4, oao_mo_coeffs=coeffs, freeze_active=freeze_active)
    return pqc, batch, objs, probs


def oracle_small_hessians(N, G, freeze):
    """The oracle's energies, full gradients and full Hessians of the G synthetic geometries of
    test_batched_full_hessian_vs_oracle_small (seeds 20262 + 1000 g, theta from default_rng(5)): up to two minutes of
    host CPU, kept under tests/golden/oracle_cache/."""
    def compute():
        rng = np.random.default_rng(5)
        thetas = torch.tensor(rng.uniform(0, 2 * np.pi, (G, 4)))
        en, gr, he = [], [], []
        for g in range(G):
            P = R.synthetic_problem(N, 20262 + 1000 * g)
            omol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 16)
            ooo = R.OracleOOPQC(R.OraclePQC(3, 4, "ucc"), omol, 3, 4, P["oao_mo_coeff"], freeze_active=freeze)
            en.append(ooo.energy_from_parameters(thetas[g]).item())
            gr.append(ooo.full_gradient(thetas[g]).numpy())
            he.append(pack_sym(ooo.full_hessian(thetas[g]).numpy()))
        return {"energy": np.array(en), "gradient": np.stack(gr), "hessian_triu": np.stack(he)}
    return oracle_values(f"small_hessians_n{N}_g{G}_{'frozen' if freeze else 'free'}", compute)


@pytest.mark.parametrize("N,G,freeze", [(13, 3, False), (13, 2, True), (20, 5, False)])
def test_batched_full_hessian_vs_oracle_small(N, G, freeze):
    """OO_pqc_batch.energy_gradient_hessian (ONE call for all geometries, oovqe_oo_hessian_batch)
    against the oracle's full_gradient / full_hessian (oo_pqc.py:132-148), geometry by geometry, with
    a different theta per geometry."""
    pqc, batch, objs, probs = _batch_of(N, G, freeze_active=freeze)
    rng = np.random.default_rng(5)
    thetas = torch.tensor(rng.uniform(0, 2 * np.pi, (G, pqc.theta_shape)))
    E, grad, H = batch.energy_gradient_hessian(thetas.cuda())
    n = batch.n_theta + batch.n_kappa
    assert E.shape == (G,) and grad.shape == (G, n) and H.shape == (G, n, n)
    ref = oracle_small_hessians(N, G, freeze)
    for g in range(G):
        assert abs(E[g].item() - float(ref["energy"][g])) < 1e-9
        assert (grad[g].cpu() - torch.tensor(ref["gradient"][g])).abs().max() < 1e-8
        assert (H[g].cpu() - torch.tensor(unpack_sym(ref["hessian_triu"][g]))).abs().max() < 1e-8


def oracle_hessian_n43_rng9():
    """The oracle's full Hessian of geometry 0 (seed 20262, N = 43) at the first theta of default_rng(9).uniform --
    minutes of host CPU, kept under tests/golden/oracle_cache/."""
    def compute():
        theta = torch.tensor(np.random.default_rng(9).uniform(0, 2 * np.pi, (6, 4)))[0]
        P = R.synthetic_problem(43, 20262)
        omol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 16)
        ooo = R.OracleOOPQC(R.OraclePQC(3, 4, "ucc"), omol, 3, 4, P["oao_mo_coeff"])
        return {"theta": theta.numpy(), "hessian_triu": pack_sym(ooo.full_hessian(theta).numpy()),
                "gradient": ooo.full_gradient(theta).numpy()}
    return oracle_values("hessian_n43_s20262_rng9", compute)


def test_batched_full_hessian_equals_single_geometry_path_at_cc_pvdz_shape():
    """configs[3] shape (N = 43, 331 x 331): the batched call against OO_pqc.full_hessian /
    full_gradient of each geometry (itself pinned to the oracle by
    test_config3_unit_of_work_at_cc_pvdz_shape), all three blocks, and geometry 0 against the oracle."""
    pqc, batch, objs, probs = _batch_of(43, 6)
    rng = np.random.default_rng(9)
    thetas = torch.tensor(rng.uniform(0, 2 * np.pi, (6, pqc.theta_shape))).cuda()
    E, grad, H = batch.energy_gradient_hessian(thetas)
    nt = batch.n_theta
    for g, oo in enumerate(objs):
        e1, g1 = oo.energy_and_gradient(thetas[g])
        h1 = oo.full_hessian(thetas[g])
        assert abs(E[g].item() - e1.item()) < 1e-11
        assert (grad[g] - g1).abs().max() < 1e-10
        assert (H[g][:nt, :nt] - h1[:nt, :nt]).abs().max() < 1e-10
        assert (H[g][nt:, :nt] - h1[nt:, :nt]).abs().max() < 1e-10
        assert (H[g][nt:, nt:] - h1[nt:, nt:]).abs().max() < 1e-9
        assert torch.equal(H[g], H[g].T) or (H[g] - H[g].T).abs().max() < 1e-10
    ref = oracle_hessian_n43_rng9()
    assert (H[0].cpu() - torch.tensor(unpack_sym(ref["hessian_triu"]))).abs().max() < 1e-8
    # the two convenience views
    assert torch.equal(batch.full_hessian(thetas), H)
    assert (batch.full_gradient(thetas) - grad).abs().max() < 1e-12


@pytest.mark.parametrize("N,G,nelec", [(13, 3, 16), (20, 2, 16), (43, 4, 16), (48, 2, 16),
                                       (20, 3, 28), (24, 3, 4), (43, 2, 30), (17, 3, 10)])
def test_batched_hessian_quarter_transform_from_stage1_equals_its_own_pass(N, G, nelec):
    """The K-type quarter transform of the orbital Hessian leaves stage 1 together with the J-type one
    (half_transform_kernel's Vk output + t2k_tri_kernel, hessian.hip) when the integrals are p <-> q
    symmetric; option hess_vk_pass = 1 brings back the round-2 form (its own pass over the whole AO
    tensor through K1).  Same sums in the same order: equal to rounding, for odd and even N, one, two
    and three column tiles, and 3 ... 16 occupied + active orbitals (nelec: the accumulator count of
    t2k_tri_kernel is a template argument)."""
    from auto_oo_amd import _lib
    pqc, batch, objs, probs = _batch_of(N, G, nelec=nelec)
    rng = np.random.default_rng(11)
    thetas = torch.tensor(rng.uniform(0, 2 * np.pi, (G, pqc.theta_shape))).cuda()
    E, grad, H = batch.energy_gradient_hessian(thetas)
    with _lib.debug_options(hess_vk_pass=1):
        E0, grad0, H0 = batch.energy_gradient_hessian(thetas)
    assert torch.equal(E, E0) and torch.equal(grad, grad0)
    scale = H0.abs().max().item()
    assert (H - H0).abs().max().item() <= 1e-12 * scale
    assert torch.isfinite(H).all()
    # the evaluation inside the call takes its J from the orbital Hessian's T2 (one stage 1 for both); option
    # hess_own_stage1 = 1: it streams the (packed) integrals itself, as energy_and_gradient does
    with _lib.debug_options(hess_own_stage1=1):
        E1, grad1, H1 = batch.energy_gradient_hessian(thetas)
    assert (E - E1).abs().max().item() < 1e-11
    assert (grad - grad1).abs().max().item() < 1e-11
    assert (H - H1).abs().max().item() <= 1e-12 * scale
    EG = batch.energy_and_gradient(thetas)
    assert (EG[:, 0] - E1).abs().max().item() < 1e-12 and (EG[:, 1:] - grad1).abs().max().item() < 1e-12


def test_transition_rdms_small_register_kernel_vs_oracle():
    """oovqe_rdms with different bra and ket on a register that fits LDS (rdms_small_kernel: one
    workgroup per pair) against the oracle's transition RDMs; 6 and 8 qubits, several pairs per call."""
    for ncas, nelec in ((3, 4), (4, 4)):
        D = 1 << (2 * ncas)
        rng = np.random.default_rng(ncas)
        bra = torch.tensor(rng.standard_normal((5, D)))
        ket = torch.tensor(rng.standard_normal((5, D)))
        g1, g2 = ops.rdms(bra.cuda(), ket.cuda(), ncas)
        E = R.RdmOperators(ncas).E                     # scipy sparse E_pq (active_space.py:29-53)
        for b in range(5):
            x, y = bra[b].numpy(), ket[b].numpy()
            for p_ in range(ncas):
                for q_ in range(ncas):
                    assert abs(g1[b, p_, q_].item() - x @ (E[p_][q_] @ y)) < 1e-11
                    for r_ in range(ncas):
                        for s_ in range(ncas):
                            ref = x @ (E[p_][q_] @ (E[r_][s_] @ y))
                            if q_ == r_:
                                ref -= x @ (E[p_][s_] @ y)
                            assert abs(g2[b, p_, q_, r_, s_].item() - ref) < 1e-11


def test_batched_rotation_and_energy_at_kappa():
    """OO_pqc_batch.energy(thetas, kappas) == OO_pqc.energy_from_parameters(theta, kappa) per geometry
    (one launch rotates all geometries); rotate_() == the orbital update of oo_pqc.py:191."""
    pqc, batch, objs, probs = _batch_of(43, 4)
    rng = np.random.default_rng(3)
    thetas = torch.tensor(rng.uniform(0, 2 * np.pi, (4, pqc.theta_shape))).cuda()
    kappas = torch.tensor(rng.normal(0, 0.05, (4, batch.n_kappa))).cuda()
    e_b = batch.energy(thetas, kappas)
    for g, oo in enumerate(objs):
        e_s = oo.energy_from_parameters(thetas[g], kappas[g])
        assert abs(e_b[g].item() - e_s.item()) < 1e-11
    before = batch.oao_mo_coeff.clone()
    batch.rotate_(kappas)
    for g, oo in enumerate(objs):
        U = oo.kappa_to_mo_coeff(kappas[g])
        assert (batch.oao_mo_coeff[g] - before[g] @ U).abs().max() < 1e-13
        assert (batch.mo_coeff[g] - batch.oao_coeff[g] @ batch.oao_mo_coeff[g]).abs().max() < 1e-13
    assert (batch.energy(thetas) - e_b).abs().max() < 1e-11


def test_batched_newton_step_equals_per_geometry_steps():
    """configs[3] end to end on the batched device path: OO_pqc_batch.damped_newton_step (one
    gradient+Hessian call, one direction launch, batched line-search trials) against NewtonStep on
    the OO_pqc object of each geometry, and geometry 0 against the oracle's step (1e-9 Ha)."""
    pqc, batch, objs, probs = _batch_of(43, 8)
    theta0 = torch.full((8, pqc.theta_shape), 0.1, dtype=torch.float64, device="cuda")
    c_before = batch.oao_mo_coeff.clone()
    new_t, e_new, low = batch.damped_newton_step(theta0)
    # the step adopted the accepted trial's orbitals (buffers exchanged, nothing recomputed): they are the
    # orbitals the returned energies were evaluated at, and mo_coeff = S^-1/2 C_oao holds
    assert (batch.energy(new_t) - e_new).abs().max().item() == 0.0
    for g in range(8):
        assert (batch.mo_coeff[g] - batch.oao_coeff[g] @ batch.oao_mo_coeff[g]).abs().max() < 1e-13
    opt = aoo.NewtonStep(verbose=0)
    for g, oo in enumerate(objs):
        kappa = torch.zeros(oo.n_kappa, dtype=torch.float64, device="cuda")
        new, low_s = opt.damped_newton_step(oo.energy_from_parameters, (theta0[g], kappa),
                                            oo.full_gradient(theta0[g]), oo.full_hessian(theta0[g]))
        e_s = oo.energy_from_parameters(new[0], new[1]).item()
        assert (batch.oao_mo_coeff[g] - c_before[g] @ oo.kappa_to_mo_coeff(new[1])).abs().max() < 1e-9
        assert abs(low[g].item() - low_s) < 1e-10
        assert abs(e_new[g].item() - e_s) < 1e-10
        assert (new_t[g] - new[0]).abs().max() < 1e-9
        assert e_new[g].item() < oo.energy_from_parameters(theta0[g]).item()
    ref = oracle_config3_n43()          # (geometry 0 at theta = 0.1 is that problem: seed 20262)
    assert abs(low[0].item() - float(ref["lowest_eigenvalue"])) < 1e-9
    assert abs(e_new[0].item() - float(ref["new_energy"])) < 1e-9


def test_batched_newton_step_that_gives_up_keeps_parameters_and_orbitals():
    """newton_raphson.py:177-183 in lockstep: with an Armijo constant no step can meet and lmax = 1 every problem
    gives up, the old parameters come back and the orbitals of the batch are the old ones (the trial's orbitals
    are NOT adopted)."""
    pqc, batch, objs, probs = _batch_of(43, 4)
    theta0 = torch.full((4, pqc.theta_shape), 0.1, dtype=torch.float64, device="cuda")
    c_before, m_before = batch.oao_mo_coeff.clone(), batch.mo_coeff.clone()
    e_before = batch.energy(theta0)
    opt = aoo.BatchedNewtonStep(verbose=0, alpha=50.0, lmax=1)
    new_t, e_new, low = batch.damped_newton_step(theta0, opt)
    assert opt.last_search_gave_up
    assert torch.equal(new_t, theta0)
    assert torch.equal(batch.oao_mo_coeff, c_before)
    assert (batch.mo_coeff - m_before).abs().max() < 1e-14
    assert (e_new - e_before).abs().max() < 1e-11
    # and a step with the usual constants afterwards goes down from there
    new_t, e_new, low = batch.damped_newton_step(theta0)
    assert (e_new < e_before).all()


def test_one_call_newton_step_equals_the_step_driven_call_by_call():
    """oovqe_oo_newton_step_batch (the whole lockstep step up to the line search's first verdict enqueued by one
    call) against the same step driven through the single entry points (``step_by_calls``): the same launches, so
    the same bits -- thetas, energies, orbitals at every step of a short optimisation that starts with
    indefinite Hessians (band route, backtracking) and ends with positive definite ones, where the one-call form
    stops waiting for the band route (``speculate``).  A speculation that turns out wrong (forced here) is
    repaired: the band route's directions then come from the side stream with another workgroup count, so
    rounding-level agreement."""
    N, G, steps = 20, 3, 40
    runs = []
    for by_calls in (False, True):
        pqc, batch, objs, probs = _batch_of(N, G, freeze_active=True)
        batch.step_by_calls = by_calls
        th = torch.full((G, pqc.theta_shape), 0.1, dtype=torch.float64, device="cuda")
        traj, spec = [], []
        for it in range(steps):
            spec.append(batch._all_pd_last_step)
            th, e, low = batch.damped_newton_step(th, defer_lowest=True)
            traj.append((th.clone(), e.clone(), batch.oao_mo_coeff.clone(), batch.mo_coeff.clone(),
                         low.result().clone()))
            if not by_calls and sum(spec) >= 3:
                steps = it + 1                  # (three speculative steps seen: the other run makes as many)
                break
        runs.append((traj, spec))
    (tb, spec_b), (ta, _) = runs
    for it, (a, b) in enumerate(zip(ta, tb)):
        for x, y in zip(a[:4], b[:4]):
            assert torch.equal(x, y), f"step {it}"
        assert (a[4] - b[4]).abs().max() < 1e-10, f"lowest eigenvalues, step {it}"
    assert (ta[-1][1] < ta[0][1]).all()
    # the optimisation reached positive definite Hessians and the one-call form speculated there
    assert any(spec_b), "no step ran with speculate = 1 (the trajectory never became positive definite)"
    # a wrong speculation: indefinite Hessians (the start point) with the flag forced on
    pqc, batch, objs, probs = _batch_of(N, G, freeze_active=True)
    th0 = torch.full((G, pqc.theta_shape), 0.1, dtype=torch.float64, device="cuda")
    batch._all_pd_last_step = True
    th1, e1, low1 = batch.damped_newton_step(th0)
    assert batch._all_pd_last_step is False
    assert (th1 - ta[0][0]).abs().max() < 1e-9 and (e1 - ta[0][1]).abs().max() < 1e-9
    assert (batch.oao_mo_coeff - ta[0][2]).abs().max() < 1e-9
    assert (low1 - ta[0][4]).abs().max() < 1e-9


@pytest.mark.parametrize("N,freeze", [(13, False), (43, True), (52, False)])
def test_full_hessian_one_call_equals_the_three_blocks(N, freeze):
    """OO_pqc.full_hessian (oo_pqc.py:136-148) as ONE library call (the batched entry point on a stack of one
    geometry, on the object's own tensors) against the three calls + concatenations it replaces
    (``hessian_by_blocks``): the same launches, the same bits; beyond N = 48 the three calls stay."""
    ooo, opqc, oo, pqc = _setup(N, 20261 + N, freeze_active=freeze)
    theta = torch.tensor(np.random.default_rng(N).uniform(0, 2 * np.pi, pqc.theta_shape))
    H1 = oo.full_hessian(theta)
    assert ("_hess1_plans" in oo.__dict__) == (N <= 48)          # (beyond N = 48 the three calls stay)
    oo.hessian_by_blocks = True
    H3 = oo.full_hessian(theta)
    assert H1.shape == H3.shape and torch.equal(H1, H3)
    oo.hessian_by_blocks = False
    assert torch.equal(oo.full_hessian(theta.cuda()), H1)
    if N == 13:
        # integrals without the p <-> q / r <-> s symmetries (an in-place edit: the flags are re-verified): the
        # general-tensor kernels behind both forms
        oo.int2e_ao[0, 1, 2, 3] += 0.25
        assert oo._eri_flags() == 0
        Hg1 = oo.full_hessian(theta)
        oo.hessian_by_blocks = True
        assert torch.equal(oo.full_hessian(theta), Hg1) and not torch.equal(Hg1, H1)


def test_one_call_newton_step_beyond_the_cholesky_kernel():
    """N = 64: n_theta + n_kappa = 520 is beyond the one-workgroup Cholesky (495), so every direction of the
    one-call step comes from the band route on the calling stream (no fast path, nothing on the side stream) --
    the same launches as the step driven call by call, the same bits; and the tile-packed integrals (N > 48)
    with the rotation's workspace."""
    N, G = 64, 2
    outs = []
    for by_calls in (True, False):
        pqc, batch, objs, probs = _batch_of(N, G, seed0=515)
        assert batch.n_theta + batch.n_kappa > aoo._lib.load().oovqe_newton_direction_pd_max_n()
        batch.step_by_calls = by_calls
        th = torch.full((G, pqc.theta_shape), 0.1, dtype=torch.float64, device="cuda")
        e0 = batch.energy(th)
        new_t, e_new, low = batch.damped_newton_step(th)
        assert (e_new < e0).all()
        outs.append((new_t.clone(), e_new.clone(), low.clone(), batch.oao_mo_coeff.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("N,G", [(13, 3), (43, 8), (43, 1)])
def test_hessian_call_chains_on_internal_streams_equal_the_one_stream_call(N, G):
    """oovqe_oo_hessian_batch runs the chains of its graph that do not depend on each other beside each other (the
    evaluation, the J-type chain and the circuit block on the caller's stream, the K-type chain + assembly on the
    library's internal stream, forked and joined inside the call) -- against the same call with every launch on
    the caller's stream (option ``one_stream``): the same launches, the same bits; repeated calls reuse the
    workspace while the previous call's chains have been joined."""
    from auto_oo_amd import _lib
    pqc, batch, objs, probs = _batch_of(N, G, seed0=811, freeze_active=(N == 43))
    rng = np.random.default_rng(N + G)
    sets = [torch.tensor(rng.uniform(0, 2 * np.pi, (G, pqc.theta_shape)), device="cuda") for _ in range(3)]
    with _lib.debug_options(one_stream=1):
        ref = [tuple(t.clone() for t in batch.energy_gradient_hessian(th)) for th in sets]
    got = [tuple(t.clone() for t in batch.energy_gradient_hessian(th)) for th in sets]
    for r, g_ in zip(ref, got):
        for a, b in zip(r, g_):
            assert torch.equal(a, b)
    assert not torch.equal(ref[0][2], ref[1][2])
    H = got[0][2]
    assert (H - H.transpose(1, 2)).abs().max() < 1e-9


def test_one_call_newton_step_with_symmetric_integrals_and_more_than_16_orbitals():
    """N <= 48 with n_occ + ncas > 16: the integrals carry both symmetry flags but no packed copy exists for that
    shape (the packed-triangle kernels end at M = 16) -- the one-call step must hand the library a null packed
    pointer, not fail on it; same bits as the step driven call by call."""
    N, G = 24, 2
    outs = []
    for by_calls in (True, False):
        pqc, batch, objs, probs = _batch_of(N, G, seed0=733, nelec=32)
        assert batch.eri_flags == 3 and batch._eri_packed is None and batch._n_occ + batch.ncas == 17
        batch.step_by_calls = by_calls
        th = torch.full((G, pqc.theta_shape), 0.1, dtype=torch.float64, device="cuda")
        e0 = batch.energy(th)
        new_t, e_new, low = batch.damped_newton_step(th)
        assert (e_new < e0).all()
        outs.append((new_t.clone(), e_new.clone(), low.clone(), batch.oao_mo_coeff.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # the same through OO_pqc.full_optimization's one-geometry stack
    oo = objs[0]
    e_l, th_l, *_ = oo.full_optimization(torch.full((pqc.theta_shape,), 0.1, dtype=torch.float64), max_iterations=3,
                                         verbose=None)
    assert e_l[-1] < e_l[0]


def test_missing_side_stream_eigenvalue_is_recomputed_not_raised():
    """ops.PendingLowest: a NaN eigenvalue from the side route (a hand-off that timed out) is computed again on
    the calling stream with one workgroup per problem instead of aborting the caller -- forced here by
    overwriting the delivered values."""
    rng = np.random.default_rng(3)
    n, G = 58, 4
    H = torch.tensor(_pd_stack(rng, n, G), device="cuda")
    g = torch.tensor(rng.standard_normal((G, n)), device="cuda")
    dp, low, nu = ops.newton_direction(H, g, defer_lowest=True)
    want = low.result().clone()
    low._tensor[1] = float("nan")
    vals = low.tolist()
    assert max(abs(a - b) for a, b in zip(vals, want.tolist())) < 1e-10
    ev = np.linalg.eigvalsh(H.cpu().numpy())[:, 0]
    assert np.abs(np.array(vals) - ev).max() < 1e-9


def _pd_stack(rng, n, G, low=0.05):
    out = []
    for k in range(G):
        A = rng.standard_normal((n, n))
        Q, _ = np.linalg.qr(A)
        ev = np.sort(rng.uniform(low, 4.0, n))
        ev[0] = low * (1.0 + 0.1 * k)
        out.append((Q * ev) @ Q.T)
    S = np.stack(out)
    return 0.5 * (S + S.transpose(0, 2, 1))


@pytest.mark.parametrize("n,G", [(331, 1), (331, 8), (331, 64), (100, 5), (33, 7), (17, 3), (16, 2), (15, 2), (2, 3),
                                 (1, 1), (479, 2), (495, 2)])
def test_cholesky_fast_path_vs_band_route_and_numpy(n, G):
    """Positive definite Hessians (the reference does not shift: newton_raphson.py:107-128): the direction of
    the Cholesky fast path (newton_chol.hip) against numpy's solve and against the band route (debug option
    newton_no_chol), info == 1, no shift; the lowest eigenvalue -- band route on the side stream -- equals the
    one-route value bit for bit."""
    rng = np.random.default_rng(17 * n + G)
    Hn = _pd_stack(rng, n, G)
    gn = rng.standard_normal((G, n))
    Hs, gs = torch.tensor(Hn).cuda(), torch.tensor(gn).cuda()
    dp, low, nu, info = ops.newton_direction(Hs, gs, want_info=True)
    assert info.tolist() == [1.0] * G and float(nu.abs().max()) == 0.0
    with aoo._lib.debug_options(newton_no_chol=1):
        dpb, lowb, nub, infob = ops.newton_direction(Hs, gs, want_info=True)
    assert infob.tolist() == [0.0] * G
    # (the eigenvalue route beside the line search runs on fewer workgroups per problem than the one-route
    # call: another grid of shifts, the same number to rounding)
    assert (low - lowb).abs().max().item() < 1e-12 * max(n, 8)
    for k in range(G):
        ref = -np.linalg.solve(Hn[k], gn[k])
        cond = np.linalg.cond(Hn[k])
        tol = 1e-14 * cond * (1.0 + np.abs(ref).max()) * max(n, 8)
        assert np.abs(dp[k].cpu().numpy() - ref).max() <= tol, (k, np.abs(dp[k].cpu().numpy() - ref).max(), tol)
        assert (dp[k] - dpb[k]).abs().max().item() <= 10 * tol
        assert abs(low[k].item() - np.linalg.eigvalsh(Hn[k])[0]) < 1e-12 * n
    # one problem alone and the deferred form give the same bits as the batch
    d0, l0, _ = ops.newton_direction(Hs[0], gs[0], defer_lowest=True)
    assert torch.equal(d0, dp[0]) and abs(l0.result().item() - low[0].item()) < 1e-12 * n


def test_fast_path_decision_walks_across_lambda_min():
    """The fast path's verdict is the reference's branch (newton_raphson.py:107): lowest eigenvalue above
    lambda_min -> plain -H^-1 g from the Cholesky factor; below it (still positive, or negative) -> level
    shift through the band route.  Walking the lowest eigenvalue across lambda_min, both sides of the branch
    reproduce the oracle's step, in one batch."""
    rng = np.random.default_rng(5)
    n, lam_min = 120, 1e-6
    lows = [1.0, 1e-3, 1e-5, 2e-6, 1.2e-6, 1.01e-6, 0.99e-6, 0.8e-6, 1e-7, 0.0, -1e-7, -0.3]
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    Hn = []
    for lo in lows:
        ev = np.linspace(0.4, 2.0, n)
        ev[0] = lo
        S = (Q * ev) @ Q.T
        Hn.append(0.5 * (S + S.T))
    Hn = np.stack(Hn)
    gn = rng.standard_normal((len(lows), n))
    dp, low, nu, info = ops.newton_direction(torch.tensor(Hn).cuda(), torch.tensor(gn).cuda(), want_info=True)
    for k, lo in enumerate(lows):
        # (the constructed eigenvalue is only exact to ~1e-16 * |H|: keep a margin around the threshold)
        if lo > lam_min * 1.005:
            assert info[k].item() == 1.0 and nu[k].item() == 0.0
        if lo < lam_min * 0.995:
            assert info[k].item() == 0.0
            assert abs(nu[k].item() - (1e-6 + 1.1 * abs(low[k].item()))) < 1e-18 + 1e-12 * abs(nu[k].item())
        dr, lr = _reference_direction(torch.tensor(Hn[k]), torch.tensor(gn[k]))
        assert abs(low[k].item() - lr) < 1e-13
        shift = nu[k].item()
        Hs = Hn[k] + shift * np.eye(n)
        cond = np.linalg.cond(Hs)
        assert np.abs(dp[k].cpu().numpy() - dr.numpy()).max() <= 1e-13 * cond * (1.0 + np.abs(dr.numpy()).max()) * n


@pytest.mark.parametrize("n", [500, 700])
def test_indefinite_hessian_without_level_shift_beyond_the_pivoted_kernel(n):
    """aug = False inverts an indefinite Hessian as it stands (newton_raphson.py:107: the shift is skipped).
    Beyond the one-workgroup kernel with its pivoted solve (n > 480) the band LDL^T has no pivoting, so the
    library refuses loudly (info = -2, dp = NaN, lowest eigenvalue valid) and NewtonStep falls back to eigh;
    a positive definite Hessian still goes through."""
    rng = np.random.default_rng(n)
    H = torch.tensor(_sym(rng, n, "indefinite"))
    g = torch.tensor(rng.standard_normal(n))
    dp, low, nu, info = ops.newton_direction(H.cuda(), g.cuda(), aug=False, want_info=True)
    assert info.item() == -2.0 and bool(torch.isnan(dp).all())
    dr, lr = _reference_direction(H, g, aug=False)
    assert abs(low.item() - lr) < 1e-11 * n
    opt = aoo.NewtonStep(aug=False, verbose=0)
    dpn, lown = opt.newton_step(g.cuda(), H.cuda())
    cond = float(torch.linalg.cond(H))
    assert (dpn.cpu() - dr).abs().max() <= 1e-13 * cond * (1.0 + dr.abs().max()) * n
    assert abs(lown - lr) < 1e-11 * n
    Hp = torch.tensor(_sym(rng, n, "pd"))
    dpp, lowp, _, infop = ops.newton_direction(Hp.cuda(), g.cuda(), aug=False, want_info=True)
    assert infop.item() == 0.0
    drp, lrp = _reference_direction(Hp, g, aug=False)
    assert (dpp.cpu() - drp).abs().max() < 1e-9 * (1 + drp.abs().max()) and abs(lowp.item() - lrp) < 1e-11


def test_non_finite_hessian_raises_instead_of_returning_nan_parameters():
    """A Hessian with an Inf (or NaN) in it must not come back as NaN parameters: the damped Newton step
    raises, single and batched."""
    rng = np.random.default_rng(3)
    n = 40
    H = torch.tensor(_sym(rng, n, "pd"))
    g = torch.tensor(rng.standard_normal(n)).cuda()
    x0 = torch.zeros(n, dtype=torch.float64, device="cuda")
    Hg = H.cuda()
    fn = lambda x: 0.5 * x @ (Hg @ x) + g @ x                      # noqa: E731
    for bad in (float("inf"), float("nan")):
        Hb = H.clone()
        Hb[3, 2] = Hb[2, 3] = bad
        with pytest.raises((aoo._lib.OovqeError, AssertionError)):
            aoo.NewtonStep(verbose=0).damped_newton_step(fn, (x0,), g, Hb.cuda())
        with pytest.raises((aoo._lib.OovqeError, AssertionError)):
            aoo.BatchedNewtonStep(verbose=0).damped_newton_steps_flat(
                lambda pts: torch.stack([fn(p) for p in pts]), x0[None].repeat(2, 1), g[None].repeat(2, 1),
                torch.stack((H, Hb)).cuda())
    # and a NaN trial energy is never accepted: the step backtracks and finally keeps the old parameters
    calls = []

    def nan_objective(x):
        calls.append(1)
        return fn(x) if float(x.abs().max()) == 0.0 else torch.tensor(float("nan"), dtype=torch.float64, device="cuda")
    new, _ = aoo.NewtonStep(verbose=0, lmax=3).damped_newton_step(nan_objective, (x0,), g, Hg)
    assert float(new.abs().max()) == 0.0 and len(calls) >= 4


def test_band_route_beside_a_kernel_that_holds_the_chip():
    """The workgroups of a problem in the band route wait for each other, so they must all become resident.
    Launched while a long matrix-core kernel on ANOTHER stream holds every CU (an N = 160 four-index transform:
    four launches of ~4 ms), the call must still deliver the right step: either its workgroups get their CUs
    in time, or the hand-off times out loudly (info = -1, NaN outputs) and NewtonStep repeats the direction
    with one workgroup per problem.  Never a silent wrong answer."""
    rng = np.random.default_rng(11)
    n, G = 331, 4
    Hs = torch.tensor(np.stack([_sym(rng, n, "indefinite") for _ in range(G)])).cuda()
    gs = torch.tensor(rng.standard_normal((G, n))).cuda()
    N = 160
    big = torch.rand((N, N, N, N), dtype=torch.float64, device="cuda")
    C = torch.rand((N, N), dtype=torch.float64, device="cuda")
    out, work = torch.empty_like(big), torch.empty_like(big)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(2):
            ops.general_4index_transform(big, C, C, C, C, out=out, work=work)
    dp, low, nu, info = ops.newton_direction(Hs, gs, want_info=True)
    opt = aoo.BatchedNewtonStep(verbose=0)
    with torch.cuda.stream(side):
        ops.general_4index_transform(big, C, C, C, C, out=out, work=work)
    dp2, low2 = opt.newton_steps(gs, Hs)
    torch.cuda.synchronize()
    codes = info.tolist()
    assert all(c in (0.0, -1.0) for c in codes), codes
    for k in range(G):
        dr, lr = _reference_direction(Hs[k].cpu(), gs[k].cpu())
        cond = float(torch.linalg.cond(Hs[k] + (1e-6 + 1.1 * abs(lr)) * torch.eye(n, dtype=torch.float64, device="cuda")))
        tol = 1e-13 * cond * (1.0 + dr.abs().max()) * n
        if codes[k] == 0.0:
            assert abs(low[k].item() - lr) < 1e-11 * n and (dp[k].cpu() - dr).abs().max() <= tol
        else:
            assert bool(torch.isnan(dp[k]).all())
        assert abs(low2[k].item() - lr) < 1e-11 * n and (dp2[k].cpu() - dr).abs().max() <= tol
