"""Row N1: torch as the autodiff container of the cost functions, on the HIP kernels.

GPU replays of the reference's own autodiff checks -- test/test_oo_pqc.py:101-125
(jacobian / hessian of energy_from_parameters(theta, kappa) at kappa = 0 against the analytic
blocks) and test/test_oo_energy.py:930-943 (jacrev / hessian of energy_from_kappa) -- on synthetic
integrals, plus first derivatives at kappa != 0 against the CPU oracle's autograd.  1e-8 abs."""
import numpy as np
import pytest
import torch
from torch.autograd.functional import hessian as thessian, jacobian as tjacobian

pytestmark = pytest.mark.gpu

import auto_oo_amd as aoo            # noqa: E402
from oracle import cpu_ref as R      # noqa: E402
from tests.test_api_gpu import _setup   # noqa: E402

TOL = 1e-8


@pytest.mark.parametrize("ansatz,N,freeze", [("ucc", 13, False), ("ucc", 13, True), ("np_fabric", 13, True),
                                              ("ucc", 43, False)])
def test_jacobian_and_hessian_of_energy_from_parameters(ansatz, N, freeze):
    """test/test_oo_pqc.py:101-125 with CPU tensors in, as the reference's test passes them."""
    if ansatz == "np_fabric":
        P = R.synthetic_problem(N, 20261)
        mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 16)
        pqc = aoo.Parameterized_circuit(3, 4, None, ansatz="np_fabric", n_layers=2)
        oo = aoo.OO_pqc(pqc, mol, 3, 4, oao_mo_coeff=P["oao_mo_coeff"], freeze_active=freeze)
    else:
        _, _, oo, pqc = _setup(N, 20261 if N == 13 else 20262, freeze_active=freeze)
    rng = np.random.default_rng(11)
    theta = torch.tensor(rng.uniform(0, 2 * np.pi, pqc.theta_shape))
    kappa = torch.zeros(oo.n_kappa, dtype=torch.float64)
    grad_auto = tjacobian(oo.energy_from_parameters, (theta, kappa))
    assert not grad_auto[0].is_cuda and grad_auto[0].shape == theta.shape
    assert (grad_auto[0] - oo.circuit_gradient(theta).cpu()).abs().max() < TOL
    assert (grad_auto[1] - oo.orbital_gradient(theta).cpu()).abs().max() < TOL
    # circuit_gradient is reachable as jacobian(energy_from_parameters, theta)  (oo_pqc.py:86-95)
    gc = tjacobian(oo.energy_from_parameters, theta)
    assert (gc - oo.circuit_gradient(theta).cpu()).abs().max() < TOL
    if N > 13:
        return
    hess_auto = thessian(oo.energy_from_parameters, (theta, kappa))
    assert (hess_auto[0][0] - oo.circuit_circuit_hessian(theta).cpu()).abs().max() < TOL
    assert (hess_auto[1][0] - oo.orbital_circuit_hessian(theta).cpu()).abs().max() < TOL
    assert (hess_auto[0][1] - oo.orbital_circuit_hessian(theta).cpu().T).abs().max() < TOL
    assert (hess_auto[1][1] - oo.orbital_orbital_hessian(theta).cpu()).abs().max() < TOL


@pytest.mark.parametrize("N,seed", [(13, 20261), (20, 20263)])
def test_first_derivatives_at_nonzero_kappa_vs_oracle_autograd(N, seed):
    """dE/dtheta and dE/dkappa at kappa != 0: the chain rule through expm(-K) (Frechet derivative)
    against torch autograd through the CPU oracle's matrix_exp + einsum transforms."""
    ooo, opqc, oo, pqc = _setup(N, seed)
    rng = np.random.default_rng(3)
    theta = torch.tensor(rng.uniform(0, 2 * np.pi, pqc.theta_shape))
    kappa = torch.tensor(rng.normal(0, 0.05, oo.n_kappa))
    ga = tjacobian(oo.energy_from_parameters, (theta, kappa))
    gr = tjacobian(ooo.energy_from_parameters, (theta, kappa))
    assert (ga[0] - gr[0]).abs().max() < TOL
    assert (ga[1] - gr[1]).abs().max() < TOL
    # plain backward() with device tensors
    th = theta.cuda().requires_grad_(True)
    ka = kappa.cuda().requires_grad_(True)
    E = oo.energy_from_parameters(th, ka)
    assert abs(E.item() - ooo.energy_from_parameters(theta, kappa).item()) < 1e-9
    E.backward()
    assert (th.grad.cpu() - gr[0]).abs().max() < TOL and (ka.grad.cpu() - gr[1]).abs().max() < TOL
    # second derivatives away from kappa = 0 (oo_pqc.py:103-125 works at any point): all four blocks
    # against autograd through the oracle (1e-7: round 3)
    if N == 13:
        ha = thessian(oo.energy_from_parameters, (theta, kappa))
        hr = thessian(ooo.energy_from_parameters, (theta, kappa))
        for i in range(2):
            for j in range(2):
                assert ha[i][j].shape == hr[i][j].shape
                assert (ha[i][j] - hr[i][j]).abs().max() < 1e-7, (i, j)


@pytest.mark.parametrize("freeze", [False, True])
def test_energy_from_kappa_jacrev_and_hessian(freeze):
    """test/test_oo_energy.py:930-943: jacrev(energy_from_kappa, argnums=0) == analytic gradient,
    hessian(energy_from_kappa, argnums=0) == full_hessian_to_matrix(analytic_hessian)."""
    ooo, opqc, oo, pqc = _setup(13, 20261, freeze_active=freeze)
    theta = torch.tensor(np.random.default_rng(4).uniform(0, 2 * np.pi, pqc.theta_shape))
    one_rdm, two_rdm = opqc.get_rdms(theta)
    zero = torch.zeros(oo.n_kappa, dtype=torch.float64)
    g_auto = torch.func.jacrev(oo.energy_from_kappa, argnums=0)(zero, one_rdm, two_rdm)
    g_exact = oo.kappa_matrix_to_vector(oo.analytic_gradient(one_rdm, two_rdm))
    assert (g_auto - g_exact.cpu()).abs().max() < TOL
    h_auto = torch.func.hessian(oo.energy_from_kappa, argnums=0)(zero, one_rdm, two_rdm)
    h_exact = oo.full_hessian_to_matrix(oo.analytic_hessian(one_rdm, two_rdm))
    assert (h_auto - h_exact.cpu()).abs().max() < TOL
    h_auto2 = thessian(lambda k: oo.energy_from_kappa(k, one_rdm, two_rdm), zero)
    assert (h_auto2 - h_exact.cpu()).abs().max() < TOL
    # kappa != 0 and the RDM arguments (E is linear in them: dE/dgamma = c1, dE/dGamma = c2)
    kappa = torch.tensor(np.random.default_rng(5).normal(0, 0.05, oo.n_kappa))
    ja = tjacobian(oo.energy_from_kappa, (kappa, one_rdm, two_rdm))
    jr = tjacobian(ooo.energy_from_kappa, (kappa, one_rdm, two_rdm))
    for a, r in zip(ja, jr):
        assert (a - r).abs().max() < TOL
    # the kappa-kappa Hessian at kappa != 0: the analytic Hessian over all rotation pairs at the rotated
    # orbitals pulled back through the first and second Frechet derivatives of expm
    kappa2 = torch.tensor(np.random.default_rng(6).normal(0, 0.1, oo.n_kappa))
    ha = thessian(lambda k: oo.energy_from_kappa(k, one_rdm, two_rdm), kappa2)
    hr = thessian(lambda k: ooo.energy_from_kappa(k, one_rdm, two_rdm), kappa2)
    assert (ha - hr).abs().max() < 1e-7
    assert (ha - ha.T).abs().max() < 1e-9
    # energy_from_mo_coeff with respect to the orbital matrix itself
    C = ooo.mo_coeff.clone()
    wa = tjacobian(lambda c: oo.energy_from_mo_coeff(c, one_rdm, two_rdm), C)
    wr = tjacobian(lambda c: ooo.energy_from_mo_coeff(c, one_rdm, two_rdm), C)
    assert (wa - wr).abs().max() < TOL


def test_rdms_are_differentiable_in_theta():
    """get_rdms(theta) under autograd: Jacobians = derivative RDMs (what backprop through the
    simulator yields in the reference, oo_pqc.py:113-119)."""
    ooo, opqc, oo, pqc = _setup(13, 20261)
    theta = torch.tensor(np.random.default_rng(6).uniform(0, 2 * np.pi, pqc.theta_shape))
    ja = tjacobian(pqc.get_rdms, theta)
    jr = tjacobian(opqc.get_rdms, theta)
    assert (ja[0] - jr[0]).abs().max() < TOL and (ja[1] - jr[1]).abs().max() < TOL
    # and composed with the energy: d/dtheta of E(kappa=0; RDMs(theta)) = circuit gradient
    g = tjacobian(lambda t: oo.energy_from_kappa(torch.zeros(oo.n_kappa, dtype=torch.float64),
                                                 *pqc.get_rdms(t)), theta)
    assert (g - oo.circuit_gradient(theta).cpu()).abs().max() < TOL


def test_no_autodiff_no_graph():
    """Plain calls stay plain: no grad_fn, same numbers as under autodiff."""
    _, _, oo, pqc = _setup(13, 20261)
    theta = torch.tensor(np.random.default_rng(7).uniform(0, 2 * np.pi, pqc.theta_shape))
    E = oo.energy_from_parameters(theta)
    assert E.grad_fn is None and not E.requires_grad
    E2 = oo.energy_from_parameters(theta.clone().requires_grad_(True))
    assert E2.grad_fn is not None and E2.item() == E.item()
