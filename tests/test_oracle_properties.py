"""Properties the reference asserts on real molecules (test/test_oo_energy.py:114-185,415-971;
test/test_oo_pqc.py:38-148), replayed on synthetic 8-fold-symmetric integrals because PySCF is
absent: transform == naive einsum, analytic gradient/Hessian == autodiff, blocks == joint autodiff."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref as R


@pytest.fixture(scope="module")
def small():
    P = R.synthetic_problem(13, 20261)
    mol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 16)
    pqc = R.OraclePQC(3, 4, "ucc")
    oo = R.OracleOOPQC(pqc, mol, 3, 4, P["oao_mo_coeff"])
    theta = torch.tensor(np.random.default_rng(5).uniform(0, 2 * np.pi, 4))
    return oo, pqc, theta


def test_transform_is_naive_einsum(small):
    oo, _, _ = small
    C = oo.mo_coeff
    g = R.int2e_transform(oo.int2e_ao, C)
    gn = torch.einsum("pi,qj,rk,sl,pqrs->ijkl", C, C, C, C, oo.int2e_ao)
    assert (g - gn).abs().max() < 1e-12


def test_analytic_gradient_is_autodiff(small):
    oo, pqc, theta = small
    g1, g2 = pqc.get_rdms(theta)
    kappa = torch.zeros(oo.n_kappa, dtype=torch.float64)
    ga = torch.func.jacrev(oo.energy_from_kappa)(kappa, g1, g2)
    assert (oo.orbital_gradient(theta) - ga).abs().max() < 1e-11


def test_analytic_hessian_is_autodiff(small):
    oo, pqc, theta = small
    g1, g2 = pqc.get_rdms(theta)
    kappa = torch.zeros(oo.n_kappa, dtype=torch.float64)
    Ha = torch.func.hessian(oo.energy_from_kappa)(kappa, g1, g2)
    assert (oo.orbital_orbital_hessian(theta) - Ha).abs().max() < 1e-10


def test_blocks_are_joint_autodiff(small):
    oo, _, theta = small
    kappa = torch.zeros(oo.n_kappa, dtype=torch.float64)
    J = torch.autograd.functional.jacobian(oo.energy_from_parameters, (theta, kappa))
    assert (J[0] - oo.circuit_gradient(theta)).abs().max() < 1e-11
    assert (J[1] - oo.orbital_gradient(theta)).abs().max() < 1e-11
