"""Real-molecule anchors (VERDICT r1 weak #1 / missing #6).

The reference's tests hold literals for formaldimine / STO-3G that need AO integrals
(test/test_moldata_pyscf.py:17-85, test/test_oo_energy.py:27-95,241-298,318-396,416-473,
test/test_oo_pqc.py:38-84; transcribed to tests/golden/molecule_cases.json by make_goldens.py).
auto_oo_amd/gaussian.py computes those integrals from scratch; here the literals pin

  * the integral generator itself: S^-1/2 (9 digits) and the RHF energy -92.66372193556138;
  * the ORACLE's integral-side half on a real molecule: energy_from_mo_coeff at the literal
    orbitals and RDMs == -92.74923236954386 (at the reference's own tolerance: its orbital literal
    has 5 digits), analytic gradient / Hessian == autodiff at the literal orbitals,
    block derivatives == joint autodiff for the np_fabric literal.

CPU only; the HIP path takes the same fixtures in tests/test_molecule_goldens_gpu.py."""
import json
import os

import numpy as np
import pytest
import torch

from auto_oo_amd.gaussian import Moldata_sto3g, zmatrix_to_cartesian
from auto_oo_amd.moldata import ao_to_oao, get_formal_geo
from oracle import cpu_ref as R

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "molecule_cases.json")) as fh:
    CASES = json.load(fh)

_MOLS = {}


def T(x):
    return torch.tensor(x, dtype=torch.float64)


def molecule(case):
    a, p = case["geometry"]["formal_geo"]
    if (a, p) not in _MOLS:
        _MOLS[(a, p)] = Moldata_sto3g(get_formal_geo(a, p))
    return _MOLS[(a, p)]


def by_test(name):
    return [c for c in CASES if c["test"] == name]


def oracle_mol(mol):
    return R.OracleMol(mol.int1e_ao, mol.int2e_ao, mol.overlap, mol.nuc, mol.nelectron)


def test_zmatrix_geometry():
    sym, xyz = zmatrix_to_cartesian(get_formal_geo(140, 80))
    assert sym == ["N", "C", "H", "H", "H"]
    d = lambda i, j: np.linalg.norm(xyz[i] - xyz[j])      # noqa: E731
    assert abs(d(0, 1) - 1.498047) < 1e-12 and abs(d(1, 2) - 1.066797) < 1e-12
    assert abs(d(1, 3) - 1.066797) < 1e-12 and abs(d(0, 4) - 0.987109) < 1e-12
    ang = lambda i, j, k: np.degrees(np.arccos(np.dot(xyz[i] - xyz[j], xyz[k] - xyz[j])    # noqa: E731
                                               / d(i, j) / d(k, j)))
    assert abs(ang(2, 1, 0) - 118.359375) < 1e-10 and abs(ang(4, 0, 1) - 140.0) < 1e-10
    assert np.allclose(xyz[0], 0) and np.allclose(xyz[1, 1:], 0)


@pytest.mark.parametrize("case", by_test("test_ao_to_oao"), ids=lambda c: c["source"])
def test_overlap_inverse_square_root_literal(case):
    """test/test_moldata_pyscf.py:86-90: ao_to_oao(int1e_ovlp) == the 13 x 13 literal."""
    mol = molecule(case)
    ref = np.array(case["oao_coeff_ref"])
    assert np.allclose(ao_to_oao(mol.overlap), ref)            # the reference's assertion
    assert np.abs(mol.oao_coeff - ref).max() < 2e-8              # 9-digit literal
    # ao_to_oao((R R)^-1) == R for the literal R itself
    assert np.abs(ao_to_oao(np.linalg.inv(ref @ ref)) - ref).max() < 1e-10
    assert np.abs(np.diag(mol.overlap) - 1).max() < 1e-12 and np.abs(mol.overlap - mol.overlap.T).max() == 0
    g = mol.int2e_ao
    for perm in ((1, 0, 2, 3), (0, 1, 3, 2), (2, 3, 0, 1)):
        assert np.array_equal(g, g.transpose(perm))               # exact 8-fold symmetry


@pytest.mark.parametrize("case", by_test("test_mo_ao_to_oao"), ids=lambda c: c["source"])
def test_rhf_orbitals_literal(case):
    """test/test_oo_energy.py:98-103: S^1/2 C_HF == literal (columns up to sign), C^T C = 1."""
    mol = molecule(case)
    mol.run_rhf()
    assert abs(mol.hf.e_tot - (-92.66372193556138)) < 1e-9      # test/test_oo_energy.py:396
    ref = np.array(case["hf_oao_coeff_ref"])
    assert np.abs(ref.T @ ref - np.eye(13)).max() < 1e-7
    from auto_oo_amd.oo_energy import mo_ao_to_mo_oao
    assert np.allclose(mo_ao_to_mo_oao(mol.oao_coeff, mol.overlap), np.eye(13))
    mine = mo_ao_to_mo_oao(mol.hf.mo_coeff, mol.overlap)
    sign = np.sign(np.sum(mine * ref, axis=0))
    # the literal comes from an SCF stopped at PySCF's default tolerance (1e-9 in the energy, ~1e-5
    # in the orbitals); ours is converged to 1e-12
    assert np.abs(mine * sign - ref).max() < 2e-5


@pytest.mark.parametrize("case", by_test("test_energy_from_mo_coeff"), ids=lambda c: c["source"])
def test_oracle_energy_from_mo_coeff_literal(case):
    """test/test_oo_energy.py:299-310: the integral transform + CAS coefficients + energy chain of
    the oracle on the real molecule against the reference-held number."""
    mol = molecule(case)
    oo = R.OracleOOEnergy(oracle_mol(mol), case["ncas"], case["nelecas"], np.eye(13),
                          freeze_active=case["freeze_active"])
    e = oo.energy_from_mo_coeff(T(case["mo_coeff"]), T(case["one_rdm"]),
                                T(case["two_rdm"]))
    assert np.allclose(e.item(), case["e_ref"])                  # the reference's assertion
    assert abs(e.item() - case["e_ref"][0]) < 2e-4               # what 5-digit orbitals allow


@pytest.mark.parametrize("case", by_test("test_orbital_optimization"), ids=lambda c: c["source"])
def test_oracle_energy_at_rhf_orbitals_is_the_rhf_energy(case):
    """test/test_oo_energy.py:397-404: orbital optimisation with the Hartree-Fock RDMs from the RHF
    orbitals ends at e_ref = E_RHF; the oracle's energy there is E_RHF already and its analytic
    orbital gradient vanishes (the optimisation has nothing left to do)."""
    mol = molecule(case)
    mol.run_rhf()
    from auto_oo_amd.oo_energy import mo_ao_to_mo_oao
    oo = R.OracleOOEnergy(oracle_mol(mol), case["ncas"], case["nelecas"],
                          mo_ao_to_mo_oao(mol.hf.mo_coeff, mol.overlap), freeze_active=case["freeze_active"])
    g1, g2 = T(case["one_rdm"]), T(case["two_rdm"])
    e = oo.energy_from_mo_coeff(oo.mo_coeff, g1, g2).item()
    assert abs(e - case["e_ref"][0]) < 1e-9
    grad = oo.kappa_matrix_to_vector(oo.analytic_gradient(g1, g2))
    assert grad.abs().max() < 1e-6
    # the literal orbital matrix of this case is S^-1/2 (OAO basis): it reproduces our S^-1/2
    assert np.abs(np.array(case["mo_coeff"]) - mol.oao_coeff).max() < 2e-8


@pytest.mark.parametrize("case", by_test("test_analytical_derivatives"), ids=lambda c: c["source"])
def test_oracle_analytic_derivatives_on_the_real_molecule(case):
    """test/test_oo_energy.py:925-943 with the oracle: jacobian / hessian of energy_from_kappa at
    kappa = 0 == analytic gradient / Hessian (same assertions, PySCF integrals replaced by ours)."""
    mol = molecule(case)
    mol.run_rhf()
    from auto_oo_amd.oo_energy import mo_ao_to_mo_oao
    oo = R.OracleOOEnergy(oracle_mol(mol), case["ncas"], case["nelecas"],
                          mo_ao_to_mo_oao(mol.hf.mo_coeff, mol.overlap), freeze_active=case["freeze_active"])
    g1, g2 = T(case["one_rdm"]), T(case["two_rdm"])
    zero = torch.zeros(oo.n_kappa, dtype=torch.float64)
    ga = torch.autograd.functional.jacobian(lambda k: oo.energy_from_kappa(k, g1, g2), zero)
    ge = oo.kappa_matrix_to_vector(oo.analytic_gradient(g1, g2))
    assert torch.allclose(ga, ge) and (ga - ge).abs().max() < 1e-9
    if case["check_hess"]:
        ha = torch.autograd.functional.hessian(lambda k: oo.energy_from_kappa(k, g1, g2), zero)
        he = oo.full_hessian_to_matrix(oo.analytic_hessian(g1, g2))
        assert torch.allclose(ha, he) and (ha - he).abs().max() < 1e-8
