"""The HIP engine on the real molecule: the reference's STO-3G literals and its recorded notebook
runs (tests/golden/molecule_cases.json, notebook_runs.json; integrals from auto_oo_amd/gaussian.py)
through the drop-in API -- OO_energy / OO_pqc / NewtonStep / bogoliubov_atob_cas -- the way the
reference's own tests and notebooks call it."""
import numpy as np
import pytest
import torch
from torch.autograd.functional import hessian as thessian, jacobian as tjacobian

pytestmark = pytest.mark.gpu

import auto_oo_amd as aoo                          # noqa: E402
from auto_oo_amd.berry import bogoliubov_atob_cas, state_overlap   # noqa: E402
from oracle import cpu_ref as R                    # noqa: E402
from tests import _replay as P                     # noqa: E402
from tests.test_molecule_goldens import by_test, molecule, oracle_mol, T   # noqa: E402


@pytest.mark.parametrize("case", by_test("test_energy_from_mo_coeff"), ids=lambda c: c["source"])
def test_energy_from_mo_coeff_literal(case):
    """test/test_oo_energy.py:299-310 as written there (the object starts from RHF orbitals)."""
    mol = molecule(case)
    oo = aoo.OO_energy(mol, case["ncas"], case["nelecas"], freeze_active=case["freeze_active"])
    e = oo.energy_from_mo_coeff(T(case["mo_coeff"]), T(case["one_rdm"]), T(case["two_rdm"]))
    assert np.allclose(e.item(), case["e_ref"])
    ooo = R.OracleOOEnergy(oracle_mol(mol), case["ncas"], case["nelecas"], np.eye(13),
                           freeze_active=case["freeze_active"])
    e_o = ooo.energy_from_mo_coeff(T(case["mo_coeff"]), T(case["one_rdm"]), T(case["two_rdm"]))
    assert abs(e.item() - e_o.item()) < 1e-9


@pytest.mark.parametrize("case", by_test("test_orbital_optimization"), ids=lambda c: c["source"])
def test_orbital_optimization_reaches_the_rhf_energy(case):
    """test/test_oo_energy.py:397-404 verbatim: orbital_optimization(one_rdm, two_rdm)[-1] == e_ref."""
    mol = molecule(case)
    oo = aoo.OO_energy(mol, case["ncas"], case["nelecas"], freeze_active=case["freeze_active"])
    energy_l = oo.orbital_optimization(T(case["one_rdm"]), T(case["two_rdm"]), verbose=None)
    assert np.allclose(case["e_ref"], energy_l[-1])
    assert abs(energy_l[-1] - case["e_ref"][0]) < 1e-9


@pytest.mark.parametrize("case", by_test("test_analytical_derivatives"), ids=lambda c: c["source"])
def test_analytical_derivatives_on_the_real_molecule(case):
    """test/test_oo_energy.py:925-943 verbatim (torch.func.jacrev / hessian through the kernels)."""
    mol = molecule(case)
    oo = aoo.OO_energy(mol, case["ncas"], case["nelecas"], freeze_active=case["freeze_active"])
    g1, g2 = T(case["one_rdm"]), T(case["two_rdm"])
    zero = torch.zeros(oo.n_kappa, dtype=torch.float64)
    ga = torch.func.jacrev(oo.energy_from_kappa, argnums=0)(zero, g1, g2)
    ge = oo.kappa_matrix_to_vector(oo.analytic_gradient(g1, g2)).cpu()
    assert torch.allclose(ga, ge) and (ga - ge).abs().max() < 1e-9
    ha = torch.func.hessian(oo.energy_from_kappa, argnums=0)(zero, g1, g2)
    he = oo.full_hessian_to_matrix(oo.analytic_hessian(g1, g2)).cpu()
    assert torch.allclose(ha, he) and (ha - he).abs().max() < 1e-8
    # and against the oracle's autodiff of its einsum chain on the same integrals
    mol.run_rhf()
    ooo = R.OracleOOEnergy(oracle_mol(mol), case["ncas"], case["nelecas"],
                           aoo.mo_ao_to_mo_oao(mol.hf.mo_coeff, mol.overlap),
                           freeze_active=case["freeze_active"])
    go = tjacobian(lambda k: ooo.energy_from_kappa(k, g1, g2), zero)
    assert (ga - go).abs().max() < 1e-8


@pytest.mark.parametrize("case", by_test("test_full_derivatives"), ids=lambda c: c["source"])
def test_full_derivatives_literal(case):
    """test/test_oo_pqc.py:85-125 verbatim: np_fabric, the literal OAO orbitals and theta; joint
    autodiff of energy_from_parameters == the analytic blocks."""
    mol = molecule(case)
    pqc = aoo.Parameterized_circuit(case["ncas"], case["nelecas"], None, ansatz="np_fabric",
                                    n_layers=case["n_layers"])
    oo = aoo.OO_pqc(pqc, mol, case["ncas"], case["nelecas"], oao_mo_coeff=T(case["oao_mo_coeff"]),
                    freeze_active=case["freeze_active"])
    theta = T(case["theta"])
    kappa = torch.zeros(oo.n_kappa, dtype=torch.float64)
    grad_auto = tjacobian(oo.energy_from_parameters, (theta, kappa))
    assert torch.allclose(grad_auto[0], oo.circuit_gradient(theta).cpu())
    assert torch.allclose(grad_auto[1], oo.orbital_gradient(theta).cpu())
    hess_auto = thessian(oo.energy_from_parameters, (theta, kappa))
    assert torch.allclose(hess_auto[0][0], oo.circuit_circuit_hessian(theta).cpu())
    assert torch.allclose(hess_auto[1][0], oo.orbital_circuit_hessian(theta).cpu())
    assert torch.allclose(hess_auto[1][1], oo.orbital_orbital_hessian(theta).cpu())
    # the oracle on the same literal inputs
    omol = oracle_mol(mol)
    ooo = R.OracleOOPQC(R.OraclePQC(case["ncas"], case["nelecas"], "np_fabric", n_layers=case["n_layers"]),
                        omol, case["ncas"], case["nelecas"], np.array(case["oao_mo_coeff"]),
                        freeze_active=case["freeze_active"])
    assert abs(oo.energy_from_parameters(theta).item() - ooo.energy_from_parameters(theta).item()) < 1e-9
    J = tjacobian(ooo.energy_from_parameters, (theta, kappa))
    assert (grad_auto[0] - J[0]).abs().max() < 1e-8 and (grad_auto[1] - J[1]).abs().max() < 1e-8


def _engine(run):
    pqc = aoo.Parameterized_circuit(run["ncas"], run["nelecas"], None, ansatz=run["ansatz"],
                                    n_layers=run["n_layers"])

    def make(mol, oao_mo_coeff):
        return aoo.OO_pqc(pqc, mol, run["ncas"], run["nelecas"], oao_mo_coeff=oao_mo_coeff,
                          freeze_active=run["freeze_active"])
    return pqc, make


def test_tutorial_oo_vqe_trajectory():
    """examples/Tutorial_auto_oo.ipynb cell 52: the recorded 19 Newton iterations of
    OO_pqc.full_optimization (CAS(4e,3o), np_fabric 2 layers) down to -92.74995368139427."""
    run = P.RUNS["tutorial_auto_oo"]
    pqc, make = _engine(run)
    oo = make(P.sto3g_molecule(*run["formal_geo"]), P.reference_hf_orbitals())
    energy_l, theta_l, kappa_l, coeff_l, eig_l = oo.full_optimization(pqc.init_zeros(), verbose=None)
    ref = run["energies"]
    assert len(energy_l) == len(ref) - 1
    assert np.abs(np.array(energy_l) - np.array(ref[1:])).max() < 5e-7
    assert np.abs(np.array(energy_l[-2:]) - np.array(ref[-2:])).max() < 1e-9
    assert abs(energy_l[-1] - run["E_fin"]) < 1e-9
    assert abs(energy_l[-1] - run["printed_hf_casci_casscf"][2]) < 1e-6      # the CASSCF energy
    # from our own (tighter) RHF orbitals the path differs in the 6th digit, the minimum does not
    mol = P.sto3g_molecule(*run["formal_geo"])
    oo2 = aoo.OO_pqc(pqc, mol, run["ncas"], run["nelecas"], freeze_active=True)    # runs RHF itself
    e0 = oo2.energy_from_parameters(pqc.init_zeros()).item()
    assert abs(e0 - ref[0]) < 1e-9 and abs(e0 - run["printed_hf_casci_casscf"][0]) < 1e-6
    e2 = oo2.full_optimization(pqc.init_zeros(), verbose=None)[0]
    assert abs(e2[-1] - run["E_fin"]) < 1e-8


def test_tutorial_berry_phase_loop():
    """examples/Tutorial_Berry_phase.ipynb cells 17-32 on the engine: pre-optimisation, one damped
    Newton step per loop point (the configs[3] workload on the real molecule), the state overlaps
    under the active-space rotation, and the sign flip of the closed loop."""
    run = P.RUNS["tutorial_berry_phase"]
    pqc, make = _engine(run)
    out = P.berry_loop(make, aoo.NewtonStep(verbose=0), run, torch.device("cuda"))
    assert abs(out["preopt"][0] - run["preopt_energies"][0]) < 1e-9
    assert abs(out["preopt"][-1] - run["preopt_E_fin"]) < 1e-9
    assert abs(out["lowest"] - run["preopt_lowest_hessian_eigenvalue"]) < 1e-6
    assert np.abs(np.array(out["energies"]) - np.array(run["loop_energies"])).max() < 1e-8
    states = [pqc.qnode(t) for t in out["thetas"]]
    n = len(states)
    ovl = []
    for i in range(n):
        j = (i + 1) % n
        mo_atob = out["orbitals"][i].T @ out["orbitals"][j]
        rot = bogoliubov_atob_cas(mo_atob, out["act_idx"], run["nelecas"])
        ovl.append(state_overlap(states[j], rot, states[i]).real.item())
    assert np.abs(np.array(ovl[:-1]) - np.array(run["overlaps"])).max() < 5e-7
    assert abs(ovl[-1] - run["final_overlap"]) < 5e-7 and ovl[-1] < -0.99
