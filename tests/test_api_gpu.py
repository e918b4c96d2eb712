"""GPU parity of the drop-in API (OO_energy / OO_pqc / Parameterized_circuit) against the CPU
oracle on identical synthetic inputs.  Energies to 1e-9 Ha, gradients /
Hessian blocks to 1e-8 abs, as stated in the north star."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import auto_oo_amd as aoo            # noqa: E402
from oracle import cpu_ref as R      # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    with open(os.path.join(HERE, "golden", name)) as fh:
        return json.load(fh)


def _setup(N, seed, ncas=3, nelecas=4, nelec=16, freeze_active=False, ansatz="ucc", k=1):
    P = R.synthetic_problem(N, seed)
    omol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    opqc = R.OraclePQC(ncas, nelecas, "ucc" if ansatz == "ucc" else "kupccd", k=k)
    ooo = R.OracleOOPQC(opqc, omol, ncas, nelecas, P["oao_mo_coeff"], freeze_active=freeze_active)
    mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz=ansatz, k=k)
    oo = aoo.OO_pqc(pqc, mol, ncas, nelecas, oao_mo_coeff=P["oao_mo_coeff"],
                    freeze_active=freeze_active)
    return ooo, opqc, oo, pqc


@pytest.mark.parametrize("case", _load("pqc_states.json"), ids=lambda c: c["source"])
def test_qnode_golden(case):
    pqc = aoo.Parameterized_circuit(case["ncas"], case["nelecas"], None, ansatz=case["ansatz"],
                                    n_layers=case["n_layers"] or 3,
                                    add_singles=bool(case["add_singles"]))
    state = pqc.qnode(torch.tensor(case["theta"], dtype=torch.float64))
    assert state.dtype == torch.complex128 and state.shape == (2 ** (2 * case["ncas"]),)
    ref = np.array(case["state_real"]) + 1j * np.array(case["state_imag"])
    assert np.allclose(state.cpu().numpy(), ref, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("case", _load("pqc_rdms.json"), ids=lambda c: c["source"])
def test_get_rdms_golden(case):
    pqc = aoo.Parameterized_circuit(case["ncas"], case["nelecas"], None, ansatz=case["ansatz"],
                                    n_layers=case["n_layers"] or 3,
                                    add_singles=bool(case["add_singles"]))
    theta = torch.tensor(case["theta"], dtype=torch.float64)
    g1, g2 = pqc.get_rdms(theta)
    assert np.allclose(g1.cpu().numpy(), np.array(case["one_rdm"]), rtol=1e-5, atol=1e-8)
    assert np.allclose(g2.cpu().numpy(), np.array(case["two_rdm"]), rtol=1e-5, atol=1e-8)
    s1, s2 = pqc.get_rdms_from_state(pqc.qnode(theta))
    assert (s1 - g1).abs().max() < 1e-13 and (s2 - g2).abs().max() < 1e-13


@pytest.mark.parametrize("case", _load("skew_pack.json")[:1], ids=lambda c: c["source"])
def test_skew_pack(case):
    v = torch.tensor(case["vector"], dtype=torch.float64, device="cuda")
    m = aoo.vector_to_skew_symmetric(v)
    assert np.array_equal(m.cpu().numpy(), np.array(case["matrix"]))
    assert np.array_equal(aoo.skew_symmetric_to_vector(m).cpu().numpy(), np.array(case["vector"]))


@pytest.mark.parametrize("N,seed,freeze", [(13, 20261, False), (13, 20261, True), (43, 20262, False)])
def test_energy_and_gradients(N, seed, freeze):
    ooo, opqc, oo, pqc = _setup(N, seed, freeze_active=freeze)
    assert oo.n_kappa == ooo.n_kappa and np.array_equal(oo.params_idx, ooo.params_idx)
    rng = np.random.default_rng(5)
    theta = torch.tensor(rng.uniform(0, 2 * np.pi, pqc.theta_shape))
    kappa = torch.tensor(rng.normal(0, 0.05, oo.n_kappa))
    # energies
    assert abs(oo.energy_from_parameters(theta).item() - ooo.energy_from_parameters(theta).item()) < 1e-9
    assert abs(oo.energy_from_parameters(theta, kappa).item()
               - ooo.energy_from_parameters(theta, kappa).item()) < 1e-9
    # gradients
    gc_ref = ooo.circuit_gradient(theta)
    go_ref = ooo.orbital_gradient(theta)
    assert (oo.circuit_gradient(theta).cpu() - gc_ref).abs().max() < 1e-8
    assert (oo.orbital_gradient(theta).cpu() - go_ref).abs().max() < 1e-8
    fg = oo.full_gradient(theta).cpu()
    assert (fg - torch.cat((gc_ref, go_ref))).abs().max() < 1e-8
    E, g = oo.energy_and_gradient(theta)
    assert abs(E.item() - ooo.energy_from_parameters(theta).item()) < 1e-9
    assert torch.equal(g.cpu(), fg)
    # mixed Hessian block
    if N <= 13:
        hoc_ref = ooo.orbital_circuit_hessian(theta)
        assert (oo.orbital_circuit_hessian(theta).cpu() - hoc_ref).abs().max() < 1e-8


def test_oo_energy_pieces():
    ooo, opqc, oo, pqc = _setup(13, 20261)
    rng = np.random.default_rng(9)
    theta = torch.tensor(rng.uniform(0, 2 * np.pi, pqc.theta_shape))
    kappa = torch.tensor(rng.normal(0, 0.05, oo.n_kappa))
    g1, g2 = opqc.get_rdms(theta)
    C = ooo.mo_coeff
    assert (oo.mo_coeff.cpu() - C).abs().max() < 1e-13
    c0, c1, c2 = oo.get_active_integrals(oo.mo_coeff)
    r0, r1, r2 = ooo.get_active_integrals(C)
    assert abs(c0.item() - float(r0)) < 1e-10
    assert (c1.cpu() - r1).abs().max() < 1e-11 and (c2.cpu() - r2).abs().max() < 1e-11
    assert abs(oo.energy_from_kappa(kappa, g1, g2).item()
               - ooo.energy_from_kappa(kappa, g1, g2).item()) < 1e-9
    assert torch.equal(oo.kappa_vector_to_matrix(kappa).cpu(), ooo.kappa_vector_to_matrix(kappa))
    K = oo.kappa_vector_to_matrix(kappa)
    assert torch.equal(oo.kappa_matrix_to_vector(K).cpu(), kappa)
    assert (oo.kappa_to_mo_coeff(kappa).cpu() - ooo.kappa_to_mo_coeff(kappa)).abs().max() < 1e-10
    G = oo.analytic_gradient(g1, g2).cpu()
    assert (G - ooo.analytic_gradient(g1, g2)).abs().max() < 1e-10
    # API taking full MO integrals
    h_mo = aoo.int1e_transform(ooo.int1e_ao, C)
    g_mo = aoo.int2e_transform(ooo.int2e_ao, C)
    assert (h_mo.cpu() - R.int1e_transform(ooo.int1e_ao, C)).abs().max() < 1e-12
    assert (g_mo.cpu() - R.int2e_transform(ooo.int2e_ao, C)).abs().max() < 1e-11
    F = oo.fock_generalized(h_mo, g_mo, g1, g2).cpu()
    Fr = ooo.fock_generalized(R.int1e_transform(ooo.int1e_ao, C), R.int2e_transform(ooo.int2e_ao, C),
                              g1, g2)
    assert (F - Fr).abs().max() < 1e-10


def test_errors_match_reference_behaviour():
    P = R.synthetic_problem(8, 3)
    mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 7)
    with pytest.raises(ValueError, match="odd number of core electrons"):
        mol.get_active_space_idx(2, 2)
    with pytest.raises(ValueError):
        aoo.Parameterized_circuit(2, 2, None, ansatz="ucc").qnode(torch.zeros(3, dtype=torch.float64))


@pytest.mark.parametrize("mode", ["default", "no_ride", "t3x1", "t2"])
@pytest.mark.parametrize("N,G", [(13, 5), (43, 3)])
def test_batched_evaluation_matches_single(N, G, mode, lib_options):
    """OO_pqc_batch (one call for G geometries) == OO_pqc per geometry == oracle, for every launch
    structure of the evaluation: circuit workgroups riding along the K1 launch or launched on their
    own (option no_ride), persistent T3 kernel or T2 kernels (library test hooks)."""
    if mode == "no_ride":
        lib_options(no_ride=1)
    elif mode == "t3x1":
        lib_options(fused_chunks=1)
    elif mode == "t2":
        lib_options(cas_unfused=1)
    from auto_oo_amd.synthetic import synthetic_problem
    ncas, nelecas, nelec = 3, 4, 16
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="ucc")
    mols, coeffs, singles = [], [], []
    for g in range(G):
        P = synthetic_problem(N, 500 + g)
        mols.append(aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"] + g, nelec))
        coeffs.append(P["oao_mo_coeff"])
        singles.append(aoo.OO_pqc(pqc, mols[-1], ncas, nelecas, oao_mo_coeff=P["oao_mo_coeff"]))
    batch = aoo.OO_pqc_batch(pqc, mols, ncas, nelecas, oao_mo_coeffs=coeffs)
    rng = np.random.default_rng(3)
    thetas = torch.tensor(rng.uniform(0, 2 * np.pi, (G, pqc.theta_shape)))
    eg = batch.energy_and_gradient(thetas).cpu()
    en = batch.energy(thetas).cpu()
    for g in range(G):
        E, grad = singles[g].energy_and_gradient(thetas[g])
        assert abs(eg[g, 0].item() - E.item()) < 1e-11
        assert (eg[g, 1:] - grad.cpu()).abs().max() < 1e-11
        assert abs(en[g].item() - E.item()) < 1e-11
    # oracle for geometry 1
    P = synthetic_problem(N, 501)
    omol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"] + 1, nelec)
    ooo = R.OracleOOPQC(R.OraclePQC(ncas, nelecas, "ucc"), omol, ncas, nelecas, P["oao_mo_coeff"])
    assert abs(eg[1, 0].item() - ooo.energy_from_parameters(thetas[1]).item()) < 1e-9
    assert (eg[1, 1:] - ooo.full_gradient(thetas[1])).abs().max() < 1e-8


@pytest.mark.parametrize("N,seed,freeze", [(13, 20261, False), (13, 20261, True)])
def test_hessian_blocks(N, seed, freeze):
    """Every Hessian block of OO_pqc against the oracle (autodiff through the simulator for the
    circuit blocks, dense N^6 einsums for the orbital block): <= 1e-8 abs."""
    ooo, opqc, oo, pqc = _setup(N, seed, freeze_active=freeze)
    rng = np.random.default_rng(8)
    theta = torch.tensor(rng.uniform(0, 2 * np.pi, pqc.theta_shape))
    g1, g2 = opqc.get_rdms(theta)
    h_oo = oo.orbital_orbital_hessian(theta).cpu()
    assert (h_oo - ooo.orbital_orbital_hessian(theta)).abs().max() < 1e-8
    h_cc = oo.circuit_circuit_hessian(theta).cpu()
    assert (h_cc - ooo.circuit_circuit_hessian(theta)).abs().max() < 1e-8
    h_oc = oo.orbital_circuit_hessian(theta).cpu()
    assert (h_oc - ooo.orbital_circuit_hessian(theta)).abs().max() < 1e-8
    H = oo.full_hessian(theta).cpu()
    assert (H - ooo.full_hessian(theta)).abs().max() < 1e-8
    assert (H - H.T).abs().max() < 1e-8
    # full-tensor API + helpers
    Hfull = oo.analytic_hessian(g1, g2)
    assert (Hfull.cpu() - ooo.analytic_hessian(g1, g2)).abs().max() < 1e-8
    assert (oo.full_hessian_to_matrix(Hfull).cpu() - h_oo).abs().max() < 1e-9
    C = ooo.mo_coeff
    h_mo, g_mo = R.int1e_transform(ooo.int1e_ao, C), R.int2e_transform(ooo.int2e_ao, C)
    Hfi = oo.analytic_hessian_from_integrals(h_mo, g_mo, g1, g2)
    assert (Hfi.cpu() - ooo.analytic_hessian_from_integrals(h_mo, g_mo, g1, g2)).abs().max() < 1e-8
    f1, f2 = oo.full_rdms(g1, g2)
    r1, r2 = ooo.full_rdms(g1, g2)
    assert torch.equal(f1.cpu(), r1) and torch.equal(f2.cpu(), r2)
    assert (oo.fock_core(h_mo, g_mo).cpu() - ooo.fock_core(h_mo, g_mo)).abs().max() < 1e-11
    assert (oo.fock_active(g_mo, g1).cpu() - ooo.fock_active(g_mo, g1)).abs().max() < 1e-11
    assert (oo.y_matrix(g_mo, r2).cpu() - ooo.y_matrix(g_mo, r2)).abs().max() < 1e-9


def test_orbital_hessian_cc_pvdz_shape():
    ooo, opqc, oo, pqc = _setup(43, 20262)
    theta = torch.tensor(np.random.default_rng(8).uniform(0, 2 * np.pi, pqc.theta_shape))
    h_oo = oo.orbital_orbital_hessian(theta).cpu()
    ref = ooo.orbital_orbital_hessian(theta)
    assert h_oo.shape == (327, 327)
    assert (h_oo - ref).abs().max() < 1e-8


def test_full_optimization_converges_like_oracle_newton():
    """OO-VQE Newton optimisation (oo_pqc.py:155-207) on the synthetic STO-3G-shaped problem, run
    to the reference's stopping rule (|dE| < conv_tol, oo_pqc.py:200-201): monotone energy, a
    stationary point (max |gradient| < 1e-6, positive lowest Hessian eigenvalue at the end), and
    the first Newton energy equal to the oracle's first step."""
    ooo, opqc, oo, pqc = _setup(13, 20261, freeze_active=True)
    theta0 = torch.zeros(pqc.theta_shape, dtype=torch.float64)
    conv_tol = 1e-11
    energy_l, theta_l, kappa_l, coeff_l, eig_l = oo.full_optimization(
        theta0, max_iterations=400, conv_tol=conv_tol, verbose=None)
    assert len(energy_l) < 400 and abs(energy_l[-1] - energy_l[-2]) < conv_tol     # converged, not cut off
    assert all(b <= a + 1e-10 for a, b in zip(energy_l, energy_l[1:]))
    g = oo.full_gradient(theta_l[-1])
    assert g.abs().max().item() < 1e-6
    assert eig_l[-1] > 0.0                                   # a minimum: no augmentation at the end
    assert len(theta_l) == len(kappa_l) == len(coeff_l) == len(eig_l) == len(energy_l)
    assert kappa_l[-1] is theta_l[-1]                        # the reference's quirk (oo_pqc.py:189)
    # oracle: same first damped Newton step from the same start
    opt = R.OracleNewtonStep()
    kappa0 = torch.zeros(ooo.n_kappa, dtype=torch.float64)
    new, _ = opt.damped_newton_step(ooo.energy_from_parameters, (theta0, kappa0),
                                    ooo.full_gradient(theta0), ooo.full_hessian(theta0))
    e1 = ooo.energy_from_parameters(new[0], new[1]).item()
    assert abs(energy_l[0] - e1) < 1e-8


def test_kupccd_adjoint_path_matches_oracle():
    """OO_pqc on a kUpCCD circuit large enough to take the sector/adjoint path (CAS(6e,6o),
    12 qubits) against the forward-mode path of the same engine, and a small kUpCCD case against
    the oracle."""
    from auto_oo_amd.synthetic import synthetic_problem
    N, ncas, nelecas, nelec = 16, 6, 6, 10
    P = synthetic_problem(N, 77)
    mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="kupccd", k=1)
    assert pqc._use_sector
    oo = aoo.OO_pqc(pqc, mol, ncas, nelecas, oao_mo_coeff=P["oao_mo_coeff"])
    theta = torch.tensor(np.random.default_rng(4).uniform(0, 2 * np.pi, pqc.theta_shape))
    E, grad = oo.energy_and_gradient(theta)
    assert abs(E.item() - oo.energy_from_parameters(theta).item()) < 1e-11
    # forward-mode (tangent RDMs) through the general kernels
    gamma, Gamma = pqc.rdms_with_derivatives(theta)
    res = oo._cas_eval(oo.mo_coeff, gamma, Gamma)
    assert abs(res["E"].item() - E.item()) < 1e-10
    assert (res["dE"] - grad[:pqc.theta_shape]).abs().max() < 1e-9
    assert (res["gvec"][0] - grad[pqc.theta_shape:]).abs().max() < 1e-10
    # dense qnode of the sector engine is normalised and matches get_rdms traces
    psi = pqc.qnode(theta)
    assert abs(float((psi.real ** 2).sum()) - 1.0) < 1e-12
    g1, g2 = pqc.get_rdms(theta)
    assert abs(float(torch.trace(g1)) - nelecas) < 1e-11
    # small kUpCCD against the oracle (energy + full gradient)
    ooo, opqc, oo2, pqc2 = _setup(13, 20261, ncas=3, nelecas=2, nelec=14, ansatz="kupccd", k=2)
    th2 = torch.tensor(np.random.default_rng(6).uniform(0, 2 * np.pi, pqc2.theta_shape))
    E2, g2_ = oo2.energy_and_gradient(th2)
    assert abs(E2.item() - ooo.energy_from_parameters(th2).item()) < 1e-9
    assert (g2_.cpu() - ooo.full_gradient(th2)).abs().max() < 1e-8


def test_np_fabric_full_derivatives_like_reference_test():
    """The reference's test/test_oo_pqc.py::test_full_derivatives (np_fabric, CAS(2,2),
    freeze_active) replayed on synthetic STO-3G-shaped integrals: block gradients / Hessians of the
    engine == joint autodiff of the oracle's energy_from_parameters(theta, kappa)."""
    from auto_oo_amd.synthetic import synthetic_problem
    N, ncas, nelecas, nelec = 13, 2, 2, 16
    P = synthetic_problem(N, 4242)
    mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="np_fabric", n_layers=1)
    oo = aoo.OO_pqc(pqc, mol, ncas, nelecas, oao_mo_coeff=P["oao_mo_coeff"], freeze_active=True)
    omol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    ooo = R.OracleOOPQC(R.OraclePQC(ncas, nelecas, "np_fabric", n_layers=1), omol, ncas, nelecas,
                        P["oao_mo_coeff"], freeze_active=True)
    theta = torch.tensor([0.8324, 0.2490], dtype=torch.float64)     # test/test_oo_pqc.py:83
    kappa = torch.zeros(ooo.n_kappa, dtype=torch.float64)
    J = torch.autograd.functional.jacobian(ooo.energy_from_parameters, (theta, kappa))
    assert (oo.circuit_gradient(theta).cpu() - J[0]).abs().max() < 1e-8
    assert (oo.orbital_gradient(theta).cpu() - J[1]).abs().max() < 1e-8
    H = torch.autograd.functional.hessian(ooo.energy_from_parameters, (theta, kappa))
    assert (oo.circuit_circuit_hessian(theta).cpu() - H[0][0]).abs().max() < 1e-8
    assert (oo.orbital_circuit_hessian(theta).cpu() - H[1][0]).abs().max() < 1e-8
    assert (oo.orbital_orbital_hessian(theta).cpu() - H[1][1]).abs().max() < 1e-7


def test_orbital_optimization_fixed_rdms():
    """OO_energy.orbital_optimization (oo_energy.py:426-474) with fixed RDMs: the first damped
    Newton step reproduces the oracle's, the energy decreases monotonically."""
    ooo, opqc, oo, pqc = _setup(13, 20261)
    theta = torch.tensor(np.random.default_rng(2).uniform(0, 2 * np.pi, pqc.theta_shape))
    g1, g2 = opqc.get_rdms(theta)
    e0 = oo.energy_from_mo_coeff(oo.mo_coeff, g1, g2).item()
    energy_l = oo.orbital_optimization(g1, g2, conv_tol=1e-9, max_iterations=6, verbose=None)
    assert energy_l[0] < e0 and all(b <= a + 1e-10 for a, b in zip(energy_l, energy_l[1:]))
    from functools import partial
    opt = R.OracleNewtonStep()
    kappa0 = torch.zeros(ooo.n_kappa, dtype=torch.float64)
    grad = ooo.kappa_matrix_to_vector(ooo.analytic_gradient(g1, g2))
    hess = ooo.full_hessian_to_matrix(ooo.analytic_hessian(g1, g2))
    newk, _ = opt.damped_newton_step(partial(ooo.energy_from_kappa, one_rdm=g1, two_rdm=g2),
                                     (kappa0,), grad, hess)
    assert abs(ooo.energy_from_kappa(newk, g1, g2).item() - energy_l[0]) < 1e-8


@pytest.mark.parametrize("N,G", [(13, 100), (20, 37), (13, 300), (47, 7), (32, 13), (33, 12), (16, 49)])
def test_batched_evaluation_many_small_geometries(N, G):
    """Batch sizes that are no multiple of anything, on shapes where the library picks the
    persistent T3 path by itself (G * N^2 slabs fill the chip): every geometry of the batch equals
    its single evaluation."""
    from auto_oo_amd.synthetic import synthetic_problem
    ncas, nelecas, nelec = 2, 2, 6
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="ucc")
    mols, coeffs = [], []
    base = [synthetic_problem(N, 700 + g) for g in range(5)]
    for g in range(G):
        P = base[g % 5]
        mols.append(aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"] + 0.01 * g, nelec))
        coeffs.append(P["oao_mo_coeff"])
    batch = aoo.OO_pqc_batch(pqc, mols, ncas, nelecas, oao_mo_coeffs=coeffs)
    rng = np.random.default_rng(8)
    thetas = torch.tensor(rng.uniform(0, 2 * np.pi, (G, pqc.theta_shape)))
    eg = batch.energy_and_gradient(thetas).cpu()
    assert torch.isfinite(eg).all()
    for g in (0, 1, G // 2, G - 2, G - 1):
        single = aoo.OO_pqc(pqc, mols[g], ncas, nelecas, oao_mo_coeff=coeffs[g])
        E, grad = single.energy_and_gradient(thetas[g])
        assert abs(eg[g, 0].item() - E.item()) < 1e-11
        assert (eg[g, 1:] - grad.cpu()).abs().max() < 1e-11


def test_many_occupied_orbitals_use_the_staged_path():
    """A molecule-like shape with many doubly occupied orbitals (N = 30, 20 occupied + CAS(4e,3o):
    M = 23): U[n] / g_mo[n] no longer fit one workgroup's LDS, the library switches to the staged
    kernels.  Energy, theta- and kappa-gradients against the oracle, batched call included."""
    from auto_oo_amd.synthetic import synthetic_problem
    N, ncas, nelecas, nelec = 30, 3, 4, 44
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="ucc")
    P = synthetic_problem(N, 4242)
    mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    oo = aoo.OO_pqc(pqc, mol, ncas, nelecas, oao_mo_coeff=P["oao_mo_coeff"])
    theta = torch.tensor(np.random.default_rng(1).uniform(0, 2 * np.pi, pqc.theta_shape))
    E, grad = oo.energy_and_gradient(theta)
    omol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    ooo = R.OracleOOPQC(R.OraclePQC(ncas, nelecas, "ucc"), omol, ncas, nelecas, P["oao_mo_coeff"])
    E_ref = ooo.energy_from_parameters(theta)
    assert abs(E.item() - E_ref.item()) < 1e-9 * max(1.0, abs(E_ref.item()))
    g_ref = ooo.full_gradient(theta)
    assert (grad.cpu() - g_ref).abs().max() < 1e-8 * max(1.0, float(g_ref.abs().max()))
    batch = aoo.OO_pqc_batch(pqc, [mol, mol], ncas, nelecas,
                             oao_mo_coeffs=[P["oao_mo_coeff"], P["oao_mo_coeff"]])
    eg = batch.energy_and_gradient(torch.stack((theta, theta)))
    assert abs(eg[1, 0].item() - E.item()) < 1e-11
    assert (eg[1, 1:] - grad).abs().max().item() < 1e-11


def test_batch_set_molecule_keeps_flags_and_packed_copy_consistent():
    """OO_pqc_batch.set_molecule: replacing a geometry re-verifies the integrals' symmetry and
    rebuilds its slice of the packed copy (symmetric replacement: still the packed path, results
    equal the single evaluation); a non-symmetric replacement switches the batch to the general
    pipeline, with the same per-geometry results."""
    from auto_oo_amd.synthetic import synthetic_problem
    N, G = 20, 37
    ncas, nelecas, nelec = 2, 2, 6
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="ucc")
    base = [synthetic_problem(N, 900 + g) for g in range(4)]
    mols = [aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"] + 0.01 * g, nelec)
            for g, P in ((g, base[g % 3]) for g in range(G))]
    coeffs = [base[g % 3]["oao_mo_coeff"] for g in range(G)]
    batch = aoo.OO_pqc_batch(pqc, mols, ncas, nelecas, oao_mo_coeffs=coeffs)
    assert batch.eri_flags == 3 and batch._eri_packed is not None
    rng = np.random.default_rng(11)
    thetas = torch.tensor(rng.uniform(0, 2 * np.pi, (G, pqc.theta_shape)))

    def check_geometry(g, mol, coeff):
        eg = batch.energy_and_gradient(thetas).cpu()
        single = aoo.OO_pqc(pqc, mol, ncas, nelecas, oao_mo_coeff=coeff)
        E, grad = single.energy_and_gradient(thetas[g])
        assert abs(eg[g, 0].item() - E.item()) < 1e-11
        assert (eg[g, 1:] - grad.cpu()).abs().max() < 1e-11

    P = base[3]
    new_mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"] + 5.0, nelec)
    batch.set_molecule(5, new_mol, P["oao_mo_coeff"])
    assert batch.eri_flags == 3 and batch._eri_packed is not None
    check_geometry(5, new_mol, P["oao_mo_coeff"])
    check_geometry(4, mols[4], coeffs[4])
    g_bad = P["int2e_ao"].copy()
    g_bad[1, 2, 3, 4] += 1e-3                       # breaks p<->q and r<->s symmetry
    bad_mol = aoo.Moldata(P["int1e_ao"], g_bad, P["overlap"], P["nuc"], nelec)
    batch.set_molecule(7, bad_mol, P["oao_mo_coeff"])
    assert batch.eri_flags == 0 and batch._eri_packed is None
    check_geometry(7, bad_mol, P["oao_mo_coeff"])
    check_geometry(5, new_mol, P["oao_mo_coeff"])
    # the asymmetric geometry replaced by a symmetric one: the flags come BACK (they are kept per
    # geometry, not and-ed away for good) and the packed copy of the whole stack is rebuilt
    e_general = batch.energy_and_gradient(thetas).clone()
    batch.set_molecule(7, new_mol, P["oao_mo_coeff"])
    assert batch.eri_flags == 3 and batch._eri_packed is not None
    check_geometry(7, new_mol, P["oao_mo_coeff"])
    check_geometry(36, mols[36], coeffs[36])
    e_packed = batch.energy_and_gradient(thetas)
    keep = [g for g in range(G) if g != 7]
    assert (e_packed[keep] - e_general[keep]).abs().max().item() < 1e-10


def test_matmul_nn_batch_is_the_per_geometry_product():
    """oovqe_matmul_nn_batch (mo_coeff = S^-1/2 C_oao of a stack, one launch) against torch on the host and,
    bit for bit, against the single-product entry point."""
    from auto_oo_amd import ops
    rng = np.random.default_rng(3)
    for G, M, K, N in ((5, 43, 43, 43), (64, 13, 13, 13), (3, 50, 37, 21), (2, 200, 200, 200)):
        A = torch.tensor(rng.normal(size=(G, M, K)), device="cuda")
        B = torch.tensor(rng.normal(size=(G, K, N)), device="cuda")
        out = ops.matmul_nn_batch(A, B)
        ref = torch.bmm(A.cpu(), B.cpu())
        assert (out.cpu() - ref).abs().max().item() < 1e-11 * K
        for g in (0, G - 1):
            assert torch.equal(out[g], ops.matmul_nn(A[g].contiguous(), B[g].contiguous()))


def test_in_place_edit_of_int2e_ao_reverifies_symmetry_flags():
    """the symmetry flags (and the evaluation plans that bake them in) are
    cached per tensor STATE (object + version counter).  After an in-place edit that breaks the
    p<->q symmetry the energies must equal those of a fresh object built on the edited tensor --
    not those of the half-tensor kernels run on stale flags."""
    from auto_oo_amd.synthetic import synthetic_problem
    P = synthetic_problem(13, 20261)
    mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 16)
    pqc = aoo.Parameterized_circuit(3, 4, None, ansatz="ucc")
    oo = aoo.OO_pqc(pqc, mol, 3, 4, oao_mo_coeff=P["oao_mo_coeff"])
    theta = torch.tensor(np.random.default_rng(1).uniform(0, 2 * np.pi, pqc.theta_shape))
    e_sym, g_sym = oo.energy_and_gradient(theta)
    assert oo._eri_flags() == 3
    oo.int2e_ao[2, 5] += 0.25 * torch.rand((13, 13), dtype=torch.float64, device="cuda")   # in place
    assert oo._eri_flags() == 0
    e_new, g_new = oo.energy_and_gradient(theta)
    g_mod = oo.int2e_ao.cpu().numpy()
    mol2 = aoo.Moldata(P["int1e_ao"], g_mod, P["overlap"], P["nuc"], 16)
    oo2 = aoo.OO_pqc(pqc, mol2, 3, 4, oao_mo_coeff=P["oao_mo_coeff"])
    e_ref, g_ref = oo2.energy_and_gradient(theta)
    assert abs(e_new.item() - e_ref.item()) < 1e-12 and (g_new - g_ref).abs().max() < 1e-12
    assert abs(e_new.item() - e_sym.item()) > 1e-6            # the edit really changed the energy
    assert abs(oo.energy_from_parameters(theta).item() - e_ref.item()) < 1e-12
    omol = R.OracleMol(P["int1e_ao"], g_mod, P["overlap"], P["nuc"], 16)
    ooo = R.OracleOOPQC(R.OraclePQC(3, 4, "ucc"), omol, 3, 4, P["oao_mo_coeff"])
    assert abs(e_new.item() - ooo.energy_from_parameters(theta).item()) < 1e-9


@pytest.mark.parametrize("N,G", [(43, 9), (13, 40), (30, 11), (47, 5)])
def test_packed_stage1_realisations_agree(N, G, lib_options):
    """The three realisations of stage 1 on the packed copy (operand-shaped HBM loads, LDS-DMA ring,
    contiguous register loads with 3 / 4 slabs in flight; debug option tri_mode) form the same sums:
    in the same order: bit-identical energies and gradients, equal to the oracle's."""
    from auto_oo_amd.synthetic import synthetic_problem
    ncas, nelecas, nelec = 3, 4, 16
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="ucc")
    mols, coeffs = [], []
    for g in range(G):
        P = synthetic_problem(N, 900 + g)
        mols.append(aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec))
        coeffs.append(P["oao_mo_coeff"])
    batch = aoo.OO_pqc_batch(pqc, mols, ncas, nelecas, oao_mo_coeffs=coeffs)
    assert batch._eri_packed is not None
    thetas = torch.tensor(np.random.default_rng(2).uniform(0, 2 * np.pi, (G, pqc.theta_shape)))
    lib_options(fused_chunks=1)                      # the batched (packed-triangle) plan at any batch size
    outs = []
    for mode in (1, 2, 3, 4):
        lib_options(tri_mode=mode)
        outs.append(batch.energy_and_gradient(thetas).clone())
    # every mode runs the first products as single accumulator chains in the same order: bit-identical
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    # the builds of the q -> x, p -> n kernel behind it (one / two workgroups per CU; at N = 41 ... 44 the measured
    # and not adopted three-per-CU build): the same chains of products with more or fewer zero k-steps
    lib_options(tri_mode=0)
    for opt in ("gm_one_per_cu", "gm_two_per_cu", "gm_three_per_cu", "gm_plain_grid"):
        lib_options(**{opt: 1})
        assert torch.equal(batch.energy_and_gradient(thetas), outs[0]), opt
        lib_options(**{opt: 0})
    omol = R.OracleMol(mols[0].int1e_ao, mols[0].int2e_ao, mols[0].overlap, mols[0].nuc, nelec)
    ooo = R.OracleOOPQC(R.OraclePQC(ncas, nelecas, "ucc"), omol, ncas, nelecas, coeffs[0])
    assert abs(outs[0][0, 0].item() - ooo.energy_from_parameters(thetas[0]).item()) < 1e-9
    assert (outs[0][0, 1:].cpu() - ooo.full_gradient(thetas[0])).abs().max() < 1e-8


def test_custom_gate_table_ansatz_matches_gate_level_oracle():
    """Where the reference takes a user QNode as ``ansatz`` (src/auto_oo/pqc.py:162-163) this engine
    takes a gate table.  A hand-made circuit -- singles and doubles interleaved, one parameter shared
    by two gates, a DoubleExcitation and an OrbitalRotation from the GateFabric set -- against the
    oracle's gate-level statevector; RDMs, energy and gradient run through the usual entry points."""
    from auto_oo_amd import excitations as X
    ncas, nelecas = 3, 2
    n = 2 * ncas
    singles, doubles = X.excitations(nelecas, n)
    s_wires, d_wires = X.excitations_to_wires(singles, doubles)

    def table(ncas_, nelecas_):
        gates = []
        for i, (w1, w2) in enumerate(d_wires[:4]):
            gates.append(X.fde_gate(w1, w2, n, i))
            gates.append(X.fse_gate(s_wires[i], n, 4 + i % 2))      # parameters 4, 5 shared
        gates.append(X.double_excitation_gate([0, 1, 2, 3], n, 6))
        gates.extend(X.orbital_rotation_gates([2, 3, 4, 5], n, 7))
        return gates

    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz=table)
    assert pqc.ansatz == "custom" and pqc.theta_shape == 8
    theta = torch.tensor(np.random.default_rng(12).uniform(-1.0, 1.0, 8))
    sv = R.Statevector(n, R.hf_state(nelecas, n))
    for i, (w1, w2) in enumerate(d_wires[:4]):
        R.fermionic_double_excitation(sv, theta[i], w1, w2)
        R.fermionic_single_excitation(sv, theta[4 + i % 2], s_wires[i])
    R.double_excitation(sv, theta[6], [0, 1, 2, 3])
    R.orbital_rotation(sv, theta[7], [2, 3, 4, 5])
    ref = sv.vector()
    state = pqc.qnode(theta).cpu()
    assert (state - ref).abs().max() < 1e-12
    # the same table as a plain list, and through the cost function
    pqc2 = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz=table(ncas, nelecas))
    assert torch.equal(pqc2.qnode(theta).cpu(), state)
    P = R.synthetic_problem(9, 77)
    mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 4)
    oo = aoo.OO_pqc(pqc, mol, ncas, nelecas, oao_mo_coeff=P["oao_mo_coeff"])
    g = oo.full_gradient(theta)
    h = 1e-6
    for j in (0, 4, 7):
        tp, tm = theta.clone(), theta.clone()
        tp[j] += h
        tm[j] -= h
        fd = (oo.energy_from_parameters(tp).item() - oo.energy_from_parameters(tm).item()) / (2 * h)
        assert abs(fd - g[j].item()) < 1e-7


def test_device_generated_geometries_and_reverify_integrals():
    """Integrals written into the batch's stacked tensors on the device (synthetic_problem_device, as
    bench.py does for its 1024 geometries) + OO_pqc_batch.reverify_integrals(): flags re-verified,
    packed copy rebuilt, mo_coeff refreshed -- every geometry then equals its single evaluation built
    from the same tensors through Moldata; a broken symmetry is noticed."""
    from auto_oo_amd.synthetic import synthetic_problem, synthetic_problem_device
    from auto_oo_amd import ops
    N, ncas, nelecas, nelec, G = 17, 3, 4, 8, 3
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="ucc")
    P0 = synthetic_problem(N, 4100)
    mol0 = aoo.Moldata(P0["int1e_ao"], P0["int2e_ao"], P0["overlap"], P0["nuc"], nelec)
    batch = aoo.OO_pqc_batch(pqc, [mol0] * G, ncas, nelecas, oao_mo_coeffs=[P0["oao_mo_coeff"]] * G)
    dev = []
    for slot in range(1, G):
        P = synthetic_problem_device(N, 4100 + slot, "cuda")
        assert ops.eri_flags(P["int2e_ao"]) == 3
        batch.int2e_ao[slot].copy_(P["int2e_ao"])
        batch.int1e_ao[slot].copy_(P["int1e_ao"])
        batch.oao_coeff[slot].copy_(P["oao_coeff"])
        batch.oao_mo_coeff[slot].copy_(P["oao_mo_coeff"])
        batch.nuc[slot] = P["nuc"] + slot
        dev.append(P)
    batch.reverify_integrals()
    assert batch.eri_flags == 3 and batch._eri_packed is not None
    thetas = torch.tensor(np.random.default_rng(2).uniform(0, 2 * np.pi, (G, pqc.theta_shape)))
    eg = batch.energy_and_gradient(thetas).cpu()
    for slot in range(1, G):
        P = dev[slot - 1]
        mol = aoo.Moldata(P["int1e_ao"].cpu().numpy(), P["int2e_ao"].cpu().numpy(), P["overlap"].cpu().numpy(),
                          P["nuc"] + slot, nelec)
        assert np.abs(mol.oao_coeff - P["oao_coeff"].cpu().numpy()).max() < 1e-12
        single = aoo.OO_pqc(pqc, mol, ncas, nelecas, oao_mo_coeff=P["oao_mo_coeff"].cpu().numpy())
        E, grad = single.energy_and_gradient(thetas[slot])
        assert abs(eg[slot, 0].item() - E.item()) < 1e-10
        assert (eg[slot, 1:] - grad.cpu()).abs().max() < 1e-10
    batch.int2e_ao[1, 0, 1, 2, 3] += 1e-9            # r<->s and p<->q symmetry of geometry 1 broken
    batch.reverify_integrals()
    assert batch.eri_flags == 0 and batch._eri_packed is None
    eg2 = batch.energy_and_gradient(thetas).cpu()
    assert (eg2[2] - eg[2]).abs().max() < 1e-10      # other geometries: same numbers on the general path
