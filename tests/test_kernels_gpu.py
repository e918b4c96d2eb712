"""GPU parity tests of the individual HIP kernels (through the C ABI) against the CPU oracle and
the reference's golden vectors.  Tolerances: fp64 rounding level (<= 1e-11 absolute on O(1..50)
data); golden literals at the reference's own allclose(rtol=1e-5, atol=1e-8)."""
import json
import os

import numpy as np
import pytest
import scipy.linalg
import torch

pytestmark = pytest.mark.gpu

import auto_oo_amd as aoo                            # noqa: E402
from auto_oo_amd import ops, excitations as X   # noqa: E402
from oracle import cpu_ref as R                  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DEV = "cuda"


def _load(name):
    with open(os.path.join(HERE, "golden", name)) as fh:
        return json.load(fh)


def _rand(rng, *shape):
    return torch.tensor(rng.standard_normal(shape))


@pytest.mark.parametrize("A,K,J,B", [(1, 13, 13, 13 ** 3), (13, 13, 13, 169), (169, 13, 13, 13),
                                     (3, 43, 9, 81), (1, 43, 43, 729), (5, 17, 33, 7),
                                     (2, 100, 250, 40), (1, 4, 1, 1), (7, 41, 16, 16)])
def test_mode_contract_inner(A, K, J, B):
    rng = np.random.default_rng(A * 1000 + K)
    T = _rand(rng, A, K, B)
    C = _rand(rng, K, J)
    ref = torch.einsum("kj,akb->ajb", C, T)
    out = ops.mode_contract(T.to(DEV), C.to(DEV), A, K, J, B, last=False).cpu().reshape(A, J, B)
    assert (out - ref).abs().max() < 1e-11 * max(1.0, ref.abs().max())


@pytest.mark.parametrize("A,K,J", [(13 ** 3, 13, 13), (43, 43, 9), (100, 43, 43), (1, 5, 3),
                                   (37, 70, 210), (16, 16, 16)])
def test_mode_contract_last(A, K, J):
    rng = np.random.default_rng(A + K)
    T = _rand(rng, A, K)
    C = _rand(rng, K, J)
    ref = T @ C
    out = ops.mode_contract(T.to(DEV), C.to(DEV), A, K, J, 1, last=True).cpu().reshape(A, J)
    assert (out - ref).abs().max() < 1e-11 * max(1.0, ref.abs().max())


def test_mode_contract_asymmetric_identity():
    """A = I with an asymmetric second operand catches a swapped C/D row/col map."""
    K = 32
    T = torch.arange(K * 48, dtype=torch.float64).reshape(1, K, 48)
    out = ops.mode_contract(T.to(DEV), torch.eye(K, dtype=torch.float64).to(DEV), 1, K, K, 48,
                            last=False).cpu().reshape(K, 48)
    assert torch.equal(out, T[0])
    T2 = torch.arange(40 * K, dtype=torch.float64).reshape(40, K)
    out2 = ops.mode_contract(T2.to(DEV), torch.eye(K, dtype=torch.float64).to(DEV), 40, K, K, 1,
                             last=True).cpu().reshape(40, K)
    assert torch.equal(out2, T2)


def test_matmuls():
    rng = np.random.default_rng(3)
    A, B = _rand(rng, 43, 43), _rand(rng, 43, 43)
    assert (ops.matmul_nn(A.to(DEV), B.to(DEV)).cpu() - A @ B).abs().max() < 1e-12
    assert (ops.matmul_tn(A.to(DEV), B.to(DEV)).cpu() - A.T @ B).abs().max() < 1e-12
    A, B = _rand(rng, 30, 17), _rand(rng, 17, 55)
    assert (ops.matmul_nn(A.to(DEV), B.to(DEV)).cpu() - A @ B).abs().max() < 1e-12
    A, B = _rand(rng, 17, 30), _rand(rng, 17, 55)
    assert (ops.matmul_tn(A.to(DEV), B.to(DEV)).cpu() - A.T @ B).abs().max() < 1e-12


@pytest.mark.parametrize("N", [13, 43, 20])
def test_general_4index_transform(N):
    rng = np.random.default_rng(N)
    M = _rand(rng, N, N, N, N)
    Cs = [_rand(rng, N, N) for _ in range(4)]
    ref = R.general_4index_transform(M, *Cs)
    out = ops.general_4index_transform(M.to(DEV), *[c.to(DEV) for c in Cs]).cpu()
    assert (out - ref).abs().max() < 1e-11 * ref.abs().max()


@pytest.mark.parametrize("N", [4, 13, 43, 48, 49, 64, 100])
def test_expm(N):
    rng = np.random.default_rng(N)
    for scale in (0.01, 0.3, 3.0):
        Xm = _rand(rng, N, N) * scale / np.sqrt(N)
        Xm = Xm - Xm.T
        # torch.linalg.matrix_exp (what the reference calls, oo_energy.py:230) is itself only
        # ~1e-11 accurate for small norms (low-degree Taylor): parity at 1e-10 against it, and
        # fp64-rounding accuracy against scipy's Pade-13 expm.
        ref = torch.linalg.matrix_exp(-Xm)
        exact = torch.tensor(scipy.linalg.expm(-Xm.numpy()))
        out = ops.expm(Xm.to(DEV), sign=-1.0).cpu()
        assert (out - ref).abs().max() < 1e-10
        assert (out - exact).abs().max() < 2e-13
        assert (out.T @ out - torch.eye(N, dtype=torch.float64)).abs().max() < 1e-12


@pytest.mark.parametrize("N,no,na", [(13, 6, 3), (43, 6, 3), (60, 5, 4)])
def test_expm_skew(N, no, na):
    occ = list(range(no)); act = list(range(no, no + na)); virt = list(range(no + na, N))
    pidx = X.non_redundant_indices(occ, act, virt, False)
    rows, cols = X.tril_tables(N, pidx)
    rng = np.random.default_rng(N)
    kappa = torch.tensor(rng.normal(0, 0.05, len(pidx)))
    total = torch.zeros(N * (N - 1) // 2, dtype=torch.float64)
    total[pidx] = kappa
    Kref = R.vector_to_skew_symmetric(total)
    Uref = torch.linalg.matrix_exp(-Kref)
    U, K = ops.expm_skew(kappa.to(DEV), torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV), N,
                         want_K=True)
    assert torch.equal(K.cpu(), Kref)
    assert (U.cpu() - Uref).abs().max() < 1e-10
    assert (U.cpu() - torch.tensor(scipy.linalg.expm(-Kref.numpy()))).abs().max() < 2e-13


def _gates_dev(gates):
    return torch.tensor(X.gates_to_numpy(gates)).to(DEV)


@pytest.mark.parametrize("case", [c for c in _load("pqc_states.json") if c["ansatz"] == "ucc"],
                         ids=lambda c: c["source"])
def test_circuit_state_golden(case):
    n = 2 * case["ncas"]
    gates, n_theta = X.uccd_gates(case["ncas"], case["nelecas"], bool(case["add_singles"]))
    theta = torch.tensor([case["theta"]], dtype=torch.float64).to(DEV)
    psi = ops.circuit_state(theta, _gates_dev(gates), len(gates), n,
                            X.basis_index(X.hf_state(case["nelecas"], n))).cpu().numpy()[0]
    ref = np.array(case["state_real"])
    assert np.allclose(psi, ref, rtol=1e-5, atol=1e-8)
    if not case["source"].endswith(":36"):
        assert np.abs(psi - ref).max() < 1e-8


@pytest.mark.parametrize("case", [c for c in _load("pqc_rdms.json") if c["ansatz"] == "ucc"],
                         ids=lambda c: c["source"])
def test_rdms_golden(case):
    ncas = case["ncas"]
    n = 2 * ncas
    gates, _ = X.uccd_gates(ncas, case["nelecas"], bool(case["add_singles"]))
    theta = torch.tensor([case["theta"]], dtype=torch.float64).to(DEV)
    psi = ops.circuit_state(theta, _gates_dev(gates), len(gates), n,
                            X.basis_index(X.hf_state(case["nelecas"], n)))
    g1, g2 = ops.rdms(psi, psi, ncas)
    assert np.allclose(g1.cpu().numpy()[0], np.array(case["one_rdm"]), rtol=1e-5, atol=1e-8)
    assert np.allclose(g2.cpu().numpy()[0], np.array(case["two_rdm"]), rtol=1e-5, atol=1e-8)
    assert np.abs(g1.cpu().numpy()[0] - np.array(case["one_rdm"])).max() < 1e-8
    assert np.abs(g2.cpu().numpy()[0] - np.array(case["two_rdm"])).max() < 1e-8


@pytest.mark.parametrize("ncas,nelecas,kind", [(3, 4, "uccd"), (3, 4, "uccsd"), (4, 4, "kupccd"),
                                               (2, 2, "uccd")])
def test_state_tangents_and_rdms_vs_oracle(ncas, nelecas, kind):
    n = 2 * ncas
    if kind == "kupccd":
        gates, n_theta = X.kupccd_gates(ncas, 1)
        pqc = R.OraclePQC(ncas, nelecas, "kupccd", k=1)
    else:
        gates, n_theta = X.uccd_gates(ncas, nelecas, kind == "uccsd")
        pqc = R.OraclePQC(ncas, nelecas, "ucc", add_singles=(kind == "uccsd"))
    rng = np.random.default_rng(11)
    th = torch.tensor(rng.uniform(0, 2 * np.pi, (2, n_theta)))
    psi, dpsi = ops.circuit_state(th.to(DEV), _gates_dev(gates), len(gates), n,
                                  X.basis_index(X.hf_state(nelecas, n)), tangents=True)
    for b in range(2):
        ref = pqc.qnode(th[b]).real
        assert (psi[b].cpu() - ref).abs().max() < 1e-13
        jac = torch.func.jacfwd(lambda t: pqc.qnode(t).real)(th[b])  # [D, n_theta]
        assert (dpsi[b].cpu() - jac.T).abs().max() < 1e-12
    g1, g2 = ops.rdms(psi, psi, ncas)
    for b in range(2):
        r1, r2 = pqc.get_rdms(th[b])
        assert (g1[b].cpu() - r1).abs().max() < 1e-12
        assert (g2[b].cpu() - r2).abs().max() < 1e-12


def _problem(N, seed, nelec, ncas, nelecas):
    P = R.synthetic_problem(N, seed)
    mol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    pqc = R.OraclePQC(ncas, nelecas, "ucc")
    oo = R.OracleOOPQC(pqc, mol, ncas, nelecas, P["oao_mo_coeff"])
    return P, oo, pqc


@pytest.mark.parametrize("N,seed", [(13, 20261), (43, 20262)])
def test_cas_pipeline_vs_oracle(N, seed):
    ncas, nelecas = 3, 4
    P, oo, pqc = _problem(N, seed, 16, ncas, nelecas)
    no = len(oo.occ_idx)
    M = no + ncas
    C = oo.mo_coeff
    theta = torch.tensor(np.random.default_rng(5).uniform(0, 2 * np.pi, pqc.theta_shape))
    g1, g2 = pqc.get_rdms(theta)
    # oracle
    E_ref = oo.energy_from_parameters(theta)
    c0r, c1r, c2r = oo.get_active_integrals(C)
    G_ref = oo.analytic_gradient(g1, g2)
    gv_ref = oo.kappa_matrix_to_vector(G_ref)
    g_mo = R.int2e_transform(oo.int2e_ao, C)
    h_mo = R.int1e_transform(oo.int1e_ao, C)
    F_ref = oo.fock_generalized(h_mo, g_mo, g1, g2)
    # HIP
    Cd = C.to(DEV).contiguous()
    gd = oo.int2e_ao.to(DEV).contiguous()
    T2 = ops.cas_half_transform(gd, Cd, M)
    T2_ref = torch.einsum("ry,pqrs,sz->pqyz", C[:, :M], oo.int2e_ao, C[:, :M])
    assert (T2.cpu() - T2_ref).abs().max() < 1e-11
    Gm, hmo = ops.cas_finish_transform(T2, oo.int1e_ao.to(DEV).contiguous(), Cd, M)
    assert (Gm.cpu() - g_mo[:, :M, :M, :M]).abs().max() < 1e-11
    assert (hmo.cpu() - h_mo[:, :M]).abs().max() < 1e-11
    rows, cols = X.tril_tables(N, oo.params_idx)
    res = ops.cas_energy_gradient(Gm, hmo, g1.to(DEV)[None].contiguous(),
                                  g2.to(DEV)[None].contiguous(), oo.nuc, no, ncas,
                                  torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV))
    assert abs(res["E"].item() - E_ref.item()) < 1e-9
    assert abs(res["c0"].item() - float(c0r)) < 1e-9
    assert (res["c1"].cpu() - c1r).abs().max() < 1e-11
    assert (res["c2"].cpu() - c2r).abs().max() < 1e-11
    assert (res["fock"].cpu() - F_ref).abs().max() < 1e-10
    assert (res["gmat"].cpu() - G_ref).abs().max() < 1e-10
    assert (res["gvec"].cpu()[0] - gv_ref).abs().max() < 1e-10


@pytest.mark.parametrize("path", ["auto", "t3x1", "t3x3", "t3x6", "t2"])
@pytest.mark.parametrize("N,seed,nelec,ncas,nelecas", [(13, 20261, 16, 3, 4), (43, 20262, 16, 3, 4),
                                                     (20, 7, 8, 2, 2), (16, 9, 4, 4, 4),
                                                     (27, 11, 8, 3, 2), (47, 12, 10, 3, 4)])
def test_cas_eval_fused_matches_staged_and_oracle(N, seed, nelec, ncas, nelecas, path, lib_options):
    """oovqe_cas_eval (4 launches, column kernel) against the staged kernels and the oracle, with
    a stack of RDM sets (set 0 = RDMs, sets >= 1 = arbitrary 'derivative' RDMs).  `path` forces the
    persistent T3 kernel with 1/3/6 chunks of the q range, or the T2 kernels (library test hooks)."""
    if path.startswith("t3x"):
        lib_options(fused_chunks=int(path[3:]))
    elif path == "t2":
        lib_options(cas_unfused=1)
    P = R.synthetic_problem(N, seed)
    mol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    pqc = R.OraclePQC(ncas, nelecas, "ucc")
    oo = R.OracleOOPQC(pqc, mol, ncas, nelecas, P["oao_mo_coeff"])
    no = len(oo.occ_idx)
    M = no + ncas
    C = oo.mo_coeff
    rng = np.random.default_rng(seed)
    theta = torch.tensor(rng.uniform(0, 2 * np.pi, pqc.theta_shape))
    g1, g2 = pqc.get_rdms(theta)
    nrdm = 3
    gam = torch.stack([g1] + [_rand(rng, ncas, ncas) for _ in range(nrdm - 1)])
    Gam = torch.stack([g2] + [_rand(rng, ncas, ncas, ncas, ncas) for _ in range(nrdm - 1)])
    rows, cols = X.tril_tables(N, oo.params_idx)
    rows_d, cols_d = torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV)
    Cd = C.to(DEV).contiguous()
    gd = oo.int2e_ao.to(DEV).contiguous()
    hd = oo.int1e_ao.to(DEV).contiguous()
    res = ops.cas_eval(gd, hd, Cd, gam.to(DEV).contiguous(), Gam.to(DEV).contiguous(), oo.nuc, no,
                       ncas, rows_d, cols_d, want_matrices=True, want_integrals=True)
    T2 = ops.cas_half_transform(gd, Cd, M)
    Gm, hmo = ops.cas_finish_transform(T2, hd, Cd, M)
    st = ops.cas_energy_gradient(Gm, hmo, gam.to(DEV).contiguous(), Gam.to(DEV).contiguous(),
                                 oo.nuc, no, ncas, rows_d, cols_d)
    # the one-workgroup entry point of the same stage (oovqe_cas_energy_gradient, no scratch)
    st1 = ops.cas_energy_gradient(Gm, hmo, gam.to(DEV).contiguous(), Gam.to(DEV).contiguous(),
                                  oo.nuc, no, ncas, rows_d, cols_d, one_workgroup=True)
    for key in ("c0", "c1", "c2", "E", "fock", "gmat", "gvec", "dE"):
        assert (st1[key] - st[key]).abs().max() < 1e-11 * max(1.0, float(st[key].abs().max())), key
    g_mo = R.int2e_transform(oo.int2e_ao, C)
    h_mo = R.int1e_transform(oo.int1e_ao, C)
    assert (res["Gm"].cpu() - g_mo[:, :M, :M, :M]).abs().max() < 1e-11
    assert (res["hmo"].cpu() - h_mo[:, :M]).abs().max() < 1e-11
    for key in ("c0", "c1", "c2", "E", "fock", "gmat", "gvec", "dE"):
        assert (res[key] - st[key]).abs().max() < 1e-10, key
    assert abs(res["E"].item() - oo.energy_from_mo_coeff(C, g1, g2).item()) < 1e-9
    gv_ref = oo.kappa_matrix_to_vector(oo.analytic_gradient(g1, g2))
    assert (res["gvec"].cpu()[0] - gv_ref).abs().max() < 1e-10


@pytest.mark.parametrize("ncas,nelecas,kind", [(3, 4, "uccd"), (3, 4, "uccsd"), (2, 2, "uccd"),
                                               (2, 2, "uccsd"), (4, 4, "kupccd"), (3, 2, "kupccd")])
def test_circuit_rdms_fused_matches_staged(ncas, nelecas, kind):
    """oovqe_circuit_rdms (one-workgroup LDS kernel with the MFMA Gram for small active spaces,
    chained kernels otherwise) against oovqe_circuit_state + oovqe_rdms_tangent."""
    n = 2 * ncas
    if kind == "kupccd":
        gates, n_theta = X.kupccd_gates(ncas, 1)
    else:
        gates, n_theta = X.uccd_gates(ncas, nelecas, kind == "uccsd")
    rng = np.random.default_rng(23)
    th = torch.tensor(rng.uniform(0, 2 * np.pi, (3, n_theta))).to(DEV)
    gd = _gates_dev(gates)
    init = X.basis_index(X.hf_state(nelecas, n))
    psi, dpsi = ops.circuit_state(th, gd, len(gates), n, init, tangents=True)
    g1, g2 = ops.rdms_tangent(psi, dpsi, ncas)
    f1, f2, fpsi, fdpsi = ops.circuit_rdms(th, gd, len(gates), n, ncas, init, tangents=True,
                                           want_states=True)
    assert (fpsi - psi).abs().max() < 1e-14 and (fdpsi - dpsi).abs().max() < 1e-14
    assert (f1 - g1).abs().max() < 1e-12 and (f2 - g2).abs().max() < 1e-12
    n1, n2 = ops.circuit_rdms(th, gd, len(gates), n, ncas, init, tangents=False)
    assert (n1[:, 0] - g1[:, 0]).abs().max() < 1e-12 and (n2[:, 0] - g2[:, 0]).abs().max() < 1e-12


@pytest.mark.parametrize("ncas,nelecas,kind,k", [(4, 4, "kupccd", 1), (4, 4, "kupccd", 2),
                                                 (3, 4, "uccd", 1), (3, 4, "uccsd", 1),
                                                 (3, 2, "kupccd", 1), (4, 6, "uccd", 1)])
def test_sector_engine_state_rdms_adjoint(ncas, nelecas, kind, k):
    """(N_alpha,N_beta)-sector engine: state and RDMs against the dense kernels, reverse-mode
    theta-gradient against forward-mode tangent RDMs and against the oracle's autograd."""
    from auto_oo_amd.sector import SectorEngine
    n = 2 * ncas
    if kind == "kupccd":
        gates, n_theta = X.kupccd_gates(ncas, k)
    else:
        gates, n_theta = X.uccd_gates(ncas, nelecas, kind == "uccsd")
    hf = X.hf_state(nelecas, n)
    init = X.basis_index(hf)
    gd = _gates_dev(gates)
    eng = SectorEngine(ncas, hf, gd, len(gates), n_theta, init, torch.device(DEV))
    assert eng.fits()
    rng = np.random.default_rng(31)
    th = torch.tensor(rng.uniform(0, 2 * np.pi, (3, n_theta))).to(DEV)
    psi_c, psi = eng.state(th, dense=True)
    ref_psi, dpsi = ops.circuit_state(th, gd, len(gates), n, init, tangents=True)
    assert (psi - ref_psi).abs().max() < 1e-13
    assert abs(float((psi_c ** 2).sum(dim=1).min()) - 1.0) < 1e-12     # all weight in the sector
    g1, g2 = eng.rdms(psi_c)
    r1, r2 = ops.rdms_tangent(ref_psi, dpsi, ncas)
    assert (g1 - r1[:, 0]).abs().max() < 1e-12 and (g2 - r2[:, 0]).abs().max() < 1e-12
    # random, deliberately NON-symmetric cotangents
    c1 = torch.tensor(rng.standard_normal((ncas, ncas))).to(DEV)
    c2 = torch.tensor(rng.standard_normal((ncas,) * 4)).to(DEV)
    dth = eng.adjoint(th, psi_c, c1, c2).cpu()
    fwd = (torch.einsum("pq,bkpq->bk", c1, r1[:, 1:]) + torch.einsum("pqrs,bkpqrs->bk", c2, r2[:, 1:])).cpu()
    assert (dth - fwd).abs().max() < 1e-11
    if ncas <= 3 or kind == "kupccd" and k == 1:
        pqc = (R.OraclePQC(ncas, nelecas, "kupccd", k=k) if kind == "kupccd"
               else R.OraclePQC(ncas, nelecas, "ucc", add_singles=(kind == "uccsd")))

        def f(t):
            a, b = pqc.get_rdms(t)
            return (c1.cpu() * a).sum() + (c2.cpu() * b).sum()
        ref = torch.autograd.functional.jacobian(f, th[0].cpu())
        assert (dth[0] - ref).abs().max() < 1e-10


@pytest.mark.parametrize("N,M", [(5, 3), (16, 16), (17, 4), (31, 9), (32, 12), (33, 7), (43, 9),
                                 (44, 16), (47, 5), (48, 10)])
def test_half_transform_sizes(N, M):
    """T2[p,q,y,z] = sum_rs C[r,y] g[p,q,r,s] C[s,z] for every register-chunk variant of the
    persistent slab kernel (N <= 48, M <= 16), odd and even slab counts per wave included."""
    rng = np.random.default_rng(1000 * N + M)
    g = torch.tensor(rng.standard_normal((N, N, N, N)))
    C = torch.tensor(rng.standard_normal((N, N)))
    T2 = ops.cas_half_transform(g.to(DEV), C.to(DEV), M).cpu()
    ref = torch.einsum("ry,pqrs,sz->pqyz", C[:, :M], g, C[:, :M])
    assert T2.shape == ref.shape
    assert (T2 - ref).abs().max() < 1e-11 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("path", ["t3x1", "t3x2", "t2"])
@pytest.mark.parametrize("N,nelec,ncas,nelecas", [(43, 16, 8, 8), (30, 14, 10, 8), (24, 12, 6, 6)])
def test_cas_eval_large_active_space_random_rdms(N, nelec, ncas, nelecas, path, lib_options):
    """CAS path with M = n_occ + ncas up to 13 (config-5-like active spaces) on every transform
    path; the kernels are linear in the RDMs, so random (unphysical) RDM sets exercise them fully.
    Checked against the oracle's energy, CAS coefficients and analytic orbital gradient."""
    if path.startswith("t3x"):
        lib_options(fused_chunks=int(path[3:]))
    else:
        lib_options(cas_unfused=1)
    P = R.synthetic_problem(N, 500 + N)
    mol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    oo = R.OracleOOEnergy(mol, ncas, nelecas, P["oao_mo_coeff"])
    no = len(oo.occ_idx)
    C = oo.mo_coeff
    rng = np.random.default_rng(N)
    nrdm = 2
    gam = torch.stack([_rand(rng, ncas, ncas) for _ in range(nrdm)])
    Gam = torch.stack([_rand(rng, ncas, ncas, ncas, ncas) for _ in range(nrdm)])
    rows, cols = X.tril_tables(N, oo.params_idx)
    res = ops.cas_eval(oo.int2e_ao.to(DEV).contiguous(), oo.int1e_ao.to(DEV).contiguous(),
                       C.to(DEV).contiguous(), gam.to(DEV).contiguous(), Gam.to(DEV).contiguous(),
                       oo.nuc, no, ncas, torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV),
                       want_matrices=True, want_integrals=True)
    c0r, c1r, c2r = oo.get_active_integrals(C)
    g_mo = R.int2e_transform(oo.int2e_ao, C)
    M = no + ncas
    scale = float(g_mo.abs().max())
    assert (res["Gm"].cpu() - g_mo[:, :M, :M, :M]).abs().max() < 1e-11 * max(1.0, scale)
    assert abs(res["c0"].item() - float(c0r)) < 1e-9
    assert (res["c1"].cpu() - c1r).abs().max() < 1e-10
    assert (res["c2"].cpu() - c2r).abs().max() < 1e-10
    E_ref = oo.energy_from_mo_coeff(C, gam[0], Gam[0])
    assert abs(res["E"].item() - E_ref.item()) < 1e-8 * max(1.0, abs(E_ref.item()))
    gv_ref = oo.kappa_matrix_to_vector(oo.analytic_gradient(gam[0], Gam[0]))
    assert (res["gvec"].cpu()[0] - gv_ref).abs().max() < 1e-8 * max(1.0, float(gv_ref.abs().max()))


@pytest.mark.parametrize("N,M", [(52, 20), (64, 10), (70, 33), (96, 12)])
def test_half_transform_streaming_sizes(N, M):
    """Shapes beyond the one-chunk slab kernels (N > 48) and M > 16 (two / three 16-wide tiles of
    occupied + active orbitals): the streaming half-transform against the einsum."""
    rng = np.random.default_rng(7 * N + M)
    g = torch.tensor(rng.standard_normal((N, N, N, N)))
    C = torch.tensor(rng.standard_normal((N, N)))
    T2 = ops.cas_half_transform(g.to(DEV), C.to(DEV), M).cpu()
    ref = torch.einsum("ry,pqrs,sz->pqyz", C[:, :M], g, C[:, :M])
    assert (T2 - ref).abs().max() < 1e-11 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("N,nelec,ncas,nelecas", [(56, 12, 4, 4), (50, 40, 4, 4), (64, 60, 3, 2)])
def test_cas_eval_beyond_fused_shapes(N, nelec, ncas, nelecas):
    """The whole CAS path on shapes that only the T2 path covers (N > 48), including N * M^2 too
    large for the column kernel's LDS (M = 22, 32: staged kernels, g_mo streamed from memory)."""
    P = R.synthetic_problem(N, 900 + N)
    mol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    oo = R.OracleOOEnergy(mol, ncas, nelecas, P["oao_mo_coeff"])
    no = len(oo.occ_idx)
    C = oo.mo_coeff
    rng = np.random.default_rng(N)
    gam = torch.stack([_rand(rng, ncas, ncas)])
    Gam = torch.stack([_rand(rng, ncas, ncas, ncas, ncas)])
    rows, cols = X.tril_tables(N, oo.params_idx)
    res = ops.cas_eval(oo.int2e_ao.to(DEV).contiguous(), oo.int1e_ao.to(DEV).contiguous(),
                       C.to(DEV).contiguous(), gam.to(DEV).contiguous(), Gam.to(DEV).contiguous(),
                       oo.nuc, no, ncas, torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV))
    E_ref = oo.energy_from_mo_coeff(C, gam[0], Gam[0])
    assert abs(res["E"].item() - E_ref.item()) < 1e-8 * max(1.0, abs(E_ref.item()))
    gv_ref = oo.kappa_matrix_to_vector(oo.analytic_gradient(gam[0], Gam[0]))
    assert (res["gvec"].cpu()[0] - gv_ref).abs().max() < 1e-8 * max(1.0, float(gv_ref.abs().max()))


def test_eri_symmetry_flags():
    """oovqe_eri_symmetry_flags: exact p<->q and r<->s symmetry are detected separately; one flipped
    bit, a general tensor, or one bad geometry of a stack turn the flags off."""
    N = 11
    BOTH = ops.ERI_PQ_SYMMETRIC | ops.ERI_RS_SYMMETRIC
    P = R.synthetic_problem(N, 4242)
    g = torch.tensor(P["int2e_ao"]).to(DEV).contiguous()
    assert ops.eri_flags(g) == BOTH
    g2 = g.clone()
    g2[3, 7, 2, 5] = g2[3, 7, 2, 5] * (1.0 + 2.0 ** -52)
    assert ops.eri_flags(g2) == 0
    g2[3, 7, 5, 2] = g2[3, 7, 2, 5]          # r<->s repaired, p<->q still broken
    assert ops.eri_flags(g2) == ops.ERI_RS_SYMMETRIC
    rng = np.random.default_rng(5)
    A = torch.tensor(rng.standard_normal((N, N, N, N))).to(DEV)
    assert ops.eri_flags(A) == 0
    assert ops.eri_flags((A + A.transpose(0, 1)).contiguous()) == ops.ERI_PQ_SYMMETRIC
    assert ops.eri_flags((A + A.transpose(2, 3)).contiguous()) == ops.ERI_RS_SYMMETRIC
    stack = torch.stack([g, g, g]).contiguous()
    assert ops.eri_flags(stack) == BOTH
    stack[2, 1, 0, 0, 0] += 1.0
    assert ops.eri_flags(stack) == ops.ERI_RS_SYMMETRIC


@pytest.mark.parametrize("N", [5, 13, 16, 43, 48])
def test_eri_ingest_one_pass_equals_check_and_pack(N):
    """oovqe_eri_ingest (N <= 48: symmetry tests and packed copy from ONE pass over the stack) against torch's own
    comparison of the tensor with its transposes and against oovqe_eri_pack: per-geometry flags for a stack whose
    geometries break the symmetries in different ways (one bit flipped in a slab p < q, in a mirror slab q > p only,
    in one element (r, s) of a diagonal slab), the packed copy bit for bit."""
    rng = np.random.default_rng(100 + N)
    g = torch.tensor(R.synthetic_problem(N, 4300 + N)["int2e_ao"]).to(DEV).contiguous()
    stack = torch.stack([g] * 6).contiguous()
    eps = 1.0 + 2.0 ** -52
    stack[1, 0, N - 1, 1, 2] *= eps                 # slab (0, N-1), one element: p<->q and r<->s broken
    stack[2, 0, N - 1, 1, 2] *= eps
    stack[2, 0, N - 1, 2, 1] = stack[2, 0, N - 1, 1, 2]        # r<->s repaired: p<->q only
    stack[3, N - 1, 0, 1, 2] *= eps                 # the MIRROR slab (q, p) alone: both broken (its own r<->s test)
    stack[4, 2, 2, 0, N - 1] *= eps                 # a diagonal slab: r<->s only
    stack[5] = torch.tensor(rng.standard_normal((N, N, N, N))).to(DEV)
    want = []
    for k in range(6):
        t = stack[k]
        want.append((1 if torch.equal(t, t.transpose(0, 1)) else 0) | (2 if torch.equal(t, t.transpose(2, 3)) else 0))
    assert want == [3, 0, 2, 0, 1, 0]
    flags, packed = ops.eri_ingest(stack)
    assert flags == want
    assert torch.equal(packed[0], ops.eri_pack(g))
    assert torch.equal(packed, ops.eri_pack(stack))           # (the copy is made whatever the flags say)
    flags1, packed1 = ops.eri_ingest(g)
    assert flags1 == [3] and torch.equal(packed1, packed[0])
    assert ops.eri_ingest(stack, pack=False) == (want, None)
    assert ops.eri_flags(stack) == 0 and ops.eri_flags(stack[:1].contiguous()) == 3


@pytest.mark.parametrize("flags", [1, 3])
@pytest.mark.parametrize("path", ["auto", "t3x1", "t3x2", "two_step", "simple", "mirror", "t2"])
@pytest.mark.parametrize("N,nelec,ncas,nelecas", [(13, 16, 3, 4), (43, 16, 3, 4), (17, 8, 4, 4),
                                                  (30, 14, 10, 8), (12, 16, 3, 4), (56, 12, 4, 4),
                                                  (50, 40, 4, 4), (70, 40, 6, 6)])
def test_cas_eval_pq_symmetric_integrals(N, nelec, ncas, nelecas, path, flags, lib_options):
    """eri_flags = ERI_PQ_SYMMETRIC (only the slabs p <= q of g_ao are read), alone and together with
    ERI_RS_SYMMETRIC (only the columns y <= z of the packed triangle are kept), against eri_flags = 0
    on the same symmetric integrals, on every realisation: packed triangle + the one-launch
    q->x / p->n kernel (forced by option fused_chunks, which makes the call take the batched plan) and
    its two-launch and simple-kernel variants, mirrored T2 (one-chunk and streaming half-transform
    kernels), staged fallback.  Same energy, coefficients, gradients and
    g_mo to rounding; the flag-free result is itself checked against the oracle."""
    if path.startswith("t3x"):
        lib_options(fused_chunks=int(path[3:]))
    elif path == "two_step":      # packed triangle, then the q -> x kernel and K1 as two launches
        lib_options(fused_chunks=1)
        lib_options(sym_two_step=1)
    elif path == "simple":        # packed triangle written by the one-slab-per-wave kernel
        lib_options(fused_chunks=1)
        lib_options(sym_simple=1)
    elif path == "mirror":
        lib_options(fused_chunks=1)
        lib_options(sym_mirror=1)
    elif path == "t2":
        lib_options(cas_unfused=1)
    P = R.synthetic_problem(N, 1300 + N)
    mol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    oo = R.OracleOOEnergy(mol, ncas, nelecas, P["oao_mo_coeff"])
    no = len(oo.occ_idx)
    C = oo.mo_coeff
    rng = np.random.default_rng(N)
    nrdm = 3
    gam = torch.stack([_rand(rng, ncas, ncas) for _ in range(nrdm)]).to(DEV).contiguous()
    Gam = torch.stack([_rand(rng, ncas, ncas, ncas, ncas) for _ in range(nrdm)]).to(DEV).contiguous()
    rows, cols = X.tril_tables(N, oo.params_idx)
    g_dev = oo.int2e_ao.to(DEV).contiguous()
    assert ops.eri_flags(g_dev) == ops.ERI_PQ_SYMMETRIC | ops.ERI_RS_SYMMETRIC
    args = (g_dev, oo.int1e_ao.to(DEV).contiguous(), C.to(DEV).contiguous(), gam, Gam, oo.nuc, no, ncas,
            torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV))
    full = ops.cas_eval(*args, want_matrices=True, want_integrals=True, eri_flags=0)
    sym = ops.cas_eval(*args, want_matrices=True, want_integrals=True, eri_flags=flags)
    for key in ("c0", "c1", "c2", "E", "gvec", "dE", "fock", "gmat", "Gm", "hmo"):
        a, b = full[key], sym[key]
        scale = max(1.0, float(a.abs().max()))
        assert (a - b).abs().max() <= 1e-12 * scale, key
    E_ref = oo.energy_from_mo_coeff(C, gam[0].cpu(), Gam[0].cpu())
    assert abs(sym["E"].item() - E_ref.item()) < 1e-8 * max(1.0, abs(E_ref.item()))
    gv_ref = oo.kappa_matrix_to_vector(oo.analytic_gradient(gam[0].cpu(), Gam[0].cpu()))
    assert (sym["gvec"].cpu()[0] - gv_ref).abs().max() < 1e-8 * max(1.0, float(gv_ref.abs().max()))


@pytest.mark.parametrize("N", [7, 16, 20, 33, 43, 48])
def test_eri_pack_layout(N):
    """oovqe_eri_pack: slab t = (p <= q) of the triangle; of each slab the upper triangle with the
    diagonal halved, row r holding its columns (r & ~1) .. N-1 (0 left of the diagonal in odd rows),
    rows back to back, the slab pitch rounded up to an even number of doubles with a zero pad element
    (every slab starts on a 16-byte boundary; the copy half_tri_kernel streams when both symmetry
    flags hold)."""
    from auto_oo_amd import _lib
    lib = _lib.load()
    G = 2
    rng = np.random.default_rng(N)
    g = rng.standard_normal((G, N, N, N, N))
    psz = lib.oovqe_eri_packed_size(N)
    ref = []
    odd = sum(N - (r & ~1) for r in range(N)) & 1
    for b in range(G):
        for p_ in range(N):
            for q_ in range(p_, N):
                for r_ in range(N):
                    e = r_ & ~1
                    row = g[b, p_, q_, r_, e:].copy()
                    if r_ > e:
                        row[0] = 0.0
                    row[r_ - e] *= 0.5
                    ref.append(row)
                if odd:
                    ref.append(np.zeros(1))
    ref = np.concatenate(ref)
    assert psz * G == ref.size
    gd = torch.tensor(g).to(DEV).contiguous()
    out = torch.full((G, psz), float("nan"), dtype=torch.float64, device=DEV)
    _lib.check(lib.oovqe_eri_pack(_lib.dptr(gd), N, G, _lib.dptr(out), _lib.stream_ptr()), "oovqe_eri_pack")
    assert np.array_equal(out.cpu().numpy().reshape(-1), ref)
    assert lib.oovqe_eri_packed_size(49) == 49 * 50 // 2 * 10 * 256      # (N > 48: the tile format, test_tile_packed_copy_layout)


@pytest.mark.parametrize("ncas,nelecas,ansatz", [(2, 2, "np_fabric"), (3, 4, "ucc"), (3, 2, "kupccd")])
def test_unrestricted_rdms_vs_oracle(ncas, nelecas, ansatz):
    """get_rdms(theta, restricted=False) (pqc.py:192-221 with the spin-orbital operators) against the
    oracle's dense Jordan-Wigner matrices; spin-summing them gives the restricted 1-RDM back."""
    import auto_oo_amd as aoo
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz=ansatz, n_layers=2, k=2)
    theta = torch.tensor(np.random.default_rng(ncas).uniform(0, 2 * np.pi, pqc.theta_shape))
    g1, g2 = pqc.get_rdms(theta, restricted=False)
    n = 2 * ncas
    assert g1.shape == (n, n) and g2.shape == (n, n, n, n)
    state = pqc.qnode(theta)
    r1, r2 = R.spin_rdms_from_state(state, ncas)
    assert (g1.cpu() - r1).abs().max() < 1e-12 and (g2.cpu() - r2).abs().max() < 1e-12
    s1, s2 = pqc.get_rdms_from_state(state, restricted=False)
    assert torch.equal(s1, g1) and torch.equal(s2, g2)
    gr, _ = pqc.get_rdms(theta)
    spin_summed = g1[0::2, 0::2] + g1[1::2, 1::2]
    assert (spin_summed - gr).abs().max() < 1e-12
    assert abs(float(torch.trace(g1)) - nelecas) < 1e-12
    # a complex state: Re[psi^T O psi] = re^T O re - im^T O im (bilinear, no conjugation)
    phase = np.exp(0.3j)
    c1, c2 = pqc.get_rdms_from_state(state * phase, restricted=False)
    o1, o2 = R.spin_rdms_from_state(state.cpu() * phase, ncas)
    assert (c1.cpu() - o1).abs().max() < 1e-12 and (c2.cpu() - o2).abs().max() < 1e-12


@pytest.mark.parametrize("A,K,J,B", [(1, 60, 100, 70000), (3, 72, 208, 24578), (2, 100, 81, 40000),
                                     (1, 200, 200, 66000), (40, 64, 128, 2048)])
def test_mode_contract_two_strip_kernel(A, K, J, B, lib_options):
    """INNER contractions large enough for contract_pair.hip (two 16-wide strips per wave: >= 5 tiles
    of J, even B, >= 2048 strips): against the einsum on sampled columns and, element for element,
    against the one-strip kernel (option k1_no_pair) -- 20- and 12-row K-chunks, J padded and not,
    a last 32-column block that is mostly empty, several slabs."""
    gen = torch.Generator(device=DEV).manual_seed(A * 7 + K)
    T = torch.randn((A, K, B), generator=gen, dtype=torch.float64, device=DEV)
    C = torch.randn((K, J), generator=gen, dtype=torch.float64, device=DEV)
    out = ops.mode_contract(T, C, A, K, J, B, last=False).reshape(A, J, B)
    lib_options(k1_no_pair=1)
    one = ops.mode_contract(T, C, A, K, J, B, last=False).reshape(A, J, B)
    assert (out - one).abs().max() <= 1e-12 * float(one.abs().max())
    for sl in (slice(0, 300), slice(B // 2 - 111, B // 2 + 200), slice(B - 300, B)):
        ref = torch.einsum("kj,akb->ajb", C, T[:, :, sl])
        assert (out[:, :, sl] - ref).abs().max() < 1e-11 * max(1.0, float(ref.abs().max()))


def test_mode_contract_strides_beyond_32_bit_row_offsets():
    """B = 34M columns: 16 rows of out are more than 4 GB apart, the k-steps of a chunk cannot be
    scalar offsets of one descriptor (contract_kernel<.., WIDE>: one descriptor per k-step, plain
    stores).  13.3 GB in, 4.4 GB out; checked against the einsum on column windows at both ends and
    in the middle."""
    A, K, J, B = 1, 49, 16, 34_000_000
    gen = torch.Generator(device=DEV).manual_seed(3)
    T = torch.empty((A, K, B), dtype=torch.float64, device=DEV)
    for k in range(K):                       # (filled row by row: randn of 13 GB at once doubles the peak)
        T[0, k] = torch.randn(B, generator=gen, dtype=torch.float64, device=DEV)
    C = torch.randn((K, J), generator=gen, dtype=torch.float64, device=DEV)
    out = ops.mode_contract(T, C, A, K, J, B, last=False).reshape(A, J, B)
    for sl in (slice(0, 4096), slice(B // 2 - 1000, B // 2 + 3000), slice(B - 4096, B)):
        # (a contiguous copy of the window: the library GEMM behind einsum returns wrong numbers for a
        # view whose rows are 272 MB apart)
        ref = torch.einsum("kj,akb->ajb", C, T[:, :, sl].contiguous())
        assert (out[:, :, sl] - ref).abs().max() < 1e-11 * max(1.0, float(ref.abs().max()))
    del T, out
    torch.cuda.empty_cache()


@pytest.mark.parametrize("A,K,J,B", [(1, 49, 16, 4000), (2, 100, 250, 40), (3, 60, 33, 1000), (1, 13, 13, 2197)])
def test_mode_contract_long_stride_kernels_on_small_shapes(A, K, J, B, lib_options):
    """The long-stride build of K1 (one descriptor per k-step, 64-bit store addresses), forced by
    option k1_force_wide on shapes whose einsum is cheap: every tile count class, ragged edges."""
    lib_options(k1_force_wide=1)
    rng = np.random.default_rng(A * 31 + K)
    T = _rand(rng, A, K, B)
    C = _rand(rng, K, J)
    ref = torch.einsum("kj,akb->ajb", C, T)
    out = ops.mode_contract(T.to(DEV), C.to(DEV), A, K, J, B, last=False).cpu().reshape(A, J, B)
    assert (out - ref).abs().max() < 1e-11 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("A,K,J,B", [(3, 43, 40, 1000), (3, 50, 100, 70000)])
def test_mode_contract_inner_slabs_do_not_leak(A, K, J, B):
    """K is padded to the kernel's chunk depth with zero rows of Cm; the padded k-steps of slab a must
    not read slab a + 1 (0 x Inf = NaN): an Inf / NaN planted in slab 1 stays in slab 1's results
    (one-strip and two-strip kernels)."""
    gen = torch.Generator(device=DEV).manual_seed(K)
    T = torch.randn((A, K, B), generator=gen, dtype=torch.float64, device=DEV)
    C = torch.randn((K, J), generator=gen, dtype=torch.float64, device=DEV)
    T[1, 0, :] = float("inf")
    T[1, 3, 5] = float("nan")
    out = ops.mode_contract(T, C, A, K, J, B, last=False).reshape(A, J, B)
    assert torch.isfinite(out[0]).all() and torch.isfinite(out[2]).all()
    ref0 = torch.einsum("kj,kb->jb", C, T[0])
    assert (out[0] - ref0).abs().max() < 1e-11 * max(1.0, float(ref0.abs().max()))
    assert not torch.isfinite(out[1]).all()


@pytest.mark.parametrize("A,K,J", [(100, 43, 40), (5000, 50, 100)])
def test_mode_contract_last_rows_do_not_leak(A, K, J):
    """LAST mode: the k-steps that pad K to the chunk depth would read the start of the next row of T;
    they are masked per lane, so an Inf at the start of row a + 1 does not reach row a."""
    gen = torch.Generator(device=DEV).manual_seed(K + 1)
    T = torch.randn((A, K), generator=gen, dtype=torch.float64, device=DEV)
    C = torch.randn((K, J), generator=gen, dtype=torch.float64, device=DEV)
    T[7, 0] = float("inf")
    T[40, 1] = float("nan")
    out = ops.mode_contract(T, C, A, K, J, 1, last=True).reshape(A, J)
    good = torch.ones(A, dtype=torch.bool, device=DEV)
    good[7] = good[40] = False
    assert torch.isfinite(out[good]).all()
    ref = T[good] @ C
    assert (out[good] - ref).abs().max() < 1e-11 * max(1.0, float(ref.abs().max()))


def test_mode_contract_random_shapes():
    """Forty seeded random shapes through oovqe_mode_contract (INNER and LAST; one-chunk, 12- and
    20-row chunk kernels; one-strip and two-strip kernels; ragged edges everywhere) against einsum."""
    rng = np.random.default_rng(20262)
    for trial in range(40):
        last = trial % 4 == 3
        K = int(rng.integers(1, 131))
        J = int(rng.integers(1, 231))
        if last:
            A, B = int(rng.integers(1, 4000)), 1
        else:
            A = int(rng.integers(1, 5))
            B = int(rng.choice([1, 2, 7, 16, 33, 250, 1000, 2999, 4096, 40002, 70000]))
        gen = torch.Generator(device=DEV).manual_seed(trial)
        T = torch.randn((A, K, B), generator=gen, dtype=torch.float64, device=DEV)
        C = torch.randn((K, J), generator=gen, dtype=torch.float64, device=DEV)
        out = ops.mode_contract(T, C, A, K, J, B, last=last).reshape(A, J, B)
        ref = torch.einsum("kj,akb->ajb", C, T)
        err = float((out - ref).abs().max())
        assert err < 1e-11 * max(1.0, float(ref.abs().max())), (trial, last, A, K, J, B, err)


@pytest.mark.parametrize("ncas,nelecas,batch", [(4, 4, 5), (4, 2, 40), (8, 8, 3), (8, 8, 130), (8, 6, 33)])
def test_sector_fused_kernels_equal_the_unfused_ones(ncas, nelecas, batch):
    """Round 3: RDMs and adjoint gradient with the E_pq vectors formed chunk by chunk in LDS
    (sector_rdm_fused_kernel / sector_w_fused_kernel, the default for a^2 = 16, 64) against the round-2
    kernels that write and re-read them (debug option sector_unfused) -- same sums in another order."""
    from auto_oo_amd.sector import SectorEngine
    n = 2 * ncas
    gates, n_theta = X.kupccd_gates(ncas, 1)
    hf = X.hf_state(nelecas, n)
    gd = _gates_dev(gates)
    eng = SectorEngine(ncas, hf, gd, len(gates), n_theta, X.basis_index(hf), torch.device(DEV))
    rng = np.random.default_rng(77 + ncas + batch)
    th = torch.tensor(rng.uniform(0, 2 * np.pi, (batch, n_theta))).to(DEV)
    c1 = torch.tensor(rng.standard_normal((ncas, ncas))).to(DEV)
    c2 = torch.tensor(rng.standard_normal((ncas,) * 4)).to(DEV)
    psi_c = eng.state(th)
    g1, g2 = eng.rdms(psi_c)
    dth = eng.adjoint(th, psi_c, c1, c2)
    from auto_oo_amd._lib import debug_options
    with debug_options(sector_unfused=1):
        h1, h2 = eng.rdms(psi_c)
        dth_u = eng.adjoint(th, psi_c, c1, c2)
    assert (g1 - h1).abs().max() < 1e-12 and (g2 - h2).abs().max() < 1e-12
    assert (dth - dth_u).abs().max() < 1e-11 * max(1.0, float(dth_u.abs().max()))
    # round 4: lambda in the string-driven form (G_a Psi + Psi G_b^T + the mixed term inside LDS; the default from
    # 32 states on) and through W = Ms^T V in memory (the default below), each forced, against the round-2 kernels
    for forced in (1, 2, 4):      # (4: the multiplier / helper-wave variant of the string-driven form, a^2 = 64 only)
        with debug_options(sector_lambda_w=forced):
            dth_f = eng.adjoint(th, psi_c, c1, c2)
        assert (dth_f - dth_u).abs().max() < 1e-11 * max(1.0, float(dth_u.abs().max())), forced
        # ... and the RDMs from chunks of 128 consecutive determinants (round 3; the default below 16 states) and
        # from chunks of whole alpha rows in the sigma basis (round 4)
        if forced == 4:
            continue
        with debug_options(sector_rdm_r3=forced):
            f1, f2 = eng.rdms(psi_c)
        assert (f1 - h1).abs().max() < 1e-12 and (f2 - h2).abs().max() < 1e-12, forced
    # the adjoint no longer depends on a preceding RDM call on the same workspace
    eng2 = SectorEngine(ncas, hf, gd, len(gates), n_theta, X.basis_index(hf), torch.device(DEV))
    assert torch.equal(eng2.adjoint(th, psi_c, c1, c2), dth)


@pytest.mark.parametrize("ncas,nelecas,ansatz", [(8, 8, "kupccd"), (4, 4, "kupccd"), (3, 4, "uccsd"), (6, 6, "uccsd")])
def test_sector_pair_lists_equal_the_gate_sweeps(ncas, nelecas, ansatz):
    """Round 4: the forward and the reverse sweep read every gate's determinant pairs from lists made once per
    circuit (oovqe_sector_pairs) instead of finding them per gate and per state.  The rotations are the same
    arithmetic on the same pairs: the state is bit-identical to the plain entry point's; the gradient's per-gate sums
    run in another order (1e-12)."""
    import ctypes
    from auto_oo_amd import _lib
    from auto_oo_amd._lib import dptr, stream_ptr, check
    from auto_oo_amd.sector import SectorEngine
    n = 2 * ncas
    gates, n_theta = X.kupccd_gates(ncas, 1) if ansatz == "kupccd" else X.uccd_gates(ncas, nelecas, True)
    hf = X.hf_state(nelecas, n)
    gd = _gates_dev(gates)
    eng = SectorEngine(ncas, hf, gd, len(gates), n_theta, X.basis_index(hf), torch.device(DEV))
    rng = np.random.default_rng(5 + ncas)
    batch = 3
    th = torch.tensor(rng.uniform(0, 2 * np.pi, (batch, n_theta))).to(DEV)
    c1 = torch.tensor(rng.standard_normal((ncas, ncas))).to(DEV)
    c2 = torch.tensor(rng.standard_normal((ncas,) * 4)).to(DEV)
    pairs, max_pairs = eng.pair_lists()
    counts = pairs[:len(gates)].cpu().numpy()
    assert max_pairs == counts.max() and 0 < max_pairs <= eng.Dc // 2
    psi = eng.state(th)
    dth = eng.adjoint(th, psi, c1, c2)
    lib = _lib.load()
    psi0 = torch.empty_like(psi)
    check(lib.oovqe_sector_state(dptr(th), n_theta, dptr(gd, torch.uint8), len(gates), ncas,
                                 ctypes.c_uint32(eng.init_index), *eng._tabs(), batch, dptr(psi0), None, stream_ptr()),
          "oovqe_sector_state")
    assert torch.equal(psi, psi0)
    dth0 = torch.empty_like(dth)
    check(lib.oovqe_sector_adjoint(dptr(th), n_theta, dptr(gd, torch.uint8), len(gates), ncas, *eng._tabs(), batch,
                                   dptr(psi0), dptr(c1), dptr(c2), dptr(eng.work(batch)), dptr(dth0), stream_ptr()),
          "oovqe_sector_adjoint")
    assert (dth - dth0).abs().max() < 1e-12 * max(1.0, float(dth0.abs().max()))
    # first / second tangent states (a differentiated gate also annihilates what it leaves alone)
    pg = eng.param_gates(gates)
    if pg is not None:
        specs = [(-1, -1), (pg[0], -1), (pg[n_theta - 1], -1), (pg[0], pg[0]), (pg[0], pg[n_theta - 1])]
        der = eng.derivative_states(th, specs)
        spec = torch.as_tensor(np.asarray(specs, dtype=np.int32)).to(DEV)
        der0 = torch.empty_like(der)
        check(lib.oovqe_sector_state_deriv(dptr(th), n_theta, dptr(gd, torch.uint8), len(gates), ncas,
                                           ctypes.c_uint32(eng.init_index), *eng._tabs(), batch,
                                           dptr(spec, torch.int32), len(specs), dptr(der0), stream_ptr()),
              "oovqe_sector_state_deriv")
        assert torch.equal(der, der0)
        assert torch.equal(der[:, 0], psi)


def test_batched_circuit_hessian_beyond_one_grid_of_pairs():
    """oovqe_circuit_hessian_batch puts (geometry, pair) into 16-bit grid dimensions; a stack with
    batch * 4 * n_pairs > 65535 (here 1 700 geometries x 40) goes through in chunks of geometries.  Every
    geometry's block equals the single-geometry call bit for bit."""
    import ctypes
    from auto_oo_amd import _lib, ops
    from auto_oo_amd._lib import dptr, stream_ptr, check
    lib = _lib.load()
    pqc = aoo.Parameterized_circuit(3, 4, None, ansatz="ucc")
    nt, a = int(pqc.theta_shape), 3
    G = 1700
    rng = np.random.default_rng(2)
    theta = torch.tensor(rng.uniform(0, 2 * np.pi, (G, nt))).cuda()
    c1 = torch.tensor(rng.standard_normal((G, a * a))).cuda()
    c2 = torch.tensor(rng.standard_normal((G, a ** 4))).cuda()
    pairs_dev, _, _ = ops._hessian_pair_tables(nt, theta.device)
    n_pairs = pairs_dev.shape[0]
    assert G * 4 * n_pairs > 65535
    work = torch.empty(G * lib.oovqe_circuit_hessian_work_size(nt, pqc.n_qubits, a, n_pairs), dtype=torch.float64,
                       device="cuda")
    H = torch.empty((G, nt, nt), dtype=torch.float64, device="cuda")
    check(lib.oovqe_circuit_hessian_batch(dptr(theta), nt, dptr(pqc._gates_dev, torch.uint8), pqc._n_gates,
                                          pqc.n_qubits, a, ctypes.c_uint32(pqc._init_index), dptr(c1), dptr(c2),
                                          dptr(pairs_dev, torch.int32), n_pairs, G, dptr(work), dptr(H), stream_ptr()),
          "oovqe_circuit_hessian_batch")
    for g in (0, 1, 408, 409, 410, 819, 1638, 1699):
        Hg = ops.circuit_hessian(theta[g], pqc._gates_dev, pqc._n_gates, pqc.n_qubits, a, pqc._init_index, c1[g], c2[g])
        assert torch.equal(H[g], Hg), g
