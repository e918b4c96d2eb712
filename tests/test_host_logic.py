"""Host-side integer logic (no GPU): excitation lists, gate tables, kappa index tables, and that
the C-ABI library loads and exports every symbol declared in include/oovqe.h."""
import json
import os
import re

import numpy as np
import pytest
import torch

from auto_oo_amd import _lib, excitations as X
from oracle import cpu_ref as R
from tests._emulate import apply_gate_table

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _load(name):
    with open(os.path.join(HERE, "golden", name)) as fh:
        return json.load(fh)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    with open(os.path.join(ROOT, "include", "oovqe.h")) as fh:
        hdr = fh.read()
    declared = set(re.findall(r"\b(oovqe_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in oovqe.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes prototype"
    assert lib.oovqe_version() >= 100


def test_gate_struct_layout():
    assert _lib.GATE_NBYTES == 40


def test_newton_step_block_layout():
    """The ctypes mirror of oovqe_newton_step_t (include/oovqe.h) against the header, field by field in order, and its
    size against the library's."""
    import ctypes
    with open(os.path.join(ROOT, "include", "oovqe.h")) as fh:
        hdr = fh.read()
    body = hdr[hdr.index("typedef struct oovqe_newton_step_t {"):hdr.index("} oovqe_newton_step_t;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for piece in decl.split(","):
            names.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", piece)[-1])
    assert names == [f[0] for f in _lib.NewtonStepT._fields_]
    assert _lib.load().oovqe_newton_step_size() == ctypes.sizeof(_lib.NewtonStepT)


@pytest.mark.parametrize("case", [c for c in _load("pqc_states.json") if c["ansatz"] == "ucc"],
                         ids=lambda c: c["source"])
def test_gate_table_reproduces_reference_states(case):
    """The closed-form Givens table (what the HIP kernel executes) against the reference's own
    statevectors (test/test_pqc.py:36-136)."""
    n = 2 * case["ncas"]
    gates, n_theta = X.uccd_gates(case["ncas"], case["nelecas"], bool(case["add_singles"]))
    assert n_theta == len(case["theta"])
    psi = apply_gate_table(gates, np.array(case["theta"]), n,
                           X.basis_index(X.hf_state(case["nelecas"], n)))
    ref = np.array(case["state_real"])
    tol = 5e-5 if case["source"].endswith(":36") else 1e-8
    assert np.abs(psi - ref).max() < tol


@pytest.mark.parametrize("ncas,nelecas,k", [(2, 2, 1), (3, 2, 2), (3, 4, 1), (4, 4, 1)])
def test_kupccd_table_matches_gate_level_oracle(ncas, nelecas, k):
    """kUpCCD has no reference fixture: pin the closed form against the gate-by-gate
    FermionicDoubleExcitation decomposition (incl. reversed wire pairs r > p)."""
    n = 2 * ncas
    gates, n_theta = X.kupccd_gates(ncas, k)
    theta = np.random.default_rng(7 + ncas).uniform(0, 2 * np.pi, n_theta)
    psi = apply_gate_table(gates, theta, n, X.basis_index(X.hf_state(nelecas, n)))
    ref = R.kupccd_state(torch.tensor(theta), ncas, nelecas, k).numpy()
    assert np.abs(ref.imag).max() < 1e-13
    assert np.abs(psi - ref.real).max() < 1e-12


@pytest.mark.parametrize("case", _load("nonredundant_idx.json"), ids=lambda c: c["source"])
def test_non_redundant_indices_golden(case):
    idx = X.non_redundant_indices(case["occ_idx"], case["act_idx"], case["virt_idx"],
                                  case["freeze_active"])
    assert np.array_equal(idx, np.array(case["idx_ref"]))


def test_non_redundant_indices_matches_oracle_loop():
    for no, na, nv, fr in [(6, 3, 4, False), (6, 3, 4, True), (0, 2, 3, False), (3, 4, 0, True)]:
        occ = list(range(no)); act = list(range(no, no + na)); virt = list(range(no + na, no + na + nv))
        assert np.array_equal(X.non_redundant_indices(occ, act, virt, fr),
                              R.non_redundant_indices(occ, act, virt, fr))


def test_tril_tables_follow_skew_packing():
    case = _load("skew_pack.json")[0]
    v = np.array(case["vector"]); m = np.array(case["matrix"])
    rows, cols = X.tril_tables(m.shape[0], np.arange(len(v)))
    assert np.array_equal(m[rows, cols], v)
    assert np.array_equal(m[cols, rows], -v)


def test_excitation_lists():
    s, d = X.excitations(4, 6)
    assert d == [[0, 1, 4, 5], [0, 3, 4, 5], [1, 2, 4, 5], [2, 3, 4, 5]]
    assert s == [[0, 4], [1, 5], [2, 4], [3, 5]]
    assert X.excitations(4, 6) == R.excitations(4, 6)
    assert X.excitations(6, 12) == R.excitations(6, 12)
    assert len(X.generalized_pair_doubles(range(16))) == 56


def test_synthetic_generator_matches_oracle_copy():
    from auto_oo_amd.synthetic import synthetic_problem
    a, b = synthetic_problem(9, 123), R.synthetic_problem(9, 123)
    for key in a:
        assert np.array_equal(np.asarray(a[key]), np.asarray(b[key]))


@pytest.mark.parametrize("case", [c for c in _load("pqc_states.json") if c["ansatz"] == "np_fabric"],
                         ids=lambda c: c["source"])
def test_gatefabric_table_reproduces_reference_states(case):
    """GateFabric (np_fabric) as entries of the ordinary Givens gate table against the reference's
    own statevectors (test/test_pqc.py:137-263)."""
    n = 2 * case["ncas"]
    gates, n_theta = X.gatefabric_gates(case["ncas"], case["nelecas"], case["n_layers"])
    assert n_theta == len(case["theta"])
    psi = apply_gate_table(gates, np.array(case["theta"]), n,
                           X.basis_index(X.hf_state(case["nelecas"], n)))
    assert np.abs(psi - np.array(case["state_real"])).max() < 1e-8


def test_gatefabric_redundant_idx_matches_oracle():
    for ncas, ne in [(2, 2), (3, 4), (3, 2), (4, 4), (5, 6), (6, 6)]:
        assert X.gatefabric_redundant_idx(ncas, ne) == R.gatefabric_redundant_idx(ncas, ne)


def test_moldata_npz_roundtrip_and_packed_eri(tmp_path):
    """Moldata.from_npz: full and 8-fold packed two-electron tensors."""
    from auto_oo_amd.moldata import Moldata, unpack_eri_s8
    P = R.synthetic_problem(6, 77)
    n = 6
    mol = Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 4,
                  mo_coeff=np.asarray(P["oao_mo_coeff"]))
    mol.save_npz(tmp_path / "m.npz")
    back = Moldata.from_npz(tmp_path / "m.npz")
    for key in ("int1e_ao", "int2e_ao", "overlap", "oao_coeff"):
        assert np.array_equal(getattr(back, key), getattr(mol, key)), key
    assert back.nuc == mol.nuc and back.nelectron == 4 and back.nao == n
    back.run_rhf()
    assert np.array_equal(back.hf.mo_coeff, np.asarray(P["oao_mo_coeff"]))
    # 8-fold packing: pairs (p>=q), then pairs of pairs (pq >= rs)
    g = np.asarray(P["int2e_ao"])
    pq = np.tril_indices(n)
    sq = g[pq[0], pq[1]][:, pq[0], pq[1]]
    packed = sq[np.tril_indices(sq.shape[0])]
    assert np.abs(unpack_eri_s8(packed, n) - g).max() < 1e-15   # g is symmetric to rounding only
    np.savez(tmp_path / "p.npz", int1e_ao=mol.int1e_ao, int2e_ao=packed, overlap=mol.overlap,
             nuc=mol.nuc, nelectron=4)
    assert np.abs(Moldata.from_npz(tmp_path / "p.npz").int2e_ao - g).max() < 1e-15
    with pytest.raises(KeyError):
        np.savez(tmp_path / "bad.npz", overlap=mol.overlap)
        Moldata.from_npz(tmp_path / "bad.npz")


def _newton_optimize(cost, x0, max_iterations, conv_tol, **kw):
    """Driver of the reference's Newton property tests (test/utils/test_newton_raphson.py:44-96):
    autodiff gradient/Hessian of `cost`, damped Newton steps until the energy stalls.  On CPU this
    pins the ORACLE's restatement of the Newton step (oracle/cpu_ref.py: OracleNewtonStep) with the
    reference's own property tests; the device-side NewtonStep runs the same properties in
    tests/test_newton_gpu.py and is compared with this oracle step by step there."""
    from torch.autograd.functional import jacobian, hessian
    opt = R.OracleNewtonStep(**kw)
    theta = x0
    energies = [cost(theta).item()]
    for n in range(max_iterations):
        grad = jacobian(cost, theta)
        hess = hessian(cost, theta)
        theta, _ = opt.damped_newton_step(cost, (theta,), grad, hess)
        energies.append(cost(theta).item())
        if n > 1 and abs(energies[-1] - energies[-2]) < conv_tol:
            break
    return energies, theta


@pytest.mark.parametrize("dim,max_iterations,conv_tol,lambda_min,rho,mu",
                         [(2, 20, 1e-12, 1e-6, 2, 1e-4), (4, 20, 1e-12, 1e-6, 2, 1e-4),
                          (8, 50, 1e-10, 1e-6, 3, 1e-4)])
def test_newton_diagonalises_symmetric_matrix(dim, max_iterations, conv_tol, lambda_min, rho, mu):
    """test_newton_raphson.py:99-116 ('type a'): minimise || U^T A U - diag(eig A) ||^2 over
    U = expm(-skew(x)); the augmented-Hessian Newton iteration must reach 0."""
    gen = torch.Generator().manual_seed(dim)
    a = torch.rand(dim, dim, generator=gen, dtype=torch.float64) - 0.5
    a = a.T + a
    va = torch.linalg.eigvalsh(a)

    def cost(x):
        u = torch.linalg.matrix_exp(-R.vector_to_skew_symmetric(x))
        return ((u.T @ a @ u - torch.diag(va)) ** 2).sum()
    x0 = 1e-5 * (torch.rand(dim * (dim - 1) // 2, generator=gen, dtype=torch.float64) - 0.5)
    energies, x = _newton_optimize(cost, x0, max_iterations, conv_tol, aug=True,
                                   lambda_min=lambda_min, rho=rho, mu=mu)
    assert abs(energies[-1]) < 1e-8
    u = torch.linalg.matrix_exp(-R.vector_to_skew_symmetric(x))
    assert torch.allclose(u.T @ a @ u, torch.diag(va), atol=1e-6)


@pytest.mark.parametrize("t,max_iterations", [(4.0, 10), (3.0, 10), (0.00004, 100)])
def test_newton_log_barrier_scalar(t, max_iterations):
    """test_newton_raphson.py:119-130 ('type b'): -t log|x| + |x| - t + t log t has its minimum 0
    at x = t; plain (non-augmented) damped Newton with backtracking from x = 10."""
    def cost(x):
        return (-t * torch.log(torch.abs(x)) + torch.abs(x) - t + t * np.log(t)).sum()
    energies, x = _newton_optimize(cost, torch.tensor([10.0], dtype=torch.float64), max_iterations,
                                   1e-12, aug=False)
    assert abs(energies[-1]) < 1e-8


def test_active_space_helpers_match_oracle():
    """active_space_integrals / molecular_hamiltonian_coefficients (API helpers on full MO tensors)."""
    from auto_oo_amd.active_space import active_space_integrals, molecular_hamiltonian_coefficients
    rng = np.random.default_rng(4)
    N = 7
    h = torch.tensor(rng.standard_normal((N, N)))
    g = torch.tensor(rng.standard_normal((N, N, N, N)))         # no symmetry assumed
    occ, act = [0, 1], [2, 3, 4]
    c0, c1, c2 = active_space_integrals(h, g, occ, act)
    r0, r1, r2 = R.active_space_integrals(h, g, occ, act)
    assert abs(c0.item() - r0.item()) < 1e-12 and (c1 - r1).abs().max() < 1e-13 and torch.equal(c2, r2)
    e0, e1, e2 = molecular_hamiltonian_coefficients(1.25, h, g, occ, act)
    s0, s1, s2 = R.molecular_hamiltonian_coefficients(1.25, h, g, occ, act)
    assert abs(e0.item() - s0.item()) < 1e-12 and (e1 - s1).abs().max() < 1e-13 and torch.equal(e2, s2)
    f0, f1, f2 = molecular_hamiltonian_coefficients(1.25, h, g)
    assert f0 == 1.25 and f1 is h and torch.equal(f2, 0.5 * g)


def test_packed_eri_size_and_synthetic_symmetry():
    """Host-side pieces of the symmetric-integral path (no GPU): oovqe_eri_packed_size follows the
    documented layout (slabs p <= q, row r = its columns (r & ~1) .. N-1, the slab pitch rounded
    up to an even number of doubles so that every slab starts on a 16-byte boundary), N > 48 has the
    tile-triangle form (slabs p <= q, 16 x 16 tiles R <= S), and the synthetic generator delivers bit-for-bit p<->q / r<->s symmetric integrals (what
    the device-side flag check relies on in bench.py and the tests)."""
    from auto_oo_amd import _lib
    from auto_oo_amd.synthetic import synthetic_problem
    lib = _lib.load()
    for N in (1, 2, 7, 16, 17, 32, 33, 43, 48):
        slab = sum(N - (r & ~1) for r in range(N))
        assert lib.oovqe_eri_packed_size(N) == N * (N + 1) // 2 * ((slab + 1) & ~1)
    assert lib.oovqe_eri_packed_size(43) == 946 * 968      # 967 stored elements + 1 pad
    assert lib.oovqe_eri_packed_size(0) == 0
    for N in (49, 64, 65, 200):
        nst = (N + 15) // 16
        assert lib.oovqe_eri_packed_size(N) == N * (N + 1) // 2 * (nst * (nst + 1) // 2) * 256
    assert lib.oovqe_eri_packed_size(200) == 20100 * 91 * 256
    g = synthetic_problem(13, 77)["int2e_ao"]
    assert np.array_equal(g, g.transpose(1, 0, 2, 3))
    assert np.array_equal(g, g.transpose(0, 1, 3, 2))


def test_get_formal_geo_tokens():
    """utils/miscellaneous.py:34-45: the Z-matrix rows (token for token; white space is free)."""
    import auto_oo_amd as aoo
    rows = [ln.split() for ln in aoo.get_formal_geo(140, 80).splitlines() if ln.strip()]
    assert rows == [["N"], ["C", "1", "1.498047"], ["H", "2", "1.066797", "1", "118.359375"],
                    ["H", "2", "1.066797", "1", "118.359375", "3", "180"],
                    ["H", "1", "0.987109", "2", "140", "3", "80"]]
    rows = [ln.split() for ln in aoo.get_formal_geo(120.5, 125).splitlines() if ln.strip()]
    assert rows[-1] == ["H", "1", "0.987109", "2", "120.5", "3", "125"]
