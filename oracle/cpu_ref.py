"""CPU oracle for the OO-VQE hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The shipped package ``auto_oo_amd`` never imports it and has no CPU
fallback: without the HIP library it raises.

What this is
------------
A plain PyTorch-CPU (fp64 / complex128) restatement of the reference algorithm
(Emieeel/auto_oo), following the cited reference lines op for op: the same four einsum
strings, ``torch.linalg.matrix_exp``, a gate-by-gate statevector simulation, a looped RDM
contraction with explicit Jordan-Wigner operators and ``torch.autograd.functional``
jacobians / hessians.  It is the checker the HIP path is compared against and the
``cpu_baseline`` ("port") timed by ``bench.py``.

Third-party arithmetic that is NOT under /root/reference (restated from the published
algorithms; PennyLane >=0.31.1 and OpenFermion, both unpinned in the reference's
pyproject.toml:20-28):
  * ``qml.qchem.excitations`` / ``excitations_to_wires`` / ``hf_state``
  * ``qml.FermionicDoubleExcitation`` / ``qml.FermionicSingleExcitation`` gate decompositions
  * ``qml.DoubleExcitation`` / ``qml.OrbitalRotation`` / ``qml.GateFabric``
  * ``openfermion.get_sparse_operator`` (Jordan-Wigner, mode j == qubit j, qubit 0 = MSB)

Parity pinning
--------------
  * circuits + RDMs: PINNED by the reference's own literal known-answer vectors
    (test/test_pqc.py::test_state, ::test_rdms -> tests/golden/pqc_*.json), checked in
    tests/test_oracle_goldens.py.
  * kappa packing / non-redundant index sets: PINNED (test/test_oo_energy.py:188-231).
  * integral transforms, CAS energy, Fock gradient, orbital Hessian: PARITY UNPINNED by
    reference fixtures in this environment (every reference test of them needs PySCF AO
    integrals; PySCF is absent).  They are pinned only by the same properties the reference
    tests assert (transform == naive einsum, analytic gradient == autodiff of the energy,
    analytic Hessian == autodiff Hessian) on synthetic 8-fold-symmetric integrals.
"""
import itertools

import numpy as np
import torch

DT = torch.float64
CDT = torch.complex128


# ----------------------------------------------------------------------------------------------
# a1/a2: integral transforms                                  reference: src/auto_oo/oo_energy.py
# ----------------------------------------------------------------------------------------------
def general_4index_transform(M, C0, C1, C2, C3):
    """oo_energy.py:21-30 -- four successive mode contractions (implicit einsum outputs are the
    alphabetically sorted free indices: iqrs, ijrs, ijks, ijkl)."""
    M = torch.einsum('pi,pqrs->iqrs', C0, M)
    M = torch.einsum('qj,iqrs->ijrs', C1, M)
    M = torch.einsum('rk,ijrs->ijks', C2, M)
    M = torch.einsum('sl,ijks->ijkl', C3, M)
    return M


def uniform_4index_transform(M, C):
    """oo_energy.py:33-41"""
    return general_4index_transform(M, C, C, C, C)


def int1e_transform(int1e_ao, mo_coeff):
    """oo_energy.py:44-46"""
    return mo_coeff.T @ int1e_ao @ mo_coeff


def int2e_transform(int2e_ao, mo_coeff):
    """oo_energy.py:49-51"""
    return uniform_4index_transform(int2e_ao, mo_coeff)


def mo_ao_to_mo_oao(mo_coeff, overlap):
    """oo_energy.py:54-60 (numpy)"""
    S_eigval, S_eigvec = np.linalg.eigh(overlap)
    S_half = S_eigvec @ np.diag(S_eigval ** 0.5) @ S_eigvec.T
    return S_half @ mo_coeff


def ao_to_oao(ovlp):
    """moldata_pyscf.py:13-16 (numpy)"""
    S_eigval, S_eigvec = np.linalg.eigh(ovlp)
    return S_eigvec @ np.diag(S_eigval ** (-0.5)) @ S_eigvec.T


# ----------------------------------------------------------------------------------------------
# a3/a4: kappa packing                                                    oo_energy.py:63-118
# ----------------------------------------------------------------------------------------------
def vector_to_skew_symmetric(vector):
    """oo_energy.py:63-87 -- strict lower triangle in np.tril_indices order, upper = -lower."""
    vector = torch.as_tensor(vector, dtype=DT)
    size = int(np.sqrt(8 * vector.shape[0] + 1) + 1) // 2
    matrix = torch.zeros((size, size), dtype=vector.dtype)
    tril = np.tril_indices(size, k=-1)
    # functional form of the two set_index calls (keeps autograd intact)
    matrix = matrix.index_put((torch.as_tensor(tril[0]), torch.as_tensor(tril[1])), vector)
    matrix = matrix.index_put((torch.as_tensor(tril[1]), torch.as_tensor(tril[0])), -vector)
    return matrix


def skew_symmetric_to_vector(kappa_matrix):
    """oo_energy.py:90-94"""
    size = kappa_matrix.shape[0]
    tril = np.tril_indices(size, k=-1)
    return kappa_matrix[tril[0], tril[1]]


def non_redundant_indices(occ_idx, act_idx, virt_idx, freeze_active):
    """oo_energy.py:97-118"""
    no, na, nv = len(occ_idx), len(act_idx), len(virt_idx)
    nao = no + na + nv
    rotation_sizes = [no * na, na * nv, no * nv]
    if not freeze_active:
        rotation_sizes.append(na * (na - 1) // 2)
    n_kappa = sum(rotation_sizes)
    params_idx = []
    num = 0
    occ_s, act_s, virt_s = set(occ_idx), set(act_idx), set(virt_idx)
    for l_idx, r_idx in zip(*np.tril_indices(nao, -1)):
        if not (((l_idx in act_s and r_idx in act_s) and freeze_active)
                or (l_idx in occ_s and r_idx in occ_s)
                or (l_idx in virt_s and r_idx in virt_s)):
            params_idx.append(num)
        num += 1
    assert n_kappa == len(params_idx)
    return np.array(params_idx, dtype=int)


# ----------------------------------------------------------------------------------------------
# a7: active-space Hamiltonian coefficients            src/auto_oo/utils/active_space.py:111-212
# ----------------------------------------------------------------------------------------------
def active_space_integrals(one_body_integrals, two_body_integrals, occ_idx, act_idx):
    """active_space.py:111-174 -- the advanced-indexing expressions are kept verbatim in meaning:
    ``[occ, occ, :, :][:, occ, occ]`` picks g[i,i,j,j]; ``[occ, :, :, occ][:, occ, occ]`` picks
    g[i,j,j,i]; ``[:, :, occ, occ][act][:, act]`` picks g[p,q,i,i]; ``[:, occ, occ, :]`` g[p,i,i,q]."""
    occ_idx = np.asarray(occ_idx)
    act_idx = np.asarray(act_idx)
    obai = np.ix_(*[act_idx] * 2)
    tbai = np.ix_(*[act_idx] * 4)
    core_constant = (
        2 * torch.sum(one_body_integrals[occ_idx, occ_idx])
        + 2 * torch.sum(two_body_integrals[occ_idx, occ_idx, :, :][:, occ_idx, occ_idx])
        - torch.sum(two_body_integrals[occ_idx, :, :, occ_idx][:, occ_idx, occ_idx])
    )
    as_two_body_integrals = two_body_integrals[tbai]
    as_one_body_integrals = (
        one_body_integrals[obai]
        + 2 * torch.sum(two_body_integrals[:, :, occ_idx, occ_idx][act_idx, :, :][:, act_idx, :],
                        dim=2)
        - torch.sum(two_body_integrals[:, occ_idx, occ_idx, :][act_idx, :, :][:, :, act_idx],
                    dim=1)
    )
    return core_constant, as_one_body_integrals, as_two_body_integrals


def molecular_hamiltonian_coefficients(nuclear_repulsion, one_body_integrals, two_body_integrals,
                                       occ_idx=None, act_idx=None):
    """active_space.py:177-212"""
    if occ_idx is None and act_idx is None:
        E_constant = nuclear_repulsion
    else:
        core_adjustment, one_body_integrals, two_body_integrals = active_space_integrals(
            one_body_integrals, two_body_integrals, occ_idx, act_idx)
        E_constant = core_adjustment + nuclear_repulsion
    return E_constant, one_body_integrals, 0.5 * two_body_integrals


# ----------------------------------------------------------------------------------------------
# Molecule stand-in (moldata_pyscf.py:19-56 without PySCF: arrays in, same attributes out)
# ----------------------------------------------------------------------------------------------
class OracleMol:
    def __init__(self, int1e_ao, int2e_ao, overlap, nuc, nelectron, mo_coeff0=None):
        self.int1e_ao = np.asarray(int1e_ao)
        self.int2e_ao = np.asarray(int2e_ao)
        self.overlap = np.asarray(overlap)
        self.oao_coeff = ao_to_oao(self.overlap)
        self.nuc = float(nuc)
        self.nao = self.overlap.shape[0]
        self.nelectron = int(nelectron)
        self.mo_coeff0 = mo_coeff0

    def get_active_space_idx(self, ncas, nelecas):
        """moldata_pyscf.py:42-56"""
        nelecore = self.nelectron - nelecas
        if nelecore % 2 == 1:
            raise ValueError('odd number of core electrons')
        occ_idx = np.arange(nelecore // 2)
        act_idx = (occ_idx[-1] + 1 + np.arange(ncas) if len(occ_idx) > 0 else np.arange(ncas))
        virt_idx = np.arange(act_idx[-1] + 1, self.nao)
        return occ_idx, act_idx, virt_idx


# ----------------------------------------------------------------------------------------------
# a5-a8, a14, a15: OO_energy                                              oo_energy.py:121-424
# ----------------------------------------------------------------------------------------------
class OracleOOEnergy:
    def __init__(self, mol, ncas, nelecas, oao_mo_coeff, freeze_active=False):
        """oo_energy.py:125-171 (oao_mo_coeff must be given: no RHF without PySCF)."""
        self.oao_mo_coeff = torch.as_tensor(np.asarray(oao_mo_coeff), dtype=DT)
        self.int1e_ao = torch.as_tensor(mol.int1e_ao, dtype=DT)
        self.int2e_ao = torch.as_tensor(mol.int2e_ao, dtype=DT)
        self.overlap = mol.overlap
        self.oao_coeff = torch.as_tensor(mol.oao_coeff, dtype=DT)
        self.nuc = mol.nuc
        self.nao = mol.nao
        self.ncas = ncas
        self.nelecas = nelecas
        self.occ_idx, self.act_idx, self.virt_idx = mol.get_active_space_idx(ncas, nelecas)
        self.params_idx = non_redundant_indices(self.occ_idx, self.act_idx, self.virt_idx,
                                                freeze_active)
        self.n_kappa = len(self.params_idx)

    @property
    def mo_coeff(self):
        """oo_energy.py:173-176"""
        return self.oao_coeff @ self.oao_mo_coeff

    def get_active_integrals(self, mo_coeff):
        """oo_energy.py:204-211"""
        int1e_mo = int1e_transform(self.int1e_ao, mo_coeff)
        int2e_mo = int2e_transform(self.int2e_ao, mo_coeff)
        return molecular_hamiltonian_coefficients(self.nuc, int1e_mo, int2e_mo,
                                                  self.occ_idx, self.act_idx)

    def energy_from_mo_coeff(self, mo_coeff, one_rdm, two_rdm):
        """oo_energy.py:178-197"""
        c0, c1, c2 = self.get_active_integrals(mo_coeff)
        return c0 + torch.einsum('pq,pq', c1, one_rdm) + torch.einsum('pqrs,pqrs', c2, two_rdm)

    def energy_from_kappa(self, kappa, one_rdm, two_rdm):
        """oo_energy.py:199-202"""
        mo_coeff = self.mo_coeff @ self.kappa_to_mo_coeff(kappa)
        return self.energy_from_mo_coeff(mo_coeff, one_rdm, two_rdm)

    def kappa_vector_to_matrix(self, kappa):
        """oo_energy.py:213-219"""
        total = torch.zeros(self.nao * (self.nao - 1) // 2, dtype=kappa.dtype)
        total = total.index_put((torch.as_tensor(self.params_idx),), kappa)
        return vector_to_skew_symmetric(total)

    def kappa_matrix_to_vector(self, kappa_matrix):
        """oo_energy.py:221-224"""
        return skew_symmetric_to_vector(kappa_matrix)[self.params_idx]

    def kappa_to_mo_coeff(self, kappa):
        """oo_energy.py:226-230 -- expm(-K)"""
        return torch.linalg.matrix_exp(-self.kappa_vector_to_matrix(kappa))

    def get_transformed_mo(self, mo_coeff, kappa):
        """oo_energy.py:232-236"""
        return mo_coeff @ self.kappa_to_mo_coeff(kappa)

    # --- Fock matrices / gradient -------------------------------------------------------------
    def fock_core(self, int1e_mo, int2e_mo):
        """oo_energy.py:272-284"""
        occ = self.occ_idx
        g_tilde = (2 * torch.sum(int2e_mo[:, :, occ, occ], dim=2)
                   - torch.sum(int2e_mo[:, occ, occ, :], dim=1))
        return int1e_mo + g_tilde

    def fock_active(self, int2e_mo, one_rdm):
        """oo_energy.py:286-298"""
        act = self.act_idx
        g_tilde = (int2e_mo[:, :, :, act][:, :, act, :]
                   - 0.5 * int2e_mo[:, :, act, :][:, act, :, :].permute(0, 3, 2, 1))
        return torch.einsum('vw,mnvw->mn', one_rdm, g_tilde)

    def fock_generalized(self, int1e_mo, int2e_mo, one_rdm, two_rdm):
        """oo_energy.py:238-270"""
        occ, act = self.occ_idx, self.act_idx
        fock_C = self.fock_core(int1e_mo, int2e_mo)
        fock_A = self.fock_active(int2e_mo, one_rdm)
        rows_occ = 2 * (fock_C[:, occ] + fock_A[:, occ]).T
        rows_act = (torch.einsum('nw,vw->vn', fock_C[:, act], one_rdm)
                    + torch.einsum('vwxy,nwxy->vn', two_rdm,
                                   int2e_mo[:, :, :, act][:, :, act, :][:, act, :, :]))
        fock_general = torch.zeros_like(int1e_mo)
        fock_general = fock_general.index_put((torch.as_tensor(occ),), rows_occ)
        fock_general = fock_general.index_put((torch.as_tensor(act),), rows_act)
        return fock_general

    def analytic_gradient_from_integrals(self, int1e_mo, int2e_mo, one_rdm, two_rdm):
        """oo_energy.py:300-309"""
        F = self.fock_generalized(int1e_mo, int2e_mo, one_rdm, two_rdm)
        return 2 * (F - F.T)

    def analytic_gradient(self, one_rdm, two_rdm, mo_coeff=None):
        """oo_energy.py:404-413"""
        if mo_coeff is None:
            mo_coeff = self.mo_coeff
        int1e_mo = int1e_transform(self.int1e_ao, mo_coeff)
        int2e_mo = int2e_transform(self.int2e_ao, mo_coeff)
        return self.analytic_gradient_from_integrals(int1e_mo, int2e_mo, one_rdm, two_rdm)

    # --- Hessian ------------------------------------------------------------------------------
    def full_rdms(self, one_rdm, two_rdm):
        """oo_energy.py:342-379"""
        occ, act, n = self.occ_idx, self.act_idx, self.nao
        no = len(occ)
        one_full = torch.zeros((n, n), dtype=DT)
        two_full = torch.zeros((n, n, n, n), dtype=DT)
        eye = torch.eye(no, dtype=DT)
        one_full[occ, occ] = 2 * torch.ones(no, dtype=DT)
        one_full[np.ix_(act, act)] = one_rdm
        two_full[np.ix_(*[occ] * 4)] = (4 * torch.einsum('ij,kl->ijkl', eye, eye)
                                        - 2 * torch.einsum('il,jk->ijkl', eye, eye))
        two_full[np.ix_(occ, occ, act, act)] = 2 * torch.einsum('wv,ij->ijwv', one_rdm, eye)
        two_full[np.ix_(act, act, occ, occ)] = 2 * torch.einsum('wv,ij->wvij', one_rdm, eye)
        two_full[np.ix_(occ, act, act, occ)] = -torch.einsum('wv,ij->iwvj', one_rdm, eye)
        two_full[np.ix_(act, occ, occ, act)] = -torch.einsum('wv,ij->vjiw', one_rdm, eye)
        two_full[np.ix_(*[act] * 4)] = two_rdm
        return one_full, two_full

    def y_matrix(self, int2e_mo, two_full):
        """oo_energy.py:381-393 (dense N^6 einsums, as in the reference)"""
        y0 = torch.einsum('pmrn,qmns->pqrs', two_full, int2e_mo)
        y1 = torch.einsum('pmnr,qmns->pqrs', two_full, int2e_mo)
        y2 = torch.einsum('prmn,qsmn->pqrs', two_full, int2e_mo)
        return y0 + y1 + y2

    def analytic_hessian_from_integrals(self, int1e_mo, int2e_mo, one_rdm, two_rdm):
        """oo_energy.py:311-340"""
        one_full, two_full = self.full_rdms(one_rdm, two_rdm)
        y = self.y_matrix(int2e_mo, two_full)
        F = self.fock_generalized(int1e_mo, int2e_mo, one_rdm, two_rdm)
        Fs = F + F.T
        hess0 = 2 * torch.einsum('pr,qs->pqrs', one_full, int1e_mo)
        hess1 = -torch.einsum('pr,qs->pqrs', Fs, torch.eye(self.nao, dtype=DT))
        h0 = hess0 + hess1 + 2 * y
        return (h0 - h0.permute(0, 1, 3, 2) - h0.permute(1, 0, 2, 3) + h0.permute(1, 0, 3, 2))

    def analytic_hessian(self, one_rdm, two_rdm, mo_coeff=None):
        """oo_energy.py:415-424"""
        if mo_coeff is None:
            mo_coeff = self.mo_coeff
        int1e_mo = int1e_transform(self.int1e_ao, mo_coeff)
        int2e_mo = int2e_transform(self.int2e_ao, mo_coeff)
        return self.analytic_hessian_from_integrals(int1e_mo, int2e_mo, one_rdm, two_rdm)

    def full_hessian_to_matrix(self, full_hess):
        """oo_energy.py:395-402"""
        tril = np.tril_indices(self.nao, k=-1)
        partial = full_hess[tril[0], tril[1], :, :]
        reduced = partial[:, tril[0], tril[1]]
        return reduced[self.params_idx, :][:, self.params_idx]


# ----------------------------------------------------------------------------------------------
# PennyLane qchem helpers (published algorithm, restated)
# ----------------------------------------------------------------------------------------------
def excitations(electrons, orbitals, delta_sz=0):
    """qml.qchem.excitations: interleaved spins (even wire = alpha, sz=+1/2)."""
    sz = np.array([0.5 if (i % 2 == 0) else -0.5 for i in range(orbitals)])
    singles = [[r, p] for r in range(electrons) for p in range(electrons, orbitals)
               if sz[p] - sz[r] == delta_sz]
    doubles = [[s, r, q, p]
               for s in range(electrons - 1) for r in range(s + 1, electrons)
               for q in range(electrons, orbitals - 1) for p in range(q + 1, orbitals)
               if (sz[p] + sz[q] - sz[r] - sz[s]) == delta_sz]
    return singles, doubles


def excitations_to_wires(singles, doubles):
    """qml.qchem.excitations_to_wires"""
    s_wires = [list(range(r, p + 1)) for r, p in singles]
    d_wires = [[list(range(s, r + 1)), list(range(q, p + 1))] for s, r, q, p in doubles]
    return s_wires, d_wires


def hf_state(electrons, orbitals):
    """qml.qchem.hf_state"""
    return np.array([1 if i < electrons else 0 for i in range(orbitals)], dtype=int)


def generalized_pair_doubles(wires):
    """ansatze/kUpCCD.py:16-33"""
    wires = list(wires)
    return [[wires[r:r + 2], wires[p:p + 2]]
            for r in range(0, len(wires) - 1, 2)
            for p in range(0, len(wires) - 1, 2) if p != r]


# ----------------------------------------------------------------------------------------------
# Gate-by-gate statevector simulator (default.qubit semantics: wire 0 = most significant bit)
# ----------------------------------------------------------------------------------------------
_SQ2 = 1.0 / np.sqrt(2.0)
_H = torch.tensor([[_SQ2, _SQ2], [_SQ2, -_SQ2]], dtype=CDT)


def _rx(phi):
    phi = torch.as_tensor(phi, dtype=DT)
    c, s = torch.cos(phi / 2).to(CDT), torch.sin(phi / 2).to(CDT)
    return torch.stack([torch.stack([c, -1j * s]), torch.stack([-1j * s, c])])


def _rz(phi):
    phi = torch.as_tensor(phi, dtype=DT).to(CDT)
    z = torch.zeros((), dtype=CDT)
    return torch.stack([torch.stack([torch.exp(-0.5j * phi), z]),
                        torch.stack([z, torch.exp(0.5j * phi)])])


class Statevector:
    """State kept as a [2]*n tensor; every gate is a handful of torch ops (autograd-capable)."""

    def __init__(self, n, basis_state):
        self.n = n
        idx = int(''.join(str(int(b)) for b in basis_state), 2)
        psi = torch.zeros(2 ** n, dtype=CDT)
        psi[idx] = 1.0
        self.psi = psi.reshape([2] * n)

    def apply_1q(self, U, w):
        self.psi = torch.movedim(torch.tensordot(U, self.psi, dims=([1], [w])), 0, w)

    def cnot(self, c, t):
        s0 = self.psi.select(c, 0)
        s1 = self.psi.select(c, 1)
        tt = t if t < c else t - 1
        self.psi = torch.stack([s0, s1.flip(tt)], dim=c)

    def apply_2q(self, U4, w0, w1):
        """U4 indexed [out0,out1,in0,in1] as a (2,2,2,2) tensor."""
        res = torch.tensordot(U4, self.psi, dims=([2, 3], [w0, w1]))
        self.psi = torch.movedim(res, [0, 1], [w0, w1])

    def vector(self):
        return self.psi.reshape(-1)


def _fde_cnot_wires(wires1, wires2):
    cn = [wires1[l:l + 2] for l in range(len(wires1) - 1)]
    cn += [[wires1[-1], wires2[0]]]
    cn += [wires2[l:l + 2] for l in range(len(wires2) - 1)]
    return cn


# (basis on s, r, q, p ; sign of RZ angle) for the 8 layers of FermionicDoubleExcitation
_FDE_LAYERS = [("HHXH", +1), ("XHXX", +1), ("HXXX", +1), ("HHHX", +1),
               ("XHHH", -1), ("HXHH", -1), ("XXXH", -1), ("XXHX", -1)]


def fermionic_double_excitation(sv, weight, wires1, wires2):
    """qml.FermionicDoubleExcitation decomposition (8 layers: basis change, CNOT ladder,
    RZ(+-weight/8) on p, reversed ladder, inverse basis change)."""
    s, r, q, p = wires1[0], wires1[-1], wires2[0], wires2[-1]
    cn = _fde_cnot_wires(list(wires1), list(wires2))
    for basis, sign in _FDE_LAYERS:
        for b, w in zip(basis, (s, r, q, p)):
            sv.apply_1q(_H if b == "H" else _rx(-np.pi / 2), w)
        for c, t in cn:
            sv.cnot(c, t)
        sv.apply_1q(_rz(sign * weight / 8), p)
        for c, t in reversed(cn):
            sv.cnot(c, t)
        for b, w in zip(basis, (s, r, q, p)):
            sv.apply_1q(_H if b == "H" else _rx(np.pi / 2), w)


def fermionic_single_excitation(sv, weight, wires):
    """qml.FermionicSingleExcitation decomposition (two layers)."""
    wires = list(wires)
    r, p = wires[0], wires[-1]
    cn = [wires[l:l + 2] for l in range(len(wires) - 1)]
    for (br, bp, sign) in (("X", "H", +1), ("H", "X", -1)):
        sv.apply_1q(_rx(-np.pi / 2) if br == "X" else _H, r)
        sv.apply_1q(_rx(-np.pi / 2) if bp == "X" else _H, p)
        for c, t in cn:
            sv.cnot(c, t)
        sv.apply_1q(_rz(sign * weight / 2), p)
        for c, t in reversed(cn):
            sv.cnot(c, t)
        sv.apply_1q(_rx(np.pi / 2) if br == "X" else _H, r)
        sv.apply_1q(_rx(np.pi / 2) if bp == "X" else _H, p)


def _givens_2q(phi):
    """qml.SingleExcitation matrix on (w0,w1): |01>->c|01>+s|10>, |10>->c|10>-s|01>, with
    basis order |w0 w1>; c,s = cos,sin(phi/2)."""
    phi = torch.as_tensor(phi, dtype=DT)
    c, s = torch.cos(phi / 2).to(CDT), torch.sin(phi / 2).to(CDT)
    one, zero = torch.ones((), dtype=CDT), torch.zeros((), dtype=CDT)
    rows = [torch.stack([one, zero, zero, zero]),
            torch.stack([zero, c, -s, zero]),
            torch.stack([zero, s, c, zero]),
            torch.stack([zero, zero, zero, one])]
    return torch.stack(rows).reshape(2, 2, 2, 2)


_FSWAP = torch.tensor([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, -1]],
                      dtype=CDT).reshape(2, 2, 2, 2)


def double_excitation(sv, phi, wires):
    """qml.DoubleExcitation on 4 wires: |0011> -> c|0011> + s|1100>, |1100> -> c|1100> - s|0011>."""
    phi = torch.as_tensor(phi, dtype=DT)
    c, s = torch.cos(phi / 2).to(CDT), torch.sin(phi / 2).to(CDT)
    U = torch.eye(16, dtype=CDT)
    e = torch.zeros((16, 16), dtype=CDT)
    def unit(i, j):
        m = torch.zeros((16, 16), dtype=CDT)
        m[i, j] = 1.0
        return m
    U = (U - unit(3, 3) - unit(12, 12)
         + c * (unit(3, 3) + unit(12, 12)) + s * unit(12, 3) - s * unit(3, 12))
    del e
    U = U.reshape([2] * 8)
    res = torch.tensordot(U, sv.psi, dims=([4, 5, 6, 7], list(wires)))
    sv.psi = torch.movedim(res, [0, 1, 2, 3], list(wires))


def orbital_rotation(sv, phi, wires):
    """qml.OrbitalRotation = fSWAP[q1,q2] SingleExcitation[q0,q1] SingleExcitation[q2,q3] fSWAP[q1,q2]."""
    q0, q1, q2, q3 = wires
    sv.apply_2q(_FSWAP, q1, q2)
    sv.apply_2q(_givens_2q(phi), q0, q1)
    sv.apply_2q(_givens_2q(phi), q2, q3)
    sv.apply_2q(_FSWAP, q1, q2)


# ----------------------------------------------------------------------------------------------
# a9/a10: ansatz states                               pqc.py:69-83,121-186; ansatze/*.py
# ----------------------------------------------------------------------------------------------
def uccd_state(theta, ncas, nelecas, add_singles=False):
    """pqc.py:69-76,121-134,165-172 + ansatze/uccd.py:105-114 (UCCD) / qml.UCCSD (doubles first
    with weights[len(s_wires)+i], then singles with weights[j])."""
    n = 2 * ncas
    singles, doubles = excitations(nelecas, n)
    s_wires, d_wires = excitations_to_wires(singles, doubles)
    sv = Statevector(n, hf_state(nelecas, n))
    if add_singles:
        for i, (w1, w2) in enumerate(d_wires):
            fermionic_double_excitation(sv, theta[len(s_wires) + i], w1, w2)
        for j, w in enumerate(s_wires):
            fermionic_single_excitation(sv, theta[j], w)
    else:
        for i, (w1, w2) in enumerate(d_wires):
            fermionic_double_excitation(sv, theta[i], w1, w2)
    return sv.vector()


def kupccd_state(theta, ncas, nelecas, k=1):
    """ansatze/kUpCCD.py:94-130: k layers over generalized_pair_doubles on |HF>."""
    n = 2 * ncas
    d_wires = generalized_pair_doubles(range(n))
    theta = theta.reshape(k, len(d_wires))
    sv = Statevector(n, hf_state(nelecas, n))
    for layer in range(k):
        for i, (w1, w2) in enumerate(d_wires):
            fermionic_double_excitation(sv, theta[layer][i], w1, w2)
    return sv.vector()


def gatefabric_redundant_idx(ncas, nelecas):
    """pqc.py:147-153"""
    n_qubits = 2 * ncas
    if n_qubits > 4:
        red = [x for x in range(0, 2 * (nelecas // 4))]
        if ncas % 2 == 0:
            red += [x for x in range(2 * ((n_qubits - nelecas) // 4), 2 * (n_qubits // 4))]
    else:
        red = []
    return red


def gatefabric_state(theta, ncas, nelecas, n_layers):
    """pqc.py:79-83,136-160,174-186 + qml.GateFabric(include_pi=False)."""
    n = 2 * ncas
    full_shape = (n_layers, n // 2 - 1, 2)
    red = gatefabric_redundant_idx(ncas, nelecas)
    n_full = int(np.prod(full_shape))
    params_idx = [x for x in range(n_full) if x not in red]
    theta_full = torch.zeros(n_full, dtype=theta.dtype)
    theta_full = theta_full.index_put((torch.as_tensor(params_idx),), theta)
    theta_full = theta_full.reshape(full_shape)
    wires = list(range(n))
    blocks = [wires[i:i + 4] for i in range(0, n, 4) if i + 4 <= n]
    blocks += [wires[i:i + 4] for i in range(2, n, 4) if i + 4 <= n]
    sv = Statevector(n, hf_state(nelecas, n))
    for layer in range(n_layers):
        for i, bw in enumerate(blocks):
            double_excitation(sv, theta_full[layer, i, 0], bw)
            orbital_rotation(sv, theta_full[layer, i, 1], bw)
    return sv.vector()


# ----------------------------------------------------------------------------------------------
# a11: RDMs with explicit Jordan-Wigner operators     pqc.py:22-66,192-235; active_space.py:29-83
# ----------------------------------------------------------------------------------------------
def _jw_annihilators(n):
    """openfermion.get_sparse_operator convention: a_j = Z^{(x)j} (x) [[0,1],[0,0]] (x) I, qubit 0 MSB.
    Returns dense numpy matrices (fine up to ~10 qubits; the oracle is for small cases)."""
    import scipy.sparse as sp
    Z = sp.csr_matrix(np.array([[1.0, 0.0], [0.0, -1.0]]))
    I2 = sp.identity(2, format='csr')
    low = sp.csr_matrix(np.array([[0.0, 1.0], [0.0, 0.0]]))
    ops = []
    for j in range(n):
        m = sp.identity(1, format='csr')
        for k in range(n):
            m = sp.kron(m, Z if k < j else (low if k == j else I2), format='csr')
        ops.append(m)
    return ops


class RdmOperators:
    """E_pq = a+_{2p} a_{2q} + a+_{2p+1} a_{2q+1} (active_space.py:29-53, up_then_down=False) and
    e_pqrs = E_pq E_rs - delta_qr E_ps (active_space.py:56-83) as torch sparse matrices."""

    def __init__(self, ncas):
        self.ncas = ncas
        n = 2 * ncas
        a = _jw_annihilators(n)
        ad = [m.T.tocsr() for m in a]
        self.E = [[(ad[2 * p] @ a[2 * q] + ad[2 * p + 1] @ a[2 * q + 1]).tocsr()
                   for q in range(ncas)] for p in range(ncas)]
        self._torch_E = None
        self._torch_e2 = None

    @staticmethod
    def _to_torch(m):
        m = m.tocoo()
        idx = torch.tensor(np.vstack([m.row, m.col]), dtype=torch.int64)
        return torch.sparse_coo_tensor(idx, torch.tensor(m.data, dtype=CDT), m.shape).coalesce()

    def torch_ops(self):
        if self._torch_E is None:
            nc = self.ncas
            self._torch_E = [[self._to_torch(self.E[p][q]) for q in range(nc)] for p in range(nc)]
            self._torch_e2 = {}
            for p, q, r, s in itertools.product(range(nc), repeat=4):
                m = self.E[p][q] @ self.E[r][s]
                if q == r:
                    m = m - self.E[p][s]
                self._torch_e2[(p, q, r, s)] = self._to_torch(m)
        return self._torch_E, self._torch_e2


def spin_rdms_from_state(state, ncas):
    """pqc.py:192-218 with restricted=False: gamma_pq = Re[state @ (a+_p a_q @ state)],
    Gamma_pqrs = Re[state @ (a+_p a+_q a_r a_s @ state)] over the 2 ncas spin orbitals
    (active_space.py:44-52,79-82), dense Jordan-Wigner matrices (small registers only)."""
    n = 2 * ncas
    a = [m.toarray() for m in _jw_annihilators(n)]
    ad = [m.T for m in a]
    psi = np.asarray(state.detach().cpu() if isinstance(state, torch.Tensor) else state).astype(np.complex128)
    one = np.zeros((n, n))
    two = np.zeros((n, n, n, n))
    kets = [a[s] @ psi for s in range(n)]
    for p, q in itertools.product(range(n), repeat=2):
        one[p, q] = (psi @ (ad[p] @ kets[q])).real
    for r, s_ in itertools.product(range(n), repeat=2):
        v = a[r] @ kets[s_]
        for p, q in itertools.product(range(n), repeat=2):
            two[p, q, r, s_] = (psi @ (ad[p] @ (ad[q] @ v))).real
    return torch.tensor(one), torch.tensor(two)


def rdms_from_state(state, ops: RdmOperators):
    """pqc.py:192-218 -- bilinear form state @ (E @ state) (no conjugation), then .real."""
    nc = ops.ncas
    E, e2 = ops.torch_ops()
    state = state.to(CDT)
    col = state.reshape(-1, 1)
    one = []
    two = []
    for p, q in itertools.product(range(nc), repeat=2):
        one.append((state @ torch.sparse.mm(E[p][q], col).reshape(-1)).real)
        for r, s in itertools.product(range(nc), repeat=2):
            two.append((state @ torch.sparse.mm(e2[(p, q, r, s)], col).reshape(-1)).real)
    return torch.stack(one).reshape(nc, nc), torch.stack(two).reshape(nc, nc, nc, nc)


# ----------------------------------------------------------------------------------------------
# a12, a13, a16: OO_pqc composition                                          oo_pqc.py:30-148
# ----------------------------------------------------------------------------------------------
class OraclePQC:
    """Parameterized_circuit stand-in (pqc.py:86-235) for ansatz in {'ucc', 'np_fabric', 'kupccd'}."""

    def __init__(self, ncas, nelecas, ansatz='ucc', n_layers=3, add_singles=False, k=1):
        self.ncas, self.nelecas = ncas, nelecas
        self.n_qubits = 2 * ncas
        self.ansatz, self.n_layers, self.add_singles, self.k = ansatz, n_layers, add_singles, k
        self.ops = None
        if ansatz == 'ucc':
            self.singles, self.doubles = excitations(nelecas, self.n_qubits)
            self.theta_shape = len(self.doubles) + (len(self.singles) if add_singles else 0)
        elif ansatz == 'np_fabric':
            full = n_layers * (self.n_qubits // 2 - 1) * 2
            self.theta_shape = full - len(gatefabric_redundant_idx(ncas, nelecas))
        elif ansatz == 'kupccd':
            self.theta_shape = k * len(generalized_pair_doubles(range(self.n_qubits)))
        else:
            raise ValueError(ansatz)

    def qnode(self, theta):
        if self.ansatz == 'ucc':
            return uccd_state(theta, self.ncas, self.nelecas, self.add_singles)
        if self.ansatz == 'np_fabric':
            return gatefabric_state(theta, self.ncas, self.nelecas, self.n_layers)
        return kupccd_state(theta, self.ncas, self.nelecas, self.k)

    def get_rdms_from_state(self, state):
        if self.ops is None:
            self.ops = RdmOperators(self.ncas)
        return rdms_from_state(state, self.ops)

    def get_rdms(self, theta):
        return self.get_rdms_from_state(self.qnode(theta))


class OracleOOPQC(OracleOOEnergy):
    def __init__(self, pqc, mol, ncas, nelecas, oao_mo_coeff, freeze_active=False):
        super().__init__(mol, ncas, nelecas, oao_mo_coeff, freeze_active)
        self.pqc = pqc

    def energy_from_parameters(self, theta, kappa=None):
        """oo_pqc.py:64-84"""
        mo_coeff = self.mo_coeff if kappa is None else self.get_transformed_mo(self.mo_coeff, kappa)
        one_rdm, two_rdm = self.pqc.get_rdms(theta)
        return self.energy_from_mo_coeff(mo_coeff, one_rdm, two_rdm)

    def circuit_gradient(self, theta):
        """oo_pqc.py:86-95"""
        return torch.autograd.functional.jacobian(self.energy_from_parameters, theta).reshape(-1)

    def orbital_gradient(self, theta):
        """oo_pqc.py:97-101"""
        one_rdm, two_rdm = self.pqc.get_rdms(theta)
        return self.kappa_matrix_to_vector(self.analytic_gradient(one_rdm, two_rdm))

    def circuit_circuit_hessian(self, theta):
        """oo_pqc.py:103-111"""
        n = int(np.prod(theta.shape))
        return torch.autograd.functional.hessian(self.energy_from_parameters, theta).reshape(n, n)

    def orbital_circuit_hessian(self, theta):
        """oo_pqc.py:113-125"""
        n = int(np.prod(theta.shape))
        return torch.autograd.functional.jacobian(self.orbital_gradient, theta).reshape(
            self.n_kappa, n)

    def orbital_orbital_hessian(self, theta):
        """oo_pqc.py:127-130"""
        one_rdm, two_rdm = self.pqc.get_rdms(theta)
        return self.full_hessian_to_matrix(self.analytic_hessian(one_rdm, two_rdm))

    def full_gradient(self, theta):
        """oo_pqc.py:132-134"""
        return torch.cat((self.circuit_gradient(theta), self.orbital_gradient(theta)))

    def full_hessian(self, theta):
        """oo_pqc.py:136-148"""
        h_cc = self.circuit_circuit_hessian(theta)
        h_oc = self.orbital_circuit_hessian(theta)
        h_oo = self.orbital_orbital_hessian(theta)
        return torch.cat((torch.cat((h_cc, h_oc.T), dim=1), torch.cat((h_oc, h_oo), dim=1)), dim=0)


# ----------------------------------------------------------------------------------------------
# Damped Newton step of the reference on CPU tensors (utils/newton_raphson.py:78-211): two eigh
# calls, explicit inverse, backtracking with the Armijo rule.  Checker for the device-side step.
# ----------------------------------------------------------------------------------------------
class OracleNewtonStep:
    def __init__(self, alpha=0.0001, beta=.5, mu=1e-6, rho=1.1, lmax=20, lambda_min=1e-6, aug=True):
        """newton_raphson.py:47-77"""
        self.alpha, self.beta, self.mu, self.rho = alpha, beta, mu, rho
        self.lmax, self.lambda_min, self.aug = lmax, lambda_min, aug

    def newton_step(self, gradient, hessian):
        """newton_raphson.py:107-128"""
        vals, vecs = torch.linalg.eigh(hessian)
        lowest = vals[0].item()
        if lowest < self.lambda_min and self.aug:
            hessian = hessian + (self.mu + self.rho * abs(lowest)) * torch.eye(hessian.shape[0],
                                                                               dtype=hessian.dtype)
            vals, vecs = torch.linalg.eigh(hessian)
        dp = -((vecs @ torch.diag(1 / vals) @ vecs.T) @ gradient)
        return dp, lowest

    def backtracking(self, objective_fn, parameters, dp, gradient):
        """newton_raphson.py:142-192"""
        def cut(flat):
            out, k = [], 0
            for p in parameters:
                out.append(flat[k:k + p.numel()].reshape(p.shape))
                k += p.numel()
            return out
        armijo = lambda t: self.alpha * t * torch.dot(gradient, dp)      # noqa: E731  (:12-13)
        t = 1.
        energy = objective_fn(*parameters).item()
        flat = torch.cat([p.flatten() for p in parameters])
        test = objective_fn(*cut(flat + t * dp))
        if test > energy + armijo(t):
            assert armijo(t) < 0
            num = 0
            while test > energy + armijo(t):
                t = self.beta * t
                test = objective_fn(*cut(flat + t * dp))
                num += 1
                if num > self.lmax:
                    t = 0.
                    test = objective_fn(*parameters)
                    break
        newp = flat + t * dp
        return (tuple(cut(newp)) if len(parameters) > 1 else newp), test.item()

    def damped_newton_step(self, objective_fn, parameters, gradient, hessian):
        """newton_raphson.py:194-211"""
        dp, lowest = self.newton_step(gradient, hessian)
        new_parameters, _ = self.backtracking(objective_fn, parameters, dp, gradient)
        return new_parameters, lowest


# ----------------------------------------------------------------------------------------------
# Synthetic inputs "of the named shape" (SURVEY.md section 8(d)); shared by tests and bench.
# ----------------------------------------------------------------------------------------------
def synthetic_problem(nao, seed, n_aux=None, enuc=31.0):
    """Returns dict(int1e_ao, int2e_ao, overlap, oao_mo_coeff, nuc) as numpy fp64 arrays.
    g_ao = (1/N_aux) sum_L B_Lpq B_Lrs with B symmetric in (p,q): 8-fold symmetric and PSD."""
    rng = np.random.default_rng(seed)
    n = nao
    n_aux = n if n_aux is None else n_aux
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = rng.uniform(0.3, 1.7, size=n)
    S = (Q * lam) @ Q.T
    A = rng.standard_normal((n, n))
    h = (A + A.T) / (2 * np.sqrt(n)) - np.diag(np.linspace(3.0, 0.0, n))
    B = rng.standard_normal((n_aux, n, n))
    B = 0.5 * (B + B.transpose(0, 2, 1))
    g = np.einsum('Lpq,Lrs->pqrs', B, B, optimize=True) / n_aux
    # bit-for-bit p<->q and r<->s symmetry, as integral packages deliver it
    g = 0.5 * (g + g.transpose(1, 0, 2, 3))
    g = 0.5 * (g + g.transpose(0, 1, 3, 2))
    Qc, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return dict(int1e_ao=h, int2e_ao=g, overlap=S, oao_mo_coeff=Qc, nuc=float(enuc))


def bogoliubov_unitary(U_act):
    """Tutorial_Berry_phase.ipynb cell 27: G = exp{ sum_pq [log U]_pq c+_p c_q } over both spins of
    the active orbitals, as a dense 2^n x 2^n matrix built from the Jordan-Wigner E_pq operators
    (the notebook obtains it from openfermion.bogoliubov_transform + cirq, normalised to
    G[0, 0] = 1, which this exponential satisfies by construction).  U_act must be a proper
    rotation (orthogonal, det +1) so that log U is real antisymmetric."""
    import scipy.linalg
    U_act = np.asarray(U_act, dtype=np.float64)
    ncas = U_act.shape[0]
    logU = scipy.linalg.logm(U_act)
    if np.abs(np.imag(logU)).max() > 1e-10:
        raise ValueError("log U is not real: U must be a proper rotation")
    logU = np.real(logU)
    ops = RdmOperators(ncas)
    K = sum(logU[p, q] * ops.E[p][q].toarray() for p in range(ncas) for q in range(ncas))
    return scipy.linalg.expm(K)


def orbital_rotation_operator(U_act):
    """The same operator from its action on determinants, valid for ANY orthogonal U (also
    det U = -1, which is what a Berry-phase loop around a conical intersection returns to):
    G a+_{q sigma} G^-1 = sum_p U_pq a+_{p sigma},  G |vac> = |vac>  (the notebook's gauge
    G[0, 0] = 1, Tutorial_Berry_phase.ipynb cell 28), hence
    G |m1 m2 ...> = b+_{m1} b+_{m2} ... |vac>.  Dense 2^n x 2^n from Jordan-Wigner operators."""
    U_act = np.asarray(U_act, dtype=np.float64)
    ncas = U_act.shape[0]
    n = 2 * ncas
    a = _jw_annihilators(n)
    ad = [m.T.toarray() for m in a]
    b_dag = []
    for m in range(n):
        q, spin = divmod(m, 2)
        b_dag.append(sum(U_act[p, q] * ad[2 * p + spin] for p in range(ncas)))
    D = 1 << n
    vac = np.zeros(D)
    vac[0] = 1.0
    G = np.zeros((D, D))
    for I in range(D):
        modes = [m for m in range(n) if (I >> (n - 1 - m)) & 1]      # wire 0 = most significant bit
        vec = vac
        for m in reversed(modes):
            vec = b_dag[m] @ vec
        G[:, I] = vec
    return G

