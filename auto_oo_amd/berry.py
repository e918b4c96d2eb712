"""Berry-phase post-processing: project a circuit state onto the orbital basis of another geometry.

The reference's notebook (examples/Tutorial_Berry_phase.ipynb, cells 27-32) transforms the
state of loop point a into the molecular-orbital basis of point b with the active-space
Bogoliubov transformation

    G_{a->b} = exp{ sum_{p,q in active space} [log C_{a->b}]_pq  c+_p c_q },
    C_{a->b} = (oao_mo_coeff_a^T oao_mo_coeff_b)[active, active]   (both spins),

built there from ``openfermion.bogoliubov_transform`` + cirq as a dense 2^n x 2^n unitary, and
estimates the Berry phase from the overlaps <psi_b| G_{a->b} |psi_a> around the loop.

Here the same operator is applied without ever forming a 2^n x 2^n matrix: a determinant |I> =
|I_alpha, I_beta> goes to  sum_J  det U[J_alpha, I_alpha] det U[J_beta, I_beta] |J>  (U = C_{a->b}
made exactly orthogonal the way the notebook's route does it, see givens_orthogonal; the spin
sectors only mix inside themselves; nothing needs det U = +1, which matters: the orbitals of a loop
around a conical intersection come back with det U = -1), so in
the (N_alpha, N_beta) sector the map is  psi' = M_alpha psi M_beta^T  with the minor matrices
M[J, I] = det U[J, I] -- two small dense products on the device (``oovqe_matmul_nn``).  The sign
that reorders the interleaved Jordan-Wigner operator string (alpha_0 beta_0 alpha_1 ...) into
(all alpha)(all beta) is a per-determinant table.
"""
import numpy as np
import torch

from . import ops
from .sector import string_tables, sector_of

F64 = torch.float64


def _occupied(string, ncas):
    """Orbitals occupied in an occupation string (orbital p at bit ncas-1-p), ascending."""
    return [p for p in range(ncas) if (int(string) >> (ncas - 1 - p)) & 1]


def _spread(string):
    """bit b -> bit 2b"""
    out, b = 0, 0
    s = int(string)
    while s:
        if s & 1:
            out |= 1 << (2 * b)
        s >>= 1
        b += 1
    return out


def polar_orthogonal(mat):
    """Closest orthogonal matrix (polar factor): the active block of an orthogonal MO overlap
    matrix is only approximately orthogonal."""
    u, _, vt = np.linalg.svd(np.asarray(mat, dtype=np.float64))
    return u @ vt


def givens_orthogonal(mat):
    """The orthogonal matrix the notebook's route actually applies when the active block is not
    orthogonal (it never is exactly: the Newton step also mixes active with inactive / virtual
    orbitals): openfermion.bogoliubov_transform zeroes the upper triangle of its argument
    (= the transposed block) with Givens rotations of adjacent columns and keeps only the signs of
    the remaining diagonal, i.e. it uses the Q factor of block = Q R with a positive diagonal of R
    (Gram-Schmidt on the block's columns).  Reproduces the ten overlaps printed in
    examples/Tutorial_Berry_phase.ipynb (cell 32) to 1e-7; the polar factor differs from them by
    up to 1.3e-2."""
    q, r = np.linalg.qr(np.asarray(mat, dtype=np.float64))
    return q * np.where(np.diag(r) < 0, -1.0, 1.0)


def minor_matrix(U, strings, ncas):
    """M[J, I] = det U[occ(J), occ(I)] over the given occupation strings."""
    occ = [_occupied(s, ncas) for s in strings]
    n = len(occ)
    M = np.empty((n, n))
    for j in range(n):
        for i in range(n):
            M[j, i] = np.linalg.det(U[np.ix_(occ[j], occ[i])]) if occ[j] else 1.0
    return M


def sector_tables(ncas, n_alpha, n_beta):
    """(strings_a, strings_b, full index x[ia, ib], sign s[ia, ib]) of the (N_alpha, N_beta) sector:
    x is the computational-basis index (wire 0 = MSB, wire 2p = alpha of orbital p); s is the
    parity of moving every alpha creation operator in front of the beta ones."""
    ua, _ = string_tables(ncas, n_alpha)
    ub, _ = string_tables(ncas, n_beta)
    x = np.empty((len(ua), len(ub)), dtype=np.int64)
    s = np.empty((len(ua), len(ub)))
    occ_a = [_occupied(m, ncas) for m in ua]
    occ_b = [_occupied(m, ncas) for m in ub]
    for ia, ma in enumerate(ua):
        for ib, mb in enumerate(ub):
            x[ia, ib] = (_spread(ma) << 1) | _spread(mb)
            inv = sum(1 for p in occ_a[ia] for q in occ_b[ib] if q < p)
            s[ia, ib] = -1.0 if inv & 1 else 1.0
    return ua, ub, x, s


class ActiveSpaceRotation:
    """G = exp{ sum_pq [log U]_pq (c+_{p,alpha} c_{q,alpha} + c+_{p,beta} c_{q,beta}) } restricted to
    the (N_alpha, N_beta) sector of a 2*ncas-qubit register (the sector every UCC / kUpCCD /
    GateFabric state of the package lives in)."""

    def __init__(self, U, ncas, n_alpha, n_beta, device=None, orthogonalize="givens"):
        """orthogonalize: "givens" (default; what the notebook's openfermion route does, see
        givens_orthogonal), "polar" (closest orthogonal matrix), or False (U is used as it is)."""
        U = np.asarray(U, dtype=np.float64)
        if U.shape != (ncas, ncas):
            raise ValueError(f"U must be [{ncas}, {ncas}], got {U.shape}")
        if orthogonalize is True or orthogonalize == "givens":
            U = givens_orthogonal(U)
        elif orthogonalize == "polar":
            U = polar_orthogonal(U)
        elif orthogonalize:
            raise ValueError("orthogonalize must be 'givens', 'polar' or False")
        self.U, self.ncas = U, ncas
        self.strings_a, self.strings_b, self.index, self.sign = sector_tables(ncas, n_alpha, n_beta)
        self.M_alpha = minor_matrix(U, self.strings_a, ncas)
        self.M_beta = minor_matrix(U, self.strings_b, ncas)
        self.device = device
        self._dev = None

    def dense(self):
        """The sector block of G as a dense matrix on the full 2^n register (numpy; for tests and
        small registers)."""
        D = 1 << (2 * self.ncas)
        G = np.zeros((D, D))
        flat_x = self.index.reshape(-1)
        flat_s = self.sign.reshape(-1)
        block = np.kron(self.M_alpha, self.M_beta) * flat_s[:, None] * flat_s[None, :]
        G[np.ix_(flat_x, flat_x)] = block
        return G

    def _tables(self, device):
        if self._dev is None or self._dev[0] != str(device):
            self._dev = (str(device),
                         torch.as_tensor(self.index.reshape(-1)).to(device),
                         torch.as_tensor(self.sign.reshape(-1)).to(device),
                         torch.as_tensor(self.M_alpha).to(device).contiguous(),
                         torch.as_tensor(np.ascontiguousarray(self.M_beta.T)).to(device).contiguous())
        return self._dev[1:]

    def apply(self, state):
        """G |state> for a dense statevector of length 2^(2 ncas) on the device (real or complex);
        amplitudes outside the sector are dropped (they are zero for the package's ansaetze)."""
        state = torch.as_tensor(state)
        if state.is_complex():
            re, im = self.apply(state.real.contiguous()), self.apply(state.imag.contiguous())
            return torch.complex(re, im)
        dev = state.device if state.is_cuda else (self.device or torch.device("cuda"))
        state = ops.as_device(state.to(F64), dev).reshape(-1)
        idx, sgn, Ma, MbT = self._tables(dev)
        na, nb = self.M_alpha.shape[0], self.M_beta.shape[0]
        v = (state[idx] * sgn).reshape(na, nb).contiguous()
        v = ops.matmul_nn(ops.matmul_nn(Ma, v), MbT)            # M_alpha v M_beta^T
        out = torch.zeros_like(state)
        out[idx] = v.reshape(-1) * sgn
        return out


def bogoliubov_atob_cas(mo_atob, active_indices, nelecas, device=None, orthogonalize="givens"):
    """Notebook cell 28: the active-space Bogoliubov transformation of ``mo_atob`` =
    oao_mo_coeff_a^T @ oao_mo_coeff_b, as an operator on (N_alpha, N_beta)-sector states
    (``nelecas`` electrons, closed shell or high-spin-first as ``hf_state`` fills the wires)."""
    mo_atob = np.asarray(torch.as_tensor(mo_atob).detach().cpu(), dtype=np.float64)
    act = list(active_indices)
    U = mo_atob[np.ix_(act, act)]
    ncas = len(act)
    occ = [1 if i < nelecas else 0 for i in range(2 * ncas)]
    n_alpha, n_beta = sector_of(occ, ncas)
    return ActiveSpaceRotation(U, ncas, n_alpha, n_beta, device=device, orthogonalize=orthogonalize)


def state_overlap(state_b, rotation, state_a):
    """<state_b| G |state_a> (notebook cell 32): the Berry-phase estimator is the product / final
    value of these overlaps around the loop."""
    ga = rotation.apply(state_a)
    sb = torch.as_tensor(state_b).to(ga.device)
    if sb.is_complex() or ga.is_complex():
        return torch.sum(torch.conj(sb.to(torch.complex128)) * ga.to(torch.complex128))
    return torch.sum(sb.to(F64) * ga)
