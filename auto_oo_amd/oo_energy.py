"""OO_energy: orbital-optimized energy, analytic orbital gradient (and Hessian) on the MI355X.

Drop-in for the reference's ``auto_oo.OO_energy`` and the free functions of
src/auto_oo/oo_energy.py (same names, argument meaning and error behaviour).  All arithmetic runs
in hand-written HIP kernels behind the C ABI of include/oovqe.h; torch tensors only carry device
memory.  There is no CPU path.
"""
from functools import partial

import numpy as np
import torch

from . import _lib, excitations as X, ops
from .autodiff import differentiable_scalar, is_zero_tangent, kernel_scope, needs_autodiff, unwrap
from .newton_raphson import NewtonStep

F64 = torch.float64


# ------------------------------------------------------------------------------------------------
# free functions (oo_energy.py:21-118)
# ------------------------------------------------------------------------------------------------
def general_4index_transform(M, C0, C1, C2, C3):
    """oo_energy.py:21-30: M'_{ijkl} = sum_pqrs C0_pi C1_qj C2_rk C3_sl M_pqrs (four fp64-MFMA
    mode contractions, ``oovqe_general_4index_transform``)."""
    dev = _lib.require_device()
    return ops.general_4index_transform(ops.as_device(M, dev), ops.as_device(C0, dev),
                                        ops.as_device(C1, dev), ops.as_device(C2, dev),
                                        ops.as_device(C3, dev))


def uniform_4index_transform(M, C):
    """oo_energy.py:33-41"""
    return general_4index_transform(M, C, C, C, C)


def int1e_transform(int1e_ao, mo_coeff):
    """oo_energy.py:44-46: C^T h C"""
    dev = _lib.require_device()
    h, C = ops.as_device(int1e_ao, dev), ops.as_device(mo_coeff, dev)
    return ops.matmul_nn(ops.matmul_tn(C, h), C)


def int2e_transform(int2e_ao, mo_coeff):
    """oo_energy.py:49-51"""
    return uniform_4index_transform(int2e_ao, mo_coeff)


def mo_ao_to_mo_oao(mo_coeff, overlap):
    """oo_energy.py:54-60 (numpy on the host, once per molecule, as in the reference)"""
    S_eigval, S_eigvec = np.linalg.eigh(np.asarray(overlap, dtype=np.float64))
    S_half = S_eigvec @ np.diag(S_eigval ** 0.5) @ S_eigvec.T
    return S_half @ np.asarray(mo_coeff, dtype=np.float64)


def vector_to_skew_symmetric(vector):
    """oo_energy.py:63-87: strict lower triangle (np.tril_indices order) = vector, upper = -vector.
    Pure data movement (index scatter)."""
    vector = torch.as_tensor(vector)
    size = int(np.sqrt(8 * vector.shape[0] + 1) + 1) // 2
    matrix = torch.zeros((size, size), dtype=vector.dtype, device=vector.device)
    tril = np.tril_indices(size, k=-1)
    r = torch.as_tensor(tril[0], device=vector.device)
    c = torch.as_tensor(tril[1], device=vector.device)
    matrix[r, c] = vector
    matrix[c, r] = -vector
    return matrix


def skew_symmetric_to_vector(kappa_matrix):
    """oo_energy.py:90-94"""
    size = kappa_matrix.shape[0]
    tril = np.tril_indices(size, k=-1)
    return kappa_matrix[tril[0], tril[1]]


def non_redundant_indices(occ_idx, act_idx, virt_idx, freeze_active):
    """oo_energy.py:97-118"""
    return X.non_redundant_indices(occ_idx, act_idx, virt_idx, freeze_active)


# ------------------------------------------------------------------------------------------------
# derivative rules of the cost functions for torch's autodiff (auto_oo_amd/autodiff.py)
# ------------------------------------------------------------------------------------------------
def _is_zero(t):
    if t is None:
        return True
    with kernel_scope():
        return not bool((t != 0).any())


def _matvec(H, v):
    # (a row-wise product and sum: the second-order rules stay off the vendor BLAS)
    return None if v is None else (H * v.reshape(1, -1)).sum(dim=1)


def _apply_rows(fn, v, shape):
    """fn on ONE direction v.  A stack of directions (torch.func's vmap over the cotangents of a Hessian hands
    a derivative rule a functorch BatchedTensor) is refused loudly: the kernels take one direction per call,
    and looping over the rows of the stack from inside a rule means re-entering functorch's interpreter
    stack by hand (round 4 tried: unwrap the stack, apply row by row, re-wrap with _add_batch_dim -- fine for a
    lone vmap, but under hessian's jvp-over-vmap nesting the rows come back wrapped at the wrong level).
    torch.autograd.functional.hessian / jacobian (vectorize=False) and torch.func at kappa = 0 cover it."""
    if torch._C._functorch.is_functorch_wrapped_tensor(v) and torch._C._functorch.is_batchedtensor(v):
        raise NotImplementedError("batched tangents at kappa != 0: use torch.autograd.functional.hessian "
                                  "(or vectorize=False)")
    return fn(unwrap(v).reshape(-1)).reshape(shape)


class _OrbitalRotationRule:
    """Shared chain rule through C(kappa) = C0 expm(-K(kappa)) (oo_energy.py:226-236).

    With F the generalized Fock matrix at C (oo_energy.py:238-270) the derivative of the energy
    with respect to an unconstrained C is dE/dC = C^-T (2 F^T) (virtual rows of F are zero: the
    energy does not depend on virtual orbitals), hence dE/dU = U (2 F^T) for C = C0 U, and through
    the matrix exponential dE/dK = -L_exp(K; dE/dU), the Frechet derivative of exp at K (the adjoint
    of L_exp(-K; .) for skew K), evaluated as the upper right block of expm([[K, X], [0, K]]).
    At kappa = 0 this is the analytic orbital gradient 2 (F - F^T) of the reference."""

    def __init__(self, oo):
        self.oo = oo

    def rotated(self, kappa):
        oo = self.oo
        if _is_zero(kappa):
            return oo.mo_coeff, None, None
        U, K = ops.expm_skew(oo._t(kappa).reshape(-1), oo._kap_row, oo._kap_col, oo.nao, want_K=True)
        return ops.matmul_nn(oo.mo_coeff, U), U, K

    # ---- second order at kappa != 0 -------------------------------------------------------------
    # E(theta, kappa_bar + d) = e(theta, x(d)) with e the energy in the LOCAL rotation parameters of the
    # rotated orbitals C_bar = C0 U_bar (all N(N-1)/2 pairs: x = 0 is C_bar itself) and
    # exp(-K_full(x(d))) = U_bar^T exp(-K(kappa_bar + d)).  Every second derivative the kernels know is
    # local (analytic Hessian blocks at C_bar, oo_energy.py:311-402, oo_pqc.py:103-148); what is left is the
    # small-matrix map d -> x(d), through the first and second Frechet derivatives of the exponential:
    #   J v      = pairs( U_bar^T L(-K_bar; K(v)) )
    #   J^T w    = pairs_kappa( 1/2 L(K_bar; U_bar W) )                       W = skew(w)
    #   sum_i g_i d^2 x_i [., v] = pairs_kappa( 1/2 L2(K_bar; U_bar G, K(v)) + 1/4 L(K_bar; U_bar (G B^T + B^T G)) )
    #                              G = skew(g), B = U_bar^T L(-K_bar; K(v))
    # L(X; A) = upper right block of expm([[X, A], [0, X]]); L2(X; A, B) = the (1,3) blocks of
    # expm([[X, A, 0], [0, X, B], [0, 0, X]]) for (A, B) and (B, A); the adjoint of L(X; .) is L(X^T; .), that
    # of L2(X; ., V) is L2(X^T; ., V^T) (checked against torch autograd of the same map to 1e-14).
    def _full_pairs(self):
        oo = self.oo
        tabs = oo.__dict__.get("_full_pair_tables")
        if tabs is None:
            tr, tc = np.tril_indices(oo.nao, -1)
            tabs = (torch.as_tensor(tr.astype(np.int32)).to(oo.device),
                    torch.as_tensor(tc.astype(np.int32)).to(oo.device))
            oo.__dict__["_full_pair_tables"] = tabs
        return tabs

    def _frechet(self, X, A):
        N = X.shape[0]
        blk = torch.zeros((2 * N, 2 * N), dtype=F64, device=X.device)
        blk[:N, :N] = X
        blk[N:, N:] = X
        blk[:N, N:] = A
        return ops.expm(blk, 1.0)[:N, N:]

    def _frechet2(self, X, A, B):
        N = X.shape[0]

        def ordered(P, Q):
            blk = torch.zeros((3 * N, 3 * N), dtype=F64, device=X.device)
            for i in range(3):
                blk[i * N:(i + 1) * N, i * N:(i + 1) * N] = X
            blk[:N, N:2 * N] = P
            blk[N:2 * N, 2 * N:] = Q
            return ops.expm(blk, 1.0)[:N, 2 * N:]
        return ordered(A, B) + ordered(B, A)

    def _kmat(self, v):
        oo = self.oo
        r, c = oo._kap_row.long(), oo._kap_col.long()
        Km = torch.zeros((oo.nao, oo.nao), dtype=F64, device=oo.device)
        Km[r, c] = v.reshape(-1)
        Km[c, r] = -v.reshape(-1)
        return Km

    def _skew_full(self, w):
        oo = self.oo
        tr, tc = self._full_pairs()
        Wm = torch.zeros((oo.nao, oo.nao), dtype=F64, device=oo.device)
        Wm[tr.long(), tc.long()] = w
        Wm[tc.long(), tr.long()] = -w
        return Wm

    def local_push(self, U, K, v):
        """J v (all pairs) and B = U^T L(-K; K(v))."""
        tr, tc = self._full_pairs()
        B = ops.matmul_tn(U, self._frechet(-K, self._kmat(v)).contiguous())
        return B[tr.long(), tc.long()], B

    def local_pull(self, U, K, w):
        """J^T w on the non-redundant kappa pairs (w over all pairs)."""
        oo = self.oo
        Y = 0.5 * self._frechet(K.contiguous(), ops.matmul_nn(U, self._skew_full(w)))
        r, c = oo._kap_row.long(), oo._kap_col.long()
        return Y[r, c] - Y[c, r]

    def local_curvature(self, U, K, Gm, v, B):
        """sum_i g_i d^2 x_i / d kappa d kappa . v  for the local gradient G (skew matrix)."""
        oo = self.oo
        Z = 0.5 * self._frechet2(K.contiguous(), ops.matmul_nn(U, Gm.contiguous()), self._kmat(v)) \
            + 0.25 * self._frechet(K.contiguous(), ops.matmul_nn(U, (ops.matmul_nn(Gm.contiguous(), B.T.contiguous()) + ops.matmul_nn(B.T.contiguous(), Gm.contiguous())).contiguous()))
        r, c = oo._kap_row.long(), oo._kap_col.long()
        return Z[r, c] - Z[c, r]

    def kappa_gradient(self, fock, gvec0, U, K):
        """dE/dkappa from the Fock matrix at C0 U (gvec0: its value when kappa = 0)."""
        if U is None:
            return gvec0
        oo = self.oo
        N = oo.nao
        X = ops.matmul_nn(U, (2.0 * fock.T).contiguous())
        blk = torch.zeros((2 * N, 2 * N), dtype=F64, device=oo.device)
        blk[:N, :N] = K
        blk[N:, N:] = K
        blk[:N, N:] = X
        gK = -ops.expm(blk, 1.0)[:N, N:]
        r, c = oo._kap_row.long(), oo._kap_col.long()
        return gK[r, c] - gK[c, r]


class _KappaEnergyModel(_OrbitalRotationRule):
    """E(kappa, one_rdm, two_rdm) = OO_energy.energy_from_kappa: value, gradient
    (dE/dkappa as above, dE/d one_rdm = c1, dE/d two_rdm = c2 -- E is linear in the RDMs,
    oo_energy.py:191-197) and the kappa-kappa second derivative at kappa = 0 (the analytic orbital
    Hessian, oo_energy.py:311-402, what test/test_oo_energy.py:937-943 compares with autodiff)."""

    def value(self, kappa, one_rdm, two_rdm):
        C, _, _ = self.rotated(kappa)
        return self.oo._energy_from_mo_coeff(C, one_rdm, two_rdm)

    def grad(self, kappa, one_rdm, two_rdm):
        oo = self.oo
        C, U, K = self.rotated(kappa)
        g1, g2 = oo._rdm_stack(one_rdm, two_rdm)
        res = oo._cas_eval(C, g1, g2, want_matrices=True)
        gk = self.kappa_gradient(res["fock"], res["gvec"][0], U, K)
        return gk.reshape(kappa.shape), res["c1"].reshape(one_rdm.shape), res["c2"].reshape(two_rdm.shape)

    def hvp(self, xs, vs, needs=None):
        kappa, one_rdm, two_rdm = (unwrap(x) for x in xs)
        if (not is_zero_tangent(vs[1]) or not is_zero_tangent(vs[2])
                or (needs is not None and (needs[1] or needs[2]))):
            raise NotImplementedError("second derivatives of energy_from_kappa involving the RDMs "
                                      "are not built")
        if vs[0] is None:
            return None, None, None
        oo = self.oo
        if _is_zero(kappa):
            H = getattr(self, "_H", None)
            if H is None:
                with kernel_scope():
                    H = self._H = oo.analytic_hessian_matrix(one_rdm, two_rdm)
            return _matvec(H, vs[0]).reshape(kappa.shape), None, None
        # kappa != 0: the analytic Hessian over ALL rotation pairs at the rotated orbitals, pulled back
        cache = getattr(self, "_rot", None)
        if cache is None:
            with kernel_scope():
                C, U, K = self.rotated(kappa)
                g1, g2 = oo._rdm_stack(one_rdm, two_rdm)
                res = oo._cas_eval(C, g1, g2, want_matrices=True)
                tr, tc = self._full_pairs()
                Hm = ops.orbital_hessian(oo.int2e_ao, oo.int1e_ao, oo._t(C), g1[0].contiguous(), g2[0].contiguous(),
                                         res["fock"], oo._n_occ, oo.ncas, tr, tc, want_matrix=True)[0]
                cache = self._rot = (U, K, res["gmat"], Hm)
        U, K, Gm, Hm = cache

        def one(v):
            with kernel_scope():
                xv, B = self.local_push(U, K, v)
                return self.local_pull(U, K, _matvec(Hm, xv)) + self.local_curvature(U, K, Gm, v, B)
        return _apply_rows(one, vs[0], kappa.shape), None, None


class _MoCoeffEnergyModel:
    """E(mo_coeff, one_rdm, two_rdm) = OO_energy.energy_from_mo_coeff, first derivatives:
    dE/dC = C^-T (2 F^T), dE/d one_rdm = c1, dE/d two_rdm = c2."""

    def __init__(self, oo):
        self.oo = oo

    def value(self, C, one_rdm, two_rdm):
        return self.oo._energy_from_mo_coeff(C, one_rdm, two_rdm)

    def grad(self, C, one_rdm, two_rdm):
        oo = self.oo
        g1, g2 = oo._rdm_stack(one_rdm, two_rdm)
        res = oo._cas_eval(C, g1, g2, want_matrices=True)
        W = torch.linalg.solve(oo._t(C).T, 2.0 * res["fock"].T)
        return W, res["c1"].reshape(one_rdm.shape), res["c2"].reshape(two_rdm.shape)

    def hvp(self, xs, vs, needs=None):
        raise NotImplementedError("second derivatives through energy_from_mo_coeff are not built; "
                                  "differentiate energy_from_kappa instead")


# ------------------------------------------------------------------------------------------------
class OO_energy:
    """Orbital Optimized energy class for extracting energies for any given set of RDMs.  Can
    compute analytical orbital gradients and hessians (oo_energy.py:121-474)."""

    def __init__(self, mol, ncas, nelecas, oao_mo_coeff=None, freeze_active=False,
                 interface='torch'):
        if interface != 'torch':
            raise ValueError("auto_oo_amd supports interface='torch' only (PyTorch-ROCm tensors)")
        self.device = _lib.require_device()
        if oao_mo_coeff is None:
            mol.run_rhf()
            self.oao_mo_coeff = ops.as_device(mo_ao_to_mo_oao(mol.hf.mo_coeff, mol.overlap),
                                              self.device)
        else:
            self.oao_mo_coeff = ops.as_device(oao_mo_coeff, self.device)
        self.interface = interface
        self._freeze_active = bool(freeze_active)

        # molecular data, resident in HBM
        self.int1e_ao = ops.as_device(mol.int1e_ao, self.device)
        self.int2e_ao = ops.as_device(mol.int2e_ao, self.device)
        self.overlap = mol.overlap
        self.oao_coeff = ops.as_device(mol.oao_coeff, self.device)
        self.nuc = mol.nuc
        self.nao = mol.nao

        self.ncas = ncas
        self.nelecas = nelecas
        self.occ_idx, self.act_idx, self.virt_idx = mol.get_active_space_idx(ncas, nelecas)
        if len(self.occ_idx) and not np.array_equal(self.occ_idx, np.arange(len(self.occ_idx))):
            raise ValueError("occupied orbitals must be the leading contiguous range")
        self._n_occ = len(self.occ_idx)
        self._M = self._n_occ + ncas

        self.params_idx = non_redundant_indices(self.occ_idx, self.act_idx, self.virt_idx,
                                                freeze_active)
        self.n_kappa = len(self.params_idx)
        rows, cols = X.tril_tables(self.nao, self.params_idx)
        self._kap_row = torch.as_tensor(rows).to(self.device)
        self._kap_col = torch.as_tensor(cols).to(self.device)

    # ---- helpers ----------------------------------------------------------------------------------
    def _t(self, x):
        if (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == F64 and x.is_contiguous()
                and not x.requires_grad):
            return x
        return ops.as_device(x, self.device)

    def _cas_intermediates(self, mo_coeff):
        """Stage 1 + 2 of the CAS path: Gm[n,x,y,z] and hmo[n,x] (x,y,z < n_occ+ncas)."""
        C = self._t(mo_coeff)
        T2 = ops.cas_half_transform(self.int2e_ao, C, self._M)
        return ops.cas_finish_transform(T2, self.int1e_ao, C, self._M)

    def _eri_flags(self, g_ao=None):
        """Symmetry flags of a resident two-electron tensor (ops.eri_flags), verified once per
        tensor STATE: the cache key is the tensor object and its autograd version counter, so an
        in-place edit of ``int2e_ao`` (the reference lets users mutate it) re-verifies instead of
        keeping a stale "symmetric" promise -- same rule as the ``mo_coeff`` cache below.  PySCF
        integrals are exactly p<->q symmetric, so the N^4 pass reads half of the tensor; anything
        else is treated as a general tensor."""
        g_ao = self.int2e_ao if g_ao is None else g_ao
        hit = self.__dict__.get("_eri_flag_cache")
        if hit is None or hit[0] is not g_ao or hit[1] != g_ao._version:
            hit = (g_ao, g_ao._version, ops.eri_flags(g_ao))
            self.__dict__["_eri_flag_cache"] = hit
        return hit[2]

    def _eri_packed(self, g_ao=None):
        """The packed resident copy of ``int2e_ao`` for the N^4 pass (ops.eri_pack), kept per tensor STATE like
        the flags: only for integrals that carry both symmetry flags and N > 48 (29 % of the tensor at N = 200,
        in the order and with the weights stage 1 consumes it; below that size a single geometry's pass is
        latency-bound and reads the tensor itself).  None otherwise."""
        g_ao = self.int2e_ao if g_ao is None else g_ao
        if g_ao.shape[-1] <= 48 or self._eri_flags(g_ao) != 3:
            return None
        hit = self.__dict__.get("_eri_pack_cache")
        if hit is None or hit[0] is not g_ao or hit[1] != g_ao._version:
            hit = (g_ao, g_ao._version, ops.eri_pack(g_ao))
            self.__dict__["_eri_pack_cache"] = hit
        return hit[2]

    def reverify_integrals(self):
        """Call after writing into ``int2e_ao`` / ``int1e_ao`` through a path torch's version counter does
        not see (a kernel writing through a raw pointer, ``.data`` edits, an external producer): drops the
        cached symmetry flags and evaluation plans, so the next call verifies the tensor again bit for bit.
        In-place torch ops (``tensor.mul_(..)``, indexing assignments) bump the version counter and need no
        call.  (``OO_pqc_batch.reverify_integrals`` is the batched counterpart.)"""
        self.__dict__.pop("_eri_flag_cache", None)
        self.__dict__.pop("_eri_pack_cache", None)
        self.__dict__.pop("_plans2", None)
        self.__dict__.pop("_full_pair_tables", None)
        self.__dict__.pop("_stack1", None)          # (OO_pqc: the one-geometry stack holds a copy of the integrals)
        self.__dict__.pop("_hess1_plans", None)

    def _cas_eval(self, mo_coeff, gamma_sets, Gamma_sets, want_matrices=False):
        """Fused energy / Fock / orbital gradient for a stack of RDM sets (set 0 = RDMs, sets k>=1
        = derivative RDMs)."""
        return ops.cas_eval(self.int2e_ao, self.int1e_ao, self._t(mo_coeff), gamma_sets, Gamma_sets,
                            self.nuc, self._n_occ, self.ncas, self._kap_row, self._kap_col,
                            want_matrices=want_matrices, eri_flags=self._eri_flags(),
                            g_packed=self._eri_packed())

    def _rdm_stack(self, one_rdm, two_rdm):
        g1 = self._t(one_rdm).reshape(1, self.ncas, self.ncas)
        g2 = self._t(two_rdm).reshape((1,) + (self.ncas,) * 4)
        return g1, g2

    # ---- reference API ------------------------------------------------------------------------------
    @property
    def mo_coeff(self):
        """oo_energy.py:173-176: oao_coeff @ oao_mo_coeff.  The reference recomputes the product on
        every access; here it is recomputed whenever ``oao_mo_coeff`` was reassigned or modified in
        place (tensor identity + version counter), which yields the same values."""
        src = self.oao_mo_coeff
        key = (id(src), src._version if isinstance(src, torch.Tensor) else None,
               id(self.oao_coeff))
        cache = getattr(self, "_mo_cache", None)
        if cache is None or cache[0] != key or cache[1] is not src:
            self._mo_cache = (key, src, ops.matmul_nn(self.oao_coeff, self._t(src)))
        return self._mo_cache[2]

    def _diff_args(self, *xs):
        """Arguments of a cost function as fp64 device tensors WITHOUT leaving torch's autodiff
        graph (``.to`` is a differentiable op; ``_t`` detaches)."""
        return tuple(torch.as_tensor(x).to(device=self.device, dtype=F64).contiguous() for x in xs)

    def _energy_from_mo_coeff(self, mo_coeff, one_rdm, two_rdm):
        g1, g2 = self._rdm_stack(one_rdm, two_rdm)
        return self._cas_eval(mo_coeff, g1, g2)["E"].reshape(())

    def energy_from_mo_coeff(self, mo_coeff, one_rdm, two_rdm):
        """oo_energy.py:178-197: E = c0 + sum c1*gamma + sum c2*Gamma (0-dim tensor).
        Differentiable by torch (first order) with respect to all three arguments."""
        if needs_autodiff(mo_coeff, one_rdm, two_rdm):
            return differentiable_scalar(_MoCoeffEnergyModel(self),
                                         *self._diff_args(mo_coeff, one_rdm, two_rdm))
        return self._energy_from_mo_coeff(mo_coeff, one_rdm, two_rdm)

    def energy_from_kappa(self, kappa, one_rdm, two_rdm):
        """oo_energy.py:199-202.  Differentiable by torch autograd / torch.func: first order in
        (kappa, one_rdm, two_rdm) at any kappa, second order in kappa at kappa = 0
        (test/test_oo_energy.py:930-943)."""
        if needs_autodiff(kappa, one_rdm, two_rdm):
            return differentiable_scalar(_KappaEnergyModel(self),
                                         *self._diff_args(kappa, one_rdm, two_rdm))
        mo_coeff = ops.matmul_nn(self.mo_coeff, self.kappa_to_mo_coeff(kappa))
        return self._energy_from_mo_coeff(mo_coeff, one_rdm, two_rdm)

    def get_active_integrals(self, mo_coeff):
        """oo_energy.py:204-211: CAS Hamiltonian coefficients (c0, c1, c2) in chemist notation."""
        dummy1 = torch.zeros((1, self.ncas, self.ncas), dtype=F64, device=self.device)
        dummy2 = torch.zeros((1,) + (self.ncas,) * 4, dtype=F64, device=self.device)
        res = self._cas_eval(mo_coeff, dummy1, dummy2)
        return res["c0"].reshape(()), res["c1"], res["c2"]

    def kappa_vector_to_matrix(self, kappa):
        """oo_energy.py:213-219"""
        kappa = self._t(kappa)
        total = torch.zeros(self.nao * (self.nao - 1) // 2, dtype=F64, device=self.device)
        total[torch.as_tensor(self.params_idx, device=self.device)] = kappa
        return vector_to_skew_symmetric(total)

    def kappa_matrix_to_vector(self, kappa_matrix):
        """oo_energy.py:221-224 (index gather)"""
        return kappa_matrix[self._kap_row.long(), self._kap_col.long()]

    def kappa_to_mo_coeff(self, kappa):
        """oo_energy.py:226-230: expm(-K) (``oovqe_expm_skew``: scatter + expm in one launch)."""
        return ops.expm_skew(self._t(kappa).reshape(-1), self._kap_row, self._kap_col, self.nao)

    def get_transformed_mo(self, mo_coeff, kappa):
        """oo_energy.py:232-236"""
        return ops.matmul_nn(self._t(mo_coeff), self.kappa_to_mo_coeff(kappa))

    # ---- Fock matrices / gradient -------------------------------------------------------------------
    def _fock_from_integrals(self, int1e_mo, int2e_mo, one_rdm, two_rdm):
        M = self._M
        g = self._t(int2e_mo)
        Gm = g[:, :M, :M, :M].contiguous()
        hmo = self._t(int1e_mo)[:, :M].contiguous()
        g1, g2 = self._rdm_stack(one_rdm, two_rdm)
        return ops.cas_energy_gradient(Gm, hmo, g1, g2, self.nuc, self._n_occ, self.ncas,
                                       self._kap_row, self._kap_col, want_matrices=True)

    def _fock_ca(self, int1e_mo, int2e_mo, one_rdm, core, active):
        lib = _lib.load()
        N = self.nao
        g = self._t(int2e_mo)
        h = self._t(int1e_mo) if int1e_mo is not None else None
        gam = self._t(one_rdm) if one_rdm is not None else None
        FI = torch.empty((N, N), dtype=F64, device=self.device) if core else None
        FA = torch.empty((N, N), dtype=F64, device=self.device) if active else None
        _lib.check(lib.oovqe_fock_core_active(_lib.dptr(h), _lib.dptr(g), _lib.dptr(gam), N,
                                              self._n_occ, self.ncas, _lib.dptr(FI), _lib.dptr(FA),
                                              _lib.stream_ptr()), "oovqe_fock_core_active")
        return FI, FA

    def fock_core(self, int1e_mo, int2e_mo):
        """oo_energy.py:272-284: F^I_mn = h_mn + sum_i (2 g_mnii - g_miin)"""
        return self._fock_ca(int1e_mo, int2e_mo, None, True, False)[0]

    def fock_active(self, int2e_mo, one_rdm):
        """oo_energy.py:286-298: F^A_mn = sum_vw gamma_vw (g_mnvw - 1/2 g_mwvn)"""
        return self._fock_ca(None, int2e_mo, one_rdm, False, True)[1]

    def y_matrix(self, int2e_mo, two_full):
        """oo_energy.py:381-393: Y_pqrs = sum_mn [(G_pmrn + G_pmnr) g_qmns + G_prmn g_qsmn] for an
        arbitrary dense two_full (API helper): three N^2 x N^2 x N^2 products on the fp64 MFMA
        contraction kernel after index permutations (data movement only).  The Hessian path itself
        uses the block structure of the full-space RDM and never forms these N^4 operands."""
        n = self.nao
        g = self._t(int2e_mo)
        G = self._t(two_full)
        n2 = n * n
        a01 = (G.permute(0, 2, 1, 3) + G.permute(0, 3, 1, 2)).reshape(n2, n2).contiguous()  # [(p,r),(m,n)]
        b0 = g.permute(1, 2, 0, 3).reshape(n2, n2).contiguous()                              # [(m,n),(q,s)]
        a2 = G.reshape(n2, n2)                                                               # [(p,r),(m,n)]
        b2 = g.permute(2, 3, 0, 1).reshape(n2, n2).contiguous()                              # [(m,n),(q,s)]
        y = ops.matmul_nn(a01, b0) + ops.matmul_nn(a2.contiguous(), b2)     # [(p,r),(q,s)]
        return y.reshape(n, n, n, n).permute(0, 2, 1, 3).contiguous()

    def fock_generalized(self, int1e_mo, int2e_mo, one_rdm, two_rdm):
        """oo_energy.py:238-270: generalized Fock matrix from MO integrals (only the slices
        g_mo[:, :M, :M, :M] and h_mo[:, :M] are ever read)."""
        return self._fock_from_integrals(int1e_mo, int2e_mo, one_rdm, two_rdm)["fock"]

    def analytic_gradient_from_integrals(self, int1e_mo, int2e_mo, one_rdm, two_rdm):
        """oo_energy.py:300-309: G = 2 (F - F^T)"""
        return self._fock_from_integrals(int1e_mo, int2e_mo, one_rdm, two_rdm)["gmat"]

    def analytic_gradient(self, one_rdm, two_rdm, mo_coeff=None):
        """oo_energy.py:404-413 (the transform is redone on every call, as in the reference, but
        only the N M^3 block the gradient reads is formed)."""
        if mo_coeff is None:
            mo_coeff = self.mo_coeff
        g1, g2 = self._rdm_stack(one_rdm, two_rdm)
        return self._cas_eval(mo_coeff, g1, g2, want_matrices=True)["gmat"]

    # ---- Hessian ------------------------------------------------------------------------------------
    def full_rdms(self, one_rdm, two_rdm):
        """oo_energy.py:342-379: dense full-space RDMs (API helper; the Hessian kernels use the
        closed-form block structure directly and never build these).  Index scatter of constants and
        of the active RDMs only."""
        n, occ, act = self.nao, self.occ_idx, self.act_idx
        no = len(occ)
        g1, g2 = self._t(one_rdm), self._t(two_rdm)
        one_full = torch.zeros((n, n), dtype=F64, device=self.device)
        two_full = torch.zeros((n, n, n, n), dtype=F64, device=self.device)
        eye = torch.eye(no, dtype=F64, device=self.device)
        o = torch.as_tensor(occ, device=self.device)
        a = torch.as_tensor(act, device=self.device)
        one_full[o, o] = 2.0
        one_full[a[:, None], a[None, :]] = g1
        ix = lambda *v: tuple(x.reshape([-1 if i == k else 1 for i in range(4)])  # noqa: E731
                              for k, x in enumerate(v))
        two_full[ix(o, o, o, o)] = (4 * torch.einsum('ij,kl->ijkl', eye, eye)
                                    - 2 * torch.einsum('il,jk->ijkl', eye, eye))
        two_full[ix(o, o, a, a)] = 2 * torch.einsum('wv,ij->ijwv', g1, eye)
        two_full[ix(a, a, o, o)] = 2 * torch.einsum('wv,ij->wvij', g1, eye)
        two_full[ix(o, a, a, o)] = -torch.einsum('wv,ij->iwvj', g1, eye)
        two_full[ix(a, o, o, a)] = -torch.einsum('wv,ij->vjiw', g1, eye)
        two_full[ix(a, a, a, a)] = g2
        return one_full, two_full

    def _hessian(self, mo_coeff, one_rdm, two_rdm, g_ao=None, h_ao=None, want_matrix=False,
                 want_full=True):
        g_ao = self.int2e_ao if g_ao is None else g_ao
        h_ao = self.int1e_ao if h_ao is None else h_ao
        C = self._t(mo_coeff)
        g1, g2 = self._rdm_stack(one_rdm, two_rdm)
        res = ops.cas_eval(g_ao, h_ao, C, g1, g2, self.nuc, self._n_occ, self.ncas, self._kap_row,
                           self._kap_col, want_matrices=True,
                           eri_flags=self._eri_flags() if g_ao is self.int2e_ao else 0)
        return ops.orbital_hessian(g_ao, h_ao, C, g1[0].contiguous(), g2[0].contiguous(),
                                   res["fock"], self._n_occ, self.ncas, self._kap_row, self._kap_col,
                                   want_matrix=want_matrix, want_full=want_full)

    def analytic_hessian_from_integrals(self, int1e_mo, int2e_mo, one_rdm, two_rdm):
        """oo_energy.py:311-340: full [N,N,N,N] orbital Hessian from MO integrals (evaluated by the
        same kernels with the identity as orbital matrix)."""
        eye = torch.eye(self.nao, dtype=F64, device=self.device)
        return self._hessian(eye, one_rdm, two_rdm, g_ao=self._t(int2e_mo), h_ao=self._t(int1e_mo))[1]

    def analytic_hessian(self, one_rdm, two_rdm, mo_coeff=None):
        """oo_energy.py:415-424: full [N,N,N,N] orbital Hessian (``oovqe_orbital_hessian``)."""
        if mo_coeff is None:
            mo_coeff = self.mo_coeff
        return self._hessian(mo_coeff, one_rdm, two_rdm)[1]

    def analytic_hessian_matrix(self, one_rdm, two_rdm, mo_coeff=None):
        """Extension: the non-redundant [n_kappa, n_kappa] Hessian directly (what
        ``full_hessian_to_matrix(analytic_hessian(...))`` returns, without the N^4 tensor)."""
        if mo_coeff is None:
            mo_coeff = self.mo_coeff
        return self._hessian(mo_coeff, one_rdm, two_rdm, want_matrix=True, want_full=False)[0]

    def full_hessian_to_matrix(self, full_hess):
        """oo_energy.py:395-402 (index gathers)"""
        r, c = self._kap_row.long(), self._kap_col.long()
        return full_hess[r, c, :, :][:, r, c]

    def orbital_optimization(self, one_rdm, two_rdm, conv_tol=1e-8, max_iterations=100, verbose=0,
                             **kwargs):
        """oo_energy.py:426-474: damped-Newton orbital optimisation for fixed RDMs."""
        objective_fn = partial(self.energy_from_kappa, one_rdm=one_rdm, two_rdm=two_rdm)
        opt = NewtonStep(verbose=verbose, **kwargs)
        energy_l = []
        if verbose:
            energy = self.energy_from_mo_coeff(self.mo_coeff, one_rdm, two_rdm).item()
            print(f"Starting energy: {energy:.12f}")
        for n in range(max_iterations):
            kappa = torch.zeros(self.n_kappa, dtype=F64, device=self.device)
            gradient = self.kappa_matrix_to_vector(self.analytic_gradient(one_rdm, two_rdm))
            hessian = self.analytic_hessian_matrix(one_rdm, two_rdm)
            # (the reference's loop does not use the eigenvalue: it is never joined, an iteration does not wait for it)
            kappa, lowest_eigenvalue = opt.damped_newton_step(objective_fn, (kappa,), gradient,
                                                              hessian, defer_lowest=True)
            self.oao_mo_coeff = ops.matmul_nn(self._t(self.oao_mo_coeff),
                                              self.kappa_to_mo_coeff(kappa))
            energy = self.energy_from_mo_coeff(self.mo_coeff, one_rdm, two_rdm).item()
            energy_l.append(energy)
            if verbose is not None:
                print(f"iter = {n:03}, energy = {energy:.12f}")
            if n > 1:
                if abs(energy_l[-1] - energy_l[-2]) < conv_tol:
                    if verbose:
                        print("Orbital optimization finished.")
                        print("E_fin =", energy_l[-1])
                    break
        return energy_l
