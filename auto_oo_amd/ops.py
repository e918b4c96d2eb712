"""Thin torch-tensor wrappers over the C ABI (include/oovqe.h).

Each function allocates its outputs with torch (device memory + stream plumbing only), passes raw
device pointers to liboovqe_hip.so and returns torch tensors.  No arithmetic happens here and
there is no CPU path: every call needs a HIP device.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check, dptr, stream_ptr

F64 = torch.float64


def profile_begin(detail=False):
    """Start bracketing every half-transform launch with HIP events (bench.py roofline);
    ``detail=True`` brackets the other launches of an evaluation as well (PROFILE_LABELS)."""
    if detail:
        _lib.load().oovqe_profile_begin_detail()
    else:
        _lib.load().oovqe_profile_begin()


PROFILE_LABELS = ("half_transform", "circuit_rdms", "contract_p_to_n", "column", "final", "sym_q_contract")


def profile_end():
    """-> (half-transform kernel milliseconds, launches, {label: (ms, launches)})"""
    n = len(PROFILE_LABELS)
    ms = (ctypes.c_double * n)()
    cnt = (ctypes.c_int * n)()
    _lib.load().oovqe_profile_end_labels(ms, cnt, n)
    by = {PROFILE_LABELS[i]: (ms[i], cnt[i]) for i in range(n)}
    return ms[0], cnt[0], by


def _dev(t):
    if not t.is_cuda:
        raise _lib.OovqeError("auto_oo_amd ops need CUDA(HIP) tensors; no CPU fallback exists")
    return t.device


def as_device(x, device=None):
    """numpy / torch (any device) -> contiguous fp64 tensor on the HIP device."""
    dev = device if device is not None else _lib.require_device()
    if isinstance(x, torch.Tensor):
        return x.detach().to(device=dev, dtype=F64).contiguous()
    return torch.as_tensor(np.asarray(x, dtype=np.float64), dtype=F64, device=dev).contiguous()


def mode_contract(T, Cm, A, K, J, B, last, out=None):
    lib = _lib.load()
    dev = _dev(T)
    if out is None:
        out = torch.empty(A * J * B, dtype=F64, device=dev)
    ldc = Cm.shape[-1]
    check(lib.oovqe_mode_contract(dptr(T), dptr(Cm), dptr(out), A, K, J, B, ldc, int(bool(last)),
                                  stream_ptr()), "oovqe_mode_contract")
    return out


def matmul_nn(A, B):
    """A @ B"""
    lib = _lib.load()
    M, K = A.shape
    K2, N = B.shape
    assert K == K2
    out = torch.empty((M, N), dtype=F64, device=_dev(A))
    check(lib.oovqe_matmul_nn(dptr(A), dptr(B), M, K, N, dptr(out), stream_ptr()),
          "oovqe_matmul_nn")
    return out


def matmul_nn_batch(A, B, out=None):
    """A[b] @ B[b] for every b of a stack [G, M, K] x [G, K, N] in one launch"""
    lib = _lib.load()
    G, M, K = A.shape
    G2, K2, N = B.shape
    assert K == K2 and G == G2
    if out is None:
        out = torch.empty((G, M, N), dtype=F64, device=_dev(A))
    check(lib.oovqe_matmul_nn_batch(dptr(A), dptr(B), M, K, N, G, dptr(out), stream_ptr()),
          "oovqe_matmul_nn_batch")
    return out


def matmul_tn(A, B):
    """A.T @ B"""
    lib = _lib.load()
    K, M = A.shape
    K2, N = B.shape
    assert K == K2
    out = torch.empty((M, N), dtype=F64, device=_dev(A))
    check(lib.oovqe_matmul_tn(dptr(A), dptr(B), M, K, N, dptr(out), stream_ptr()),
          "oovqe_matmul_tn")
    return out


def general_4index_transform(M, C0, C1, C2, C3, out=None, work=None):
    lib = _lib.load()
    N = M.shape[0]
    dev = _dev(M)
    if out is None:
        out = torch.empty((N, N, N, N), dtype=F64, device=dev)
    if work is None:
        work = torch.empty((N, N, N, N), dtype=F64, device=dev)
    check(lib.oovqe_general_4index_transform(dptr(M), dptr(C0), dptr(C1), dptr(C2), dptr(C3), N,
                                             dptr(out), dptr(work), stream_ptr()),
          "oovqe_general_4index_transform")
    return out


def expm_skew(kappa, kap_row, kap_col, N, want_K=False):
    lib = _lib.load()
    dev = _dev(kappa)
    U = torch.empty((N, N), dtype=F64, device=dev)
    K = torch.empty((N, N), dtype=F64, device=dev) if want_K else None
    work = torch.empty(7 * N * N, dtype=F64, device=dev) if N > 48 else None
    check(lib.oovqe_expm_skew(dptr(kappa), dptr(kap_row, torch.int32), dptr(kap_col, torch.int32),
                              kappa.numel(), N, dptr(K), dptr(U), dptr(work), stream_ptr()),
          "oovqe_expm_skew")
    return (U, K) if want_K else U


def expm(X, sign=1.0):
    lib = _lib.load()
    N = X.shape[0]
    dev = _dev(X)
    U = torch.empty((N, N), dtype=F64, device=dev)
    work = torch.empty(6 * N * N, dtype=F64, device=dev) if N > 48 else None
    check(lib.oovqe_expm(dptr(X), float(sign), N, dptr(U), dptr(work), stream_ptr()), "oovqe_expm")
    return U


def circuit_state(theta, gates_dev, n_gates, n_qubits, init_index, tangents=False):
    """theta [batch, n_theta] -> psi [batch, D] (and dpsi [batch, n_theta, D])."""
    lib = _lib.load()
    dev = _dev(theta)
    batch, n_theta = theta.shape
    D = 1 << n_qubits
    psi = torch.empty((batch, D), dtype=F64, device=dev)
    dpsi = torch.empty((batch, n_theta, D), dtype=F64, device=dev) if tangents else None
    check(lib.oovqe_circuit_state(dptr(theta), n_theta, dptr(gates_dev, torch.uint8), n_gates,
                                  n_qubits, ctypes.c_uint32(init_index), batch, dptr(psi),
                                  dptr(dpsi), stream_ptr()), "oovqe_circuit_state")
    return (psi, dpsi) if tangents else psi


def rdms(bra, ket, ncas):
    """bra, ket [batch, D] -> gamma [batch, a, a], Gamma [batch, a, a, a, a] (transition RDMs)."""
    lib = _lib.load()
    dev = _dev(bra)
    batch, D = bra.shape
    n_qubits = 2 * ncas
    assert D == 1 << n_qubits and ket.shape == bra.shape
    gamma = torch.empty((batch, ncas, ncas), dtype=F64, device=dev)
    Gamma = torch.empty((batch, ncas, ncas, ncas, ncas), dtype=F64, device=dev)
    work = torch.empty(batch * 2 * ncas * ncas * D, dtype=F64, device=dev)
    check(lib.oovqe_rdms(dptr(bra), dptr(ket), n_qubits, ncas, batch, dptr(gamma), dptr(Gamma),
                         dptr(work), stream_ptr()), "oovqe_rdms")
    return gamma, Gamma


def spin_rdms(bra, ket, n_qubits):
    """bra, ket [batch, D] -> spin-orbital gamma [batch, n, n], Gamma [batch, n, n, n, n]
    (oovqe_spin_rdms: the reference's restricted=False RDMs)."""
    lib = _lib.load()
    dev = _dev(bra)
    batch, D = bra.shape
    assert D == 1 << n_qubits and ket.shape == bra.shape
    n = n_qubits
    gamma = torch.empty((batch, n, n), dtype=F64, device=dev)
    Gamma = torch.empty((batch, n, n, n, n), dtype=F64, device=dev)
    check(lib.oovqe_spin_rdms(dptr(bra), dptr(ket), n, batch, dptr(gamma), dptr(Gamma), stream_ptr()),
          "oovqe_spin_rdms")
    return gamma, Gamma


def rdms_tangent(psi, dpsi, ncas):
    """psi [batch, D], dpsi [batch, n_tan, D] (or None) -> gamma [batch, 1+n_tan, a, a],
    Gamma [batch, 1+n_tan, a, a, a, a]: set 0 = RDMs, set k = d/dtheta_k."""
    lib = _lib.load()
    dev = _dev(psi)
    batch, D = psi.shape
    n_tan = 0 if dpsi is None else dpsi.shape[1]
    nvec = 1 + n_tan
    gamma = torch.empty((batch, nvec, ncas, ncas), dtype=F64, device=dev)
    Gamma = torch.empty((batch, nvec, ncas, ncas, ncas, ncas), dtype=F64, device=dev)
    work = torch.empty(batch * nvec * ncas * ncas * D, dtype=F64, device=dev)
    check(lib.oovqe_rdms_tangent(dptr(psi), dptr(dpsi), 2 * ncas, ncas, n_tan, batch, dptr(gamma),
                                 dptr(Gamma), dptr(work), stream_ptr()), "oovqe_rdms_tangent")
    return gamma, Gamma


def circuit_rdms(theta, gates_dev, n_gates, n_qubits, ncas, init_index, tangents=True,
                 want_states=False):
    """theta [batch, n_theta] -> gamma [batch, 1+n_tan, a, a], Gamma [batch, 1+n_tan, a,a,a,a]
    (and psi, dpsi when want_states) through oovqe_circuit_rdms."""
    lib = _lib.load()
    dev = _dev(theta)
    batch, n_theta = theta.shape
    D = 1 << n_qubits
    n_tan = n_theta if tangents else 0
    nvec = 1 + n_tan
    small = bool(lib.oovqe_circuit_rdms_is_small(n_qubits, ncas, nvec, n_gates))
    gamma = torch.empty((batch, nvec, ncas, ncas), dtype=F64, device=dev)
    Gamma = torch.empty((batch, nvec, ncas, ncas, ncas, ncas), dtype=F64, device=dev)
    need_states = want_states or not small
    psi = torch.empty((batch, D), dtype=F64, device=dev) if need_states else None
    dpsi = (torch.empty((batch, n_theta, D), dtype=F64, device=dev)
            if (need_states and tangents) else None)
    work = None if small else torch.empty(batch * nvec * ncas * ncas * D, dtype=F64, device=dev)
    check(lib.oovqe_circuit_rdms(dptr(theta), n_theta, dptr(gates_dev, torch.uint8), n_gates,
                                 n_qubits, ncas, ctypes.c_uint32(init_index), int(bool(tangents)),
                                 batch, dptr(psi), dptr(dpsi), dptr(gamma), dptr(Gamma), dptr(work),
                                 stream_ptr()), "oovqe_circuit_rdms")
    if want_states:
        return gamma, Gamma, psi, dpsi
    return gamma, Gamma


def cas_half_transform(g_ao, C, M, out=None):
    lib = _lib.load()
    N = C.shape[0]
    if out is None:
        out = torch.empty((N, N, M, M), dtype=F64, device=_dev(g_ao))
    check(lib.oovqe_cas_half_transform(dptr(g_ao), dptr(C), N, M, dptr(out), stream_ptr()),
          "oovqe_cas_half_transform")
    return out


def cas_finish_transform(T2, h_ao, C, M, Gm=None, hmo=None, work=None):
    lib = _lib.load()
    N = C.shape[0]
    dev = _dev(T2)
    if Gm is None:
        Gm = torch.empty((N, M, M, M), dtype=F64, device=dev)
    if hmo is None:
        hmo = torch.empty((N, M), dtype=F64, device=dev)
    if work is None:
        work = torch.empty(N * M * M * M + N * M, dtype=F64, device=dev)
    check(lib.oovqe_cas_finish_transform(dptr(T2), dptr(h_ao), dptr(C), N, M, dptr(Gm), dptr(hmo),
                                         dptr(work), stream_ptr()), "oovqe_cas_finish_transform")
    return Gm, hmo


def cas_energy_gradient(Gm, hmo, gamma, Gamma, nuc, n_occ, ncas, kap_row, kap_col,
                        want_matrices=True, one_workgroup=False):
    """gamma [nrdm, a, a], Gamma [nrdm, a, a, a, a]; set 0 = the RDMs, sets >= 1 = derivative RDMs.
    Returns dict(c0, c1, c2, E, fock, gmat, gvec [nrdm, n_kappa], dE [nrdm-1]).
    ``one_workgroup``: the round-1 entry point oovqe_cas_energy_gradient (no scratch, one workgroup
    per RDM set) instead of oovqe_cas_energy_gradient_ws."""
    lib = _lib.load()
    dev = _dev(Gm)
    N = Gm.shape[0]
    nrdm = gamma.shape[0]
    n_kappa = kap_row.numel()
    c0 = torch.empty(1, dtype=F64, device=dev)
    E = torch.empty(1, dtype=F64, device=dev)
    c1 = torch.empty((ncas, ncas), dtype=F64, device=dev)
    c2 = torch.empty((ncas,) * 4, dtype=F64, device=dev)
    fock = torch.empty((N, N), dtype=F64, device=dev) if want_matrices else None
    gmat = torch.empty((N, N), dtype=F64, device=dev) if want_matrices else None
    gvec = torch.empty((nrdm, n_kappa), dtype=F64, device=dev)
    dE = torch.empty(max(nrdm - 1, 1), dtype=F64, device=dev)
    if one_workgroup:
        check(lib.oovqe_cas_energy_gradient(dptr(Gm), dptr(hmo), dptr(gamma), dptr(Gamma), nrdm,
                                            float(nuc), N, n_occ, ncas, dptr(kap_row, torch.int32),
                                            dptr(kap_col, torch.int32), n_kappa, dptr(c0), dptr(c1),
                                            dptr(c2), dptr(E), dptr(fock), dptr(gmat), dptr(gvec),
                                            dptr(dE), stream_ptr()), "oovqe_cas_energy_gradient")
        return dict(c0=c0, c1=c1, c2=c2, E=E, fock=fock, gmat=gmat, gvec=gvec, dE=dE[:nrdm - 1])
    # a workgroup per (general index, RDM set) + the assembly launch
    work = torch.empty(int(lib.oovqe_cas_energy_gradient_work_size(N, n_occ, ncas, nrdm)), dtype=F64, device=dev)
    check(lib.oovqe_cas_energy_gradient_ws(dptr(Gm), dptr(hmo), dptr(gamma), dptr(Gamma), nrdm,
                                           float(nuc), N, n_occ, ncas, dptr(kap_row, torch.int32),
                                           dptr(kap_col, torch.int32), n_kappa, dptr(c0), dptr(c1),
                                           dptr(c2), dptr(E), dptr(fock), dptr(gmat), dptr(gvec),
                                           dptr(dE), dptr(work), stream_ptr()), "oovqe_cas_energy_gradient_ws")
    return dict(c0=c0, c1=c1, c2=c2, E=E, fock=fock, gmat=gmat, gvec=gvec, dE=dE[:nrdm - 1])


ERI_PQ_SYMMETRIC = 1   # include/oovqe.h: OOVQE_ERI_PQ_SYMMETRIC
ERI_RS_SYMMETRIC = 2   # include/oovqe.h: OOVQE_ERI_RS_SYMMETRIC


def eri_flags(g_ao):
    """eri_flags for a resident g_ao ([N,N,N,N] or a stack [G,N,N,N,N]): ERI_PQ_SYMMETRIC when
    g[p,q,:,:] == g[q,p,:,:] and ERI_RS_SYMMETRIC when g[p,q,r,s] == g[p,q,s,r], bit for bit
    (oovqe_eri_symmetry_flags; one pass over the tensor, synchronises the stream).  The caller must
    not modify g_ao afterwards."""
    lib = _lib.load()
    _dev(g_ao)
    if g_ao.dtype != F64 or not g_ao.is_contiguous() or g_ao.dim() not in (4, 5):
        return 0
    N = g_ao.shape[-1]
    batch = g_ao.shape[0] if g_ao.dim() == 5 else 1
    if tuple(g_ao.shape[-4:]) != (N, N, N, N) or batch > 65535 or N > 65535:
        return 0
    if N <= 48:
        # the one-pass ingest kernel without its copy: every slab read once, the r <-> s test through LDS
        both = ERI_PQ_SYMMETRIC | ERI_RS_SYMMETRIC
        for f in eri_ingest(g_ao, pack=False)[0]:
            both &= f
        return both
    flags = ctypes.c_uint(0)
    check(lib.oovqe_eri_symmetry_flags(dptr(g_ao), N, batch, ctypes.byref(flags), stream_ptr()),
          "oovqe_eri_symmetry_flags")
    return int(flags.value)


def eri_ingest(g_ao, pack=True, out=None):
    """Ingest of resident integrals ([N,N,N,N] or a stack [G,N,N,N,N], contiguous fp64) in one library call
    (oovqe_eri_ingest; N <= 48: ONE pass over the tensor): -> (per-geometry flags [G] as a list of ints, packed
    copy [G, size] / [size] or None).  The copy of a geometry is meaningful when its flags carry both bits.
    ``out``: write the copy there.  Synchronises the stream."""
    lib = _lib.load()
    dev = _dev(g_ao)
    if g_ao.dtype != F64 or not g_ao.is_contiguous() or g_ao.dim() not in (4, 5):
        raise _lib.OovqeError("eri_ingest: g_ao must be a contiguous fp64 [N,N,N,N] tensor or a stack of them")
    N = g_ao.shape[-1]
    G = g_ao.shape[0] if g_ao.dim() == 5 else 1
    packed = None
    if pack:
        size = int(lib.oovqe_eri_packed_size(N))
        if size > 0:
            packed = out if out is not None else torch.empty((G, size) if g_ao.dim() == 5 else (size,), dtype=F64,
                                                             device=dev)
    flags = (ctypes.c_uint * G)()
    check(lib.oovqe_eri_ingest(dptr(g_ao), N, G, dptr(packed), flags, stream_ptr()), "oovqe_eri_ingest")
    return [int(f) for f in flags], packed


def eri_pack(g_ao):
    """The packed resident copy of integrals that carry BOTH symmetry flags (oovqe_eri_pack; include/oovqe.h):
    [N,N,N,N] -> [size] or a stack [G,N,N,N,N] -> [G, size]; None when this N has no packed form."""
    lib = _lib.load()
    dev = _dev(g_ao)
    N = g_ao.shape[-1]
    G = g_ao.shape[0] if g_ao.dim() == 5 else 1
    size = int(lib.oovqe_eri_packed_size(N))
    if size <= 0:
        return None
    out = torch.empty((G, size) if g_ao.dim() == 5 else (size,), dtype=F64, device=dev)
    check(lib.oovqe_eri_pack(dptr(g_ao), N, G, dptr(out), stream_ptr()), "oovqe_eri_pack")
    return out


def cas_eval(g_ao, h_ao, C, gamma, Gamma, nuc, n_occ, ncas, kap_row, kap_col, want_matrices=False,
             want_integrals=False, work=None, eri_flags=0, g_packed=None):
    """The whole CAS path (oovqe_cas_eval).  gamma [nrdm,a,a], Gamma [nrdm,a,a,a,a].
    eri_flags: see ops.eri_flags (0 = assume nothing about g_ao).  g_packed: ``eri_pack(g_ao)`` when the
    caller keeps one (both flags): stage 1 streams it where a kernel for it exists (N > 48)."""
    lib = _lib.load()
    dev = _dev(g_ao)
    N = C.shape[0]
    M = n_occ + ncas
    nrdm = gamma.shape[0]
    n_kappa = kap_row.numel()
    if work is None:
        work = torch.empty(lib.oovqe_cas_eval_work_size(N, n_occ, ncas, nrdm), dtype=F64, device=dev)
    # one packed output buffer: [c0 | E | dE (n_theta) | gvec (nrdm x n_kappa) | c1 | c2]; the slice
    # [E | dE | gvec[0]] is the energy followed by the full gradient, contiguous.
    n_t = max(nrdm - 1, 1)
    small = torch.empty(2 + n_t + nrdm * n_kappa + ncas * ncas + ncas ** 4, dtype=F64, device=dev)
    c0, E = small[0:1], small[1:2]
    o = 2
    dE = small[o:o + n_t]; o += nrdm - 1 if nrdm > 1 else 1
    gvec = small[o:o + nrdm * n_kappa].view(nrdm, n_kappa); o += nrdm * n_kappa
    c1 = small[o:o + ncas * ncas].view(ncas, ncas); o += ncas * ncas
    c2 = small[o:o + ncas ** 4].view((ncas,) * 4)
    fock = torch.empty((N, N), dtype=F64, device=dev) if want_matrices else None
    gmat = torch.empty((N, N), dtype=F64, device=dev) if want_matrices else None
    Gm = torch.empty((N, M, M, M), dtype=F64, device=dev) if want_integrals else None
    hmo = torch.empty((N, M), dtype=F64, device=dev) if want_integrals else None
    args = (dptr(g_ao), dptr(h_ao), dptr(C), dptr(gamma), dptr(Gamma), nrdm, float(nuc), N, n_occ, ncas,
            dptr(kap_row, torch.int32), dptr(kap_col, torch.int32), n_kappa, dptr(work), dptr(c0), dptr(c1),
            dptr(c2), dptr(E), dptr(gvec), dptr(dE), dptr(fock), dptr(gmat), dptr(Gm), dptr(hmo), int(eri_flags))
    if g_packed is not None:
        check(lib.oovqe_cas_eval_packed(*args, dptr(g_packed), stream_ptr()), "oovqe_cas_eval_packed")
    else:
        check(lib.oovqe_cas_eval(*args, stream_ptr()), "oovqe_cas_eval")
    packed = small[1:2 + (nrdm - 1) + n_kappa] if nrdm > 1 else None   # [E, dE..., gvec[0]...]
    return dict(c0=c0, c1=c1, c2=c2, E=E, fock=fock, gmat=gmat, gvec=gvec, dE=dE[:nrdm - 1],
                Gm=Gm, hmo=hmo, packed=packed)


class OoEvalPlan:
    """Pre-resolved arguments + persistent workspace for oovqe_oo_eval (one geometry, one circuit).
    Calling it costs one small allocation (the packed result) and one ctypes call."""

    def __init__(self, gates_dev, n_gates, n_theta, n_qubits, init_index, g_ao, h_ao, nuc, n_occ,
                 ncas, kap_row, kap_col, derivatives=True, eri_flags=0, g_packed=None):
        self.lib = _lib.load()
        self.eri_flags = int(eri_flags)
        # g_packed (eri_pack(g_ao), both flags): the call goes through oovqe_oo_eval_batch with a batch of one,
        # the entry point that takes the packed copy
        self.g_packed = g_packed
        self.dev = _dev(g_ao)
        self.N = h_ao.shape[0]
        self.n_theta, self.ncas, self.derivatives = n_theta, ncas, bool(derivatives)
        self.n_kappa = kap_row.numel()
        self.nvec = 1 + n_theta if derivatives else 1
        self.n_t = max(self.nvec - 1, 1)
        wsz = self.lib.oovqe_oo_eval_work_size(n_theta, n_gates, n_qubits, self.N, n_occ, ncas,
                                               int(self.derivatives))
        self.work = torch.empty(wsz, dtype=F64, device=self.dev)
        self.out_size = 2 + self.n_t + self.nvec * self.n_kappa + ncas ** 2 + ncas ** 4
        self._keep = (gates_dev, g_ao, h_ao, kap_row, kap_col)
        self._pre = (n_theta, ctypes.c_void_p(gates_dev.data_ptr()), n_gates, n_qubits,
                     ctypes.c_uint32(init_index), dptr(g_ao), dptr(h_ao))
        self._post = (float(nuc), self.N, n_occ, ncas, dptr(kap_row, torch.int32),
                      dptr(kap_col, torch.int32), self.n_kappa, int(self.derivatives),
                      dptr(self.work))
        if g_packed is not None:
            self._nuc_dev = torch.full((1,), float(nuc), dtype=F64, device=self.dev)
            self._post_b = (dptr(self._nuc_dev),) + self._post[1:8] + (1, dptr(self.work))

    def __call__(self, theta, C):
        """theta: contiguous fp64 device tensor [n_theta]; C: mo_coeff [N,N].  -> packed output."""
        out = torch.empty(self.out_size, dtype=F64, device=self.dev)
        if self.g_packed is not None:
            rc = self.lib.oovqe_oo_eval_batch(ctypes.c_void_p(theta.data_ptr()), *self._pre,
                                              ctypes.c_void_p(C.data_ptr()), *self._post_b,
                                              ctypes.c_void_p(out.data_ptr()), self.eri_flags,
                                              dptr(self.g_packed), stream_ptr())
            if rc != 0:
                check(rc, "oovqe_oo_eval_batch")
            return out
        rc = self.lib.oovqe_oo_eval(ctypes.c_void_p(theta.data_ptr()), *self._pre,
                                    ctypes.c_void_p(C.data_ptr()), *self._post,
                                    ctypes.c_void_p(out.data_ptr()), self.eri_flags, stream_ptr())
        if rc != 0:
            check(rc, "oovqe_oo_eval")
        return out

    def unpack(self, out):
        nk, a = self.n_kappa, self.ncas
        o = 2 + self.n_t
        gvec = out[o:o + self.nvec * nk].view(self.nvec, nk)
        o += self.nvec * nk
        c1 = out[o:o + a * a].view(a, a)
        c2 = out[o + a * a:o + a * a + a ** 4].view((a,) * 4)
        packed = out[1:2 + (self.nvec - 1) + nk] if self.nvec > 1 else None
        return dict(c0=out[0:1], E=out[1:2], dE=out[2:2 + self.nvec - 1], gvec=gvec, c1=c1, c2=c2,
                    packed=packed, fock=None, gmat=None, Gm=None, hmo=None)


def orbital_hessian(g_ao, h_ao, C, gamma, Gamma, fock, n_occ, ncas, kap_row, kap_col,
                    want_matrix=True, want_full=False):
    """Orbital-orbital Hessian (oovqe_orbital_hessian): [n_kappa, n_kappa] and/or [N,N,N,N]."""
    lib = _lib.load()
    dev = _dev(g_ao)
    N = C.shape[0]
    n_kappa = kap_row.numel()
    work = torch.empty(lib.oovqe_orbital_hessian_work_size(N, n_occ, ncas), dtype=F64, device=dev)
    Hm = torch.empty((n_kappa, n_kappa), dtype=F64, device=dev) if want_matrix else None
    Hf = torch.empty((N, N, N, N), dtype=F64, device=dev) if want_full else None
    check(lib.oovqe_orbital_hessian(dptr(g_ao), dptr(h_ao), dptr(C), dptr(gamma), dptr(Gamma),
                                    dptr(fock), N, n_occ, ncas, dptr(kap_row, torch.int32),
                                    dptr(kap_col, torch.int32), n_kappa, dptr(work), dptr(Hm),
                                    dptr(Hf), stream_ptr()), "oovqe_orbital_hessian")
    return Hm, Hf


_PAIR_TABLES = {}


def _hessian_pair_tables(n_theta, dev):
    """(pairs [n_pairs, 2] int32, j [n_pairs], k [n_pairs]) for j <= k, on `dev`, built once."""
    key = (n_theta, str(dev))
    tabs = _PAIR_TABLES.get(key)
    if tabs is None:
        pairs = [(j, k) for j in range(n_theta) for k in range(j, n_theta)]
        pairs_dev = torch.tensor(pairs, dtype=torch.int32, device=dev).contiguous()
        tabs = (pairs_dev, pairs_dev[:, 0].long().contiguous(), pairs_dev[:, 1].long().contiguous())
        _PAIR_TABLES[key] = tabs
    return tabs


_CHESS_WORK = {}


def circuit_hessian(theta, gates_dev, n_gates, n_qubits, ncas, init_index, c1, c2):
    """d^2E/dtheta^2 for E = c0 + c1.gamma(theta) + c2.Gamma(theta)  (oo_pqc.py:103-111):
    second tangents + transition RDMs + contraction in one library call (oovqe_circuit_hessian);
    theta [n_theta]."""
    lib = _lib.load()
    dev = _dev(theta)
    n_theta = theta.numel()
    th2 = theta.reshape(1, n_theta).contiguous()
    pairs_dev, _, _ = _hessian_pair_tables(n_theta, dev)   # cached: no host->device copy per call
    n_pairs = pairs_dev.shape[0]
    # scratch is cached per (shape, device, STREAM): calls on different HIP streams must not share it
    # (the caching allocator's stream ordering does not cover a buffer that outlives the call)
    key = (n_theta, n_qubits, ncas, str(dev), torch.cuda.current_stream().cuda_stream)
    work = _CHESS_WORK.get(key)
    if work is None:
        if len(_CHESS_WORK) >= 8:
            _CHESS_WORK.clear()
        work = torch.empty(lib.oovqe_circuit_hessian_work_size(n_theta, n_qubits, ncas, n_pairs), dtype=F64,
                           device=dev)
        _CHESS_WORK[key] = work
    H = torch.empty((n_theta, n_theta), dtype=F64, device=dev)
    check(lib.oovqe_circuit_hessian(dptr(th2), n_theta, dptr(gates_dev, torch.uint8), n_gates, n_qubits, ncas,
                                    ctypes.c_uint32(init_index), dptr(c1.contiguous()), dptr(c2.contiguous()),
                                    dptr(pairs_dev, torch.int32), n_pairs, dptr(work), dptr(H), stream_ptr()),
          "oovqe_circuit_hessian")
    return H


_NEWTON_WORK = {}
_NEWTON_SIDE = {}

NEWTON_INFO = {1: "direction from the Cholesky fast path", 0: "direction from the band route",
               2: "direction from the fast path; its lowest eigenvalue could not be computed",
               -1: "a hand-off between the workgroups of a problem timed out (co-residency not granted)",
               -2: "indefinite Hessian without level shift beyond the pivoted kernel (n > 480)",
               -3: "the Hessian holds a NaN or an Inf"}


def _newton_work(kind, n, G, dev, size):
    # scratch is cached per (kind, shape, device, STREAM): calls on different HIP streams must not share it
    key = (kind, n, G, str(dev), torch.cuda.current_stream().cuda_stream)
    work = _NEWTON_WORK.get(key)
    if work is None:
        if len(_NEWTON_WORK) >= 12:
            _NEWTON_WORK.clear()
        work = torch.empty(int(size), dtype=F64, device=dev)
        _NEWTON_WORK[key] = work
    return work


def side_streams(dev=None):
    """The library's two side streams of a device (created once).  A process should not create many more streams than
    these: HIP multiplexes its streams over a handful of hardware queues (four by default), and two streams that
    share a queue run one after the other -- callers that want a second stream of their own (e.g. two evaluation
    calls in flight, ``OO_pqc_batch.evaluate(slot=...)``) can take these."""
    dev = _lib.require_device() if dev is None else dev
    entry = _NEWTON_SIDE.get(str(dev))
    if entry is None:
        entry = _NEWTON_SIDE[str(dev)] = [[torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)], 0]
    return entry[0]


def _side_stream(dev):
    """Two side streams, taken in turn: the eigenvalue routes of consecutive steps run beside each other (a
    quarter of the chip each), so that a loop of steps is not paced by one of them."""
    side_streams(dev)
    entry = _NEWTON_SIDE[str(dev)]
    entry[1] ^= 1
    return entry[0][entry[1]]


class PendingTensor:
    """A result still being computed on one of the library's side streams (``OO_pqc_batch.evaluate_deferred``:
    independent evaluation calls in flight on two streams, the latency-bound tail of one under the N^4 sweep of the
    next).  ``result()`` makes the CURRENT stream wait for it and returns the tensor; until then nothing may read
    the tensor."""

    def __init__(self, tensor, event, view=None):
        self._tensor, self._event, self._view = tensor, event, view

    def result(self):
        if self._event is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(self._event)
            self._tensor.record_stream(cur)
            self._event = None
        return self._tensor if self._view is None else self._view(self._tensor)

    def wait(self):
        """Make the current stream wait for the computation WITHOUT taking the tensor (joining the last call of a side
        stream joins every call enqueued on that stream before it: a loop over many deferred calls needs one wait
        per stream, not one per call)."""
        if self._event is not None:
            torch.cuda.current_stream().wait_event(self._event)


class PendingLowest:
    """Lowest Hessian eigenvalues still being computed on the library's side stream (the band route runs
    beside the line search: the eigenvalue of a positive definite Hessian is a reported number that no step of
    the optimisation reads, newton_raphson.py:105-128).  ``result()`` makes the current stream wait for them
    and returns the tensor."""

    def __init__(self, tensor, event, retry=None):
        # retry: callable that recomputes the eigenvalues on the CALLING stream with one workgroup per problem
        # (no hand-off between workgroups left that could time out); used when the side route reports NaN
        self._tensor, self._event, self._retry = tensor, event, retry

    def result(self):
        if self._event is not None:
            torch.cuda.current_stream().wait_event(self._event)
            self._event = None
        return self._tensor

    def checked(self):
        """``result()`` with the values verified on the host: an eigenvalue the side route could not deliver (NaN: a
        hand-off between its workgroups timed out beside a kernel that held the chip) is computed again on the
        calling stream without any inter-workgroup wait.  The eigenvalue is a diagnostic no step reads
        (hess_eig_l of oo_pqc.py:155-207), so a missing one never discards an optimisation."""
        vals = self.result().reshape(-1).tolist()
        if any(v != v for v in vals) and self._retry is not None:
            retry, self._retry = self._retry, None
            retry()
        return self._tensor

    def tolist(self):
        """Host values; raises only if the eigenvalue is still missing after the one-workgroup repeat (a NaN / Inf
        in the Hessian itself)."""
        vals = self.checked().reshape(-1).tolist()
        if any(v != v for v in vals):
            raise _lib.OovqeError("lowest Hessian eigenvalue missing (NaN): the Hessian holds a NaN or an Inf")
        return vals

    def item(self):
        """The eigenvalue of ONE problem as a float (joins, reads back)."""
        (v,) = self.tolist()
        return v

    __float__ = item


def newton_direction(hessian, gradient, lambda_min=1e-6, mu=1e-6, rho=1.1, aug=True, defer_lowest=False,
                     want_info=False, max_wg=0):
    """dp = -(H + nu I)^-1 g with the reference's level shift (newton_raphson.py:78-129), on the device.
    hessian [n,n] or [G,n,n], gradient [n] or [G,n] -> (dp, lowest eigenvalues [G] (0-d for one problem),
    shifts nu[, info]).

    Positive definite Hessians (lowest eigenvalue above lambda_min: the reference does not shift) take their
    direction from a blocked Cholesky factorisation (``oovqe_newton_direction_pd``, one workgroup per
    factorisation), and their lowest eigenvalue -- which nothing downstream of the direction reads -- from the
    band-reduction route on a side stream; every other problem goes through the band route on the calling
    stream (``oovqe_newton_direction_rest``).  ``defer_lowest``: return the eigenvalues as a
    ``PendingLowest`` instead of making the calling stream wait for the side stream.  ``info`` [G] (see
    ``NEWTON_INFO``; negative = failed loudly, dp = NaN)."""
    lib = _lib.load()
    dev = _dev(hessian)
    single = hessian.dim() == 2
    H = hessian.reshape(-1, hessian.shape[-1], hessian.shape[-1])
    g = gradient.reshape(H.shape[0], -1)
    G, n = g.shape
    if n > lib.oovqe_newton_direction_max_n():
        raise _lib.OovqeError(f"newton_direction: n = {n} exceeds the band-reduction kernels "
                              f"({lib.oovqe_newton_direction_max_n()})")
    H = H if H.is_contiguous() else H.contiguous()
    g = g if g.is_contiguous() else g.contiguous()
    dpc = torch.empty((G, n), dtype=F64, device=dev)
    low = torch.empty(G, dtype=F64, device=dev)
    nu = torch.empty(G, dtype=F64, device=dev)
    info = torch.zeros(G, dtype=F64, device=dev)
    args = (float(lambda_min), float(mu), float(rho), int(bool(aug)))
    event = None
    if lib.oovqe_newton_direction_has_pd(n, int(bool(aug))) and G <= 32767:
        wpd = _newton_work("pd", n, G, dev, lib.oovqe_newton_direction_pd_work_size(n, G))
        check(lib.oovqe_newton_direction_pd(dptr(H), dptr(g), n, G, float(lambda_min), dptr(wpd), dptr(dpc),
                                            dptr(nu), dptr(info), stream_ptr()), "oovqe_newton_direction_pd")
        main = torch.cuda.current_stream()
        forked = torch.cuda.Event()
        forked.record(main)
        # the problems the fast path did not serve: here, on the calling stream (a no-op launch when there
        # are none); the lowest eigenvalue of the others: beside it
        wmain = _newton_work("rest", n, G, dev, lib.oovqe_newton_direction_rest_work_size(n, G))
        check(lib.oovqe_newton_direction_rest(dptr(H), dptr(g), n, G, *args, dptr(info), 1, int(max_wg), dptr(wmain),
                                              dptr(dpc), dptr(low), dptr(nu), stream_ptr()),
              "oovqe_newton_direction_rest")
        side = _side_stream(dev)
        # the eigenvalue route beside the line search keeps to a quarter of the chip (half for larger stacks,
        # whose band reduction would otherwise crawl on one workgroup per problem: its workgroups hold a
        # whole CU each for the length of the reduction; the evaluations of the line search need the rest,
        # and the route of the previous step may still be running on the other side stream)
        side_wg = (max(1, 64 // G) if G <= 16 else max(1, 128 // G)) if max_wg == 0 else int(max_wg)
        with torch.cuda.stream(side):
            side.wait_event(forked)
            wside = _newton_work("rest", n, G, dev, lib.oovqe_newton_direction_rest_work_size(n, G))
            check(lib.oovqe_newton_direction_rest(dptr(H), dptr(g), n, G, *args, dptr(info), 2, side_wg,
                                                  dptr(wside), dptr(dpc), dptr(low), dptr(nu), stream_ptr()),
                  "oovqe_newton_direction_rest")
            event = torch.cuda.Event()
            event.record(side)
        for t in (H, g, low, info, dpc, nu):
            t.record_stream(side)
    else:
        work = _newton_work("rest", n, G, dev, lib.oovqe_newton_direction_rest_work_size(n, G))
        check(lib.oovqe_newton_direction_rest(dptr(H), dptr(g), n, G, *args, dptr(info), 0, int(max_wg), dptr(work),
                                              dptr(dpc), dptr(low), dptr(nu), stream_ptr()),
              "oovqe_newton_direction_rest")
    if single:
        dpc, low, nu = dpc[0], low[0], nu[0]
    if defer_lowest:
        retry = None
        if event is not None:
            def retry(H=H, g=g, info=info, low_all=low if not single else low.reshape(1), dpc_all=dpc.reshape(G, n),
                      nu_all=nu.reshape(G)):
                # the eigenvalues of the problems the fast path served, once more: calling stream, ONE workgroup each
                w1 = _newton_work("rest", n, G, dev, lib.oovqe_newton_direction_rest_work_size(n, G))
                check(lib.oovqe_newton_direction_rest(dptr(H), dptr(g), n, G, *args, dptr(info), 2, 1, dptr(w1),
                                                      dptr(dpc_all), dptr(low_all), dptr(nu_all), stream_ptr()),
                      "oovqe_newton_direction_rest")
        low = PendingLowest(low, event, retry)
    elif event is not None:
        torch.cuda.current_stream().wait_event(event)
    if want_info:
        return dpc, low, nu, info
    return dpc, low, nu
