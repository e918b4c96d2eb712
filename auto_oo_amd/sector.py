"""Particle-number-sector statevector engine (host side): string tables + thin wrappers over
``oovqe_sector_*`` (include/oovqe.h).  Used by ``Parameterized_circuit`` for active spaces whose
2^n register does not fit the one-workgroup LDS kernel (n_qubits > 10), e.g. kUpCCD CAS(8e,8o)."""
import ctypes
from itertools import combinations

import numpy as np
import torch

from . import _lib
from ._lib import check, dptr, stream_ptr

F64 = torch.float64


def string_tables(ncas, n_occ_strings):
    """Occupation strings of ``n_occ_strings`` particles in ``ncas`` orbitals (orbital p at bit
    ncas-1-p), ascending by value, and the inverse map (string -> index, -1 elsewhere)."""
    strings = []
    for occ in combinations(range(ncas), n_occ_strings):
        m = 0
        for p in occ:
            m |= 1 << (ncas - 1 - p)
        strings.append(m)
    strings.sort()
    unrank = np.array(strings, dtype=np.uint32)
    rank = np.full(1 << ncas, -1, dtype=np.int32)
    rank[unrank] = np.arange(len(strings), dtype=np.int32)
    return unrank, rank


def sector_of(hfstate, ncas):
    """(N_alpha, N_beta) of an occupation vector in interleaved ordering (even wire = alpha)."""
    occ = np.asarray(hfstate, dtype=int)
    return int(occ[0::2].sum()), int(occ[1::2].sum())


class SectorEngine:
    def __init__(self, ncas, hfstate, gates_dev, n_gates, n_theta, init_index, device):
        self.lib = _lib.load()
        self.ncas, self.device = ncas, device
        self.n_alpha, self.n_beta = sector_of(hfstate, ncas)
        ua, ra = string_tables(ncas, self.n_alpha)
        ub, rb = string_tables(ncas, self.n_beta)
        self.na, self.nb = len(ua), len(ub)
        self.Dc = self.na * self.nb
        # uint32 strings are stored in int32 tensors (same bits; torch has no uint32 arithmetic)
        self.unrank_a = torch.as_tensor(ua.astype(np.int32)).to(device)
        self.unrank_b = torch.as_tensor(ub.astype(np.int32)).to(device)
        self.rank_a = torch.as_tensor(ra).to(device)
        self.rank_b = torch.as_tensor(rb).to(device)
        self.gates_dev, self.n_gates, self.n_theta = gates_dev, n_gates, n_theta
        self.init_index = init_index
        self._work = {}
        self._param_gate = None
        self._pairs = None
        self._tables_ptr = None
        self._hess_tables = None

    # ---- derivatives (round 4; parameters shared between gates: round 5) ---------------------------
    def param_gate_lists(self, gates_host):
        """The gates every parameter drives (UCCD / UCCSD / kUpCCD: one Givens pass per excitation; GateFabric's
        OrbitalRotation: two Givens rotations on the same angle, pqc.py:79-83 of the reference)."""
        if self._param_gate is None:
            owner = [[] for _ in range(self.n_theta)]
            for gi, g in enumerate(gates_host):
                if g.theta_idx >= 0:
                    owner[g.theta_idx].append(gi)
            self._param_gate = owner
        return self._param_gate

    def param_gates(self, gates_host):
        """gate index of every parameter when each drives exactly one gate (UCCD / UCCSD / kUpCCD); None otherwise."""
        gl = self.param_gate_lists(gates_host)
        return [g[0] for g in gl] if all(len(g) == 1 for g in gl) else None

    def _tangent_plan(self, gates_host, second):
        """Which differentiated circuits (``oovqe_sector_state_deriv`` specs: up to two gates differentiated) sum to
        the tangent states, by the product rule over the gates that share a parameter:
            d psi / d theta_k            = sum_{g in gates(k)} psi[g]
            d^2 psi / d theta_j d theta_k = sum_{g in gates(j)} sum_{h in gates(k)} psi[g, h]
        (psi[g, g] = the gate's own second derivative; for j = k the mixed terms g != h appear in both orders: weight 2).
        -> (specs [n_spec, 2] int32 device, owner [n_spec] int64 device: row of the output each spec adds to, weights
        [n_spec] device or None when all 1, n_out, pairs).  Row 0 = psi, rows 1..n_theta = first tangents, then the
        second tangents of the pairs j <= k in row-major order (``second=True``)."""
        key = bool(second)
        plans = self.__dict__.setdefault("_tangent_plans", {})
        if key not in plans:
            gl = self.param_gate_lists(gates_host)
            nt = self.n_theta
            specs, owner, weight = [(-1, -1)], [0], [1.0]
            for k in range(nt):
                for g in gl[k]:
                    specs.append((g, -1)); owner.append(1 + k); weight.append(1.0)
            pairs = []
            if second:
                pairs = [(j, k) for j in range(nt) for k in range(j, nt)]
                for row, (j, k) in enumerate(pairs):
                    if j == k:
                        gs = gl[j]
                        for a_, g in enumerate(gs):
                            specs.append((g, g)); owner.append(1 + nt + row); weight.append(1.0)
                            for h in gs[a_ + 1:]:
                                specs.append((g, h)); owner.append(1 + nt + row); weight.append(2.0)
                    else:
                        for g in gl[j]:
                            for h in gl[k]:
                                specs.append((g, h)); owner.append(1 + nt + row); weight.append(1.0)
            dev = self.device
            w = None if all(x == 1.0 for x in weight) else torch.as_tensor(weight, dtype=F64, device=dev)
            one_each = all(len(g_) == 1 for g_ in gl)                        # one gate per parameter: no sums at all
            plans[key] = (torch.as_tensor(np.asarray(specs, dtype=np.int32).reshape(-1, 2)).to(dev),
                          None if one_each else torch.as_tensor(owner, dtype=torch.int64, device=dev), w,
                          1 + nt + len(pairs), pairs)
        return plans[key]

    def tangent_states(self, theta, gates_host, second=False):
        """theta [batch, n_theta] -> [batch, 1 + n_theta (+ n_pairs), Dc]: psi, its first tangents and (``second``) its
        second tangents for the pairs j <= k, whatever the number of gates per parameter."""
        spec, owner, w, n_out, _ = self._tangent_plan(gates_host, second)
        st = self.derivative_states(theta, spec)
        if owner is None:
            return st
        if w is not None:
            st = st * w[None, :, None]
        out = torch.zeros((st.shape[0], n_out, self.Dc), dtype=F64, device=self.device)
        out.index_add_(1, owner, st)
        return out

    def derivative_states(self, theta, specs):
        """theta [batch, n_theta], specs: list of (gate_a, gate_b) (-1 = none) -> [batch, len(specs), Dc]:
        the circuit with those gates differentiated (oovqe_sector_state_deriv)."""
        batch = theta.shape[0]
        spec = (specs if isinstance(specs, torch.Tensor)
                else torch.as_tensor(np.asarray(specs, dtype=np.int32).reshape(-1, 2)).to(self.device))
        out = torch.empty((batch, spec.shape[0], self.Dc), dtype=F64, device=self.device)
        pairs, max_pairs = self.pair_lists()
        check(self.lib.oovqe_sector_state_deriv_pl(dptr(theta), self.n_theta, dptr(self.gates_dev, torch.uint8),
                                                   self.n_gates, self.ncas, ctypes.c_uint32(self.init_index),
                                                   *self._tabs(), batch, dptr(pairs, torch.int32), max_pairs,
                                                   dptr(spec, torch.int32), int(spec.shape[0]),
                                                   dptr(out), stream_ptr()), "oovqe_sector_state_deriv_pl")
        return out

    def rdms_chunked(self, states, chunk=256):
        """RDMs of a long list of sector vectors [n, Dc], ``chunk`` at a time (the workspace of
        ``oovqe_sector_rdms`` grows with the batch)."""
        g1, g2 = [], []
        for i in range(0, states.shape[0], chunk):
            a, b = self.rdms(states[i:i + chunk].contiguous())
            g1.append(a)
            g2.append(b)
        return torch.cat(g1), torch.cat(g2)

    def rdms_with_derivatives(self, theta, gates_host):
        """gamma [1 + n_theta, a, a], Gamma [1 + n_theta, a,a,a,a] of ONE parameter point theta [1, n_theta]:
        set 0 = the RDMs of psi, set k = their derivative with respect to theta_k.  The RDMs are quadratic
        forms of the (real) state, so with the tangent tau_k = d psi / d theta_k the derivative is EXACTLY
        [RDM(psi + tau_k) - RDM(psi - tau_k)] / 2 -- the plain RDM kernel on 2 n_theta + 1 vectors, no
        transition-RDM kernel.  (Parameters that drive several gates -- GateFabric's orbital rotations -- have their
        tangents summed over those gates: ``tangent_states``.)"""
        nt = self.n_theta
        st = self.tangent_states(theta, gates_host)[0]                              # [1 + nt, Dc]
        psi, tau = st[0:1], st[1:]
        g1, g2 = self.rdms_chunked(torch.cat((psi, psi + tau, psi - tau)))
        gamma = torch.cat((g1[0:1], 0.5 * (g1[1:1 + nt] - g1[1 + nt:])))
        Gamma = torch.cat((g2[0:1], 0.5 * (g2[1:1 + nt] - g2[1 + nt:])))
        return gamma, Gamma

    def lam(self, vecs, c1, c2):
        """vecs [n, Dc] -> (Hop + Hop^T) vecs [n, Dc], Hop the operator whose quadratic form is
        Q(v) = c1 . gamma(v) + c2 . Gamma(v) (oovqe_sector_lambda: the first stage of the adjoint on its own)."""
        n = vecs.shape[0]
        out = torch.empty((n, self.Dc), dtype=F64, device=self.device)
        self.pair_lists()
        check(self.lib.oovqe_sector_lambda(dptr(vecs), self.ncas, *self._tabs(), n, dptr(c1.contiguous()),
                                           dptr(c2.contiguous()), self._tables_ptr, dptr(self.work(n)), dptr(out),
                                           stream_ptr()),
              "oovqe_sector_lambda")
        return out

    def circuit_hessian(self, theta, gates_host, c1, c2, by_rdms=False):
        """d^2/dtheta^2 of E = c0 + c1 . gamma(theta) + c2 . Gamma(theta) = c0 + Q(psi), Q(v) = v^T Hop v the quadratic
        form of the active-space Hamiltonian (oo_pqc.py:103-111), inside the (N_alpha, N_beta) sector:
            H_jk = tau_jk^T lam(psi) + tau_j^T lam(tau_k),     lam(v) = (Hop + Hop^T) v
        with first and second tangent states from ``oovqe_sector_state_deriv`` -- 1 + n_theta applications of the
        operator (``oovqe_sector_lambda``) and two small products, where the polarisation of Q through the plain RDM
        kernel (``by_rdms=True``, the first form of round 4: [Q(a + b) - Q(a - b)] / 2 for every pair) took 4 n_pairs
        RDM evaluations: 6 384 sector vectors at CAS(8e,8o), k = 1."""
        nt = self.n_theta
        if self._hess_tables is None:      # (index tables of the pairs j <= k: once per engine, not per call)
            pairs = self._tangent_plan(gates_host, True)[4]
            self._hess_tables = (torch.as_tensor([j for j, _ in pairs], device=self.device),
                                 torch.as_tensor([k for _, k in pairs], device=self.device), len(pairs))
        st = self.tangent_states(theta, gates_host, second=True)[0]
        return self.circuit_hessian_from_states(st, gates_host, c1, c2, by_rdms)

    def hessian_pair_tables(self, gates_host):
        """(j indices, k indices, count) of the pairs j <= k in the order of the second tangents (device)."""
        if self._hess_tables is None:
            pairs = self._tangent_plan(gates_host, True)[4]
            self._hess_tables = (torch.as_tensor([j for j, _ in pairs], device=self.device),
                                 torch.as_tensor([k for _, k in pairs], device=self.device), len(pairs))
        return self._hess_tables

    def circuit_hessian_from_states(self, st, gates_host, c1, c2, by_rdms=False):
        """The theta-theta block from psi, its first and its second tangents (``tangent_states(.., second=True)[g]``,
        [1 + n_theta + n_pairs, Dc]) and the CAS coefficients of ONE geometry."""
        nt = self.n_theta
        ja, ka, npair = self.hessian_pair_tables(gates_host)
        psi, tau, tau2 = st[0], st[1:1 + nt], st[1 + nt:]
        if by_rdms:
            vecs = torch.cat((tau2 + psi, tau2 - psi, tau[ja] + tau[ka], tau[ja] - tau[ka]))
            g1, g2 = self.rdms_chunked(vecs)
            a = self.ncas
            q = ((g1.reshape(-1, a * a) * c1.reshape(1, -1)).sum(dim=1)
                 + (g2.reshape(-1, a ** 4) * c2.reshape(1, -1)).sum(dim=1))
            val = 0.5 * (q[:npair] - q[npair:2 * npair]) + 0.5 * (q[2 * npair:3 * npair] - q[3 * npair:])
        else:
            lam = self.lam(st[:1 + nt].contiguous(), c1, c2)                       # lam(psi), lam(tau_k)
            # two sets of scalar products over the sector dimension (elementwise multiply + sum: 1 596 + 56 x 56
            # rows of 4 900 -- the library's GEMM kernels have no shape for a few outputs of a long inner dimension)
            first = (tau2 * lam[0]).sum(dim=1)                                     # tau_jk . lam(psi)
            second = (tau[:, None, :] * lam[None, 1:, :]).sum(dim=2)               # [j, k] = tau_j . lam(tau_k)
            val = first + second[ja, ka]
        H = torch.zeros((nt, nt), dtype=F64, device=self.device)
        H[ja, ka] = val
        H[ka, ja] = val
        return H

    # ---- a stack of geometries, each with its own CAS coefficients (round 5) ---------------------------
    def geometry_coefficients_ok(self):
        """True when the library applies the operator / runs the reverse sweep for a stack of states with PER-GEOMETRY
        coefficients in one launch sequence (ncas = 4, 8); the callers loop over the geometries otherwise."""
        return bool(self.lib.oovqe_sector_geometry_coefficients_ok(self.ncas, self.na, self.nb))

    def adjoint_geometries(self, theta, psi_c, c1, c_stride):
        """d/dtheta (c1[g] . gamma + c2[g] . Gamma) of state g for a stack of geometries: theta [G, n_theta], psi_c
        [G, Dc]; c1 = a view whose element g starts c_stride doubles after element g - 1 and is followed by c2[g]
        (the c1 | c2 columns of ``oovqe_cas_eval_batch``'s packed outputs) -> [G, n_theta]."""
        G = theta.shape[0]
        a = self.ncas
        dth = torch.empty((G, self.n_theta), dtype=F64, device=self.device)
        pairs, max_pairs = self.pair_lists()
        p1 = c1.data_ptr()
        check(self.lib.oovqe_sector_adjoint_pg(dptr(theta), self.n_theta, dptr(self.gates_dev, torch.uint8),
                                               self.n_gates, a, *self._tabs(), G, dptr(psi_c), ctypes.c_void_p(p1),
                                               ctypes.c_void_p(p1 + 8 * a * a), int(c_stride), dptr(pairs, torch.int32),
                                               max_pairs, self._tables_ptr, dptr(self.work(G)), dptr(dth),
                                               stream_ptr()), "oovqe_sector_adjoint_pg")
        return dth

    def lam_geometries(self, vecs, group, c1, c_stride):
        """(Hop_g + Hop_g^T) v for vecs [G * group, Dc], geometry g = row // group, coefficients as in
        ``adjoint_geometries`` -> [G * group, Dc]."""
        n = vecs.shape[0]
        a = self.ncas
        out = torch.empty((n, self.Dc), dtype=F64, device=self.device)
        self.pair_lists()
        p1 = c1.data_ptr()
        check(self.lib.oovqe_sector_lambda_pg(dptr(vecs), a, *self._tabs(), n, int(group), ctypes.c_void_p(p1),
                                              ctypes.c_void_p(p1 + 8 * a * a), int(c_stride), self._tables_ptr,
                                              dptr(self.work(n)), dptr(out), stream_ptr()), "oovqe_sector_lambda_pg")
        return out

    def pair_lists(self):
        """(pairs, max_pairs): the gates of the circuit as lists of the determinant pairs they rotate
        (oovqe_sector_pairs, include/oovqe.h) -- built once per engine; the forward and the reverse sweep then read
        a gate's pairs instead of finding them (mask test, two rank look-ups per determinant, per gate, per state).
        (None, 0) beyond 32 767 determinants."""
        if self._pairs is None:
            if self.Dc > 32767:
                self._pairs = (None, 0)
            else:
                n = int(self.lib.oovqe_sector_pairs_size(self.n_gates, self.na, self.nb))
                nt = int(self.lib.oovqe_sector_tables_size(self.ncas, self.na, self.nb))
                pairs = torch.empty(n + nt, dtype=torch.int32, device=self.device)
                self._tables_ptr = ctypes.c_void_p(pairs.data_ptr() + 4 * n)   # the excitation tables behind the lists
                check(self.lib.oovqe_sector_pairs(dptr(self.gates_dev, torch.uint8), self.n_gates, self.ncas,
                                                  *self._tabs(), dptr(pairs, torch.int32), stream_ptr()),
                      "oovqe_sector_pairs")
                # (one readback when the engine is set up: the largest count sizes the sweeps' register arrays)
                self._pairs = (pairs, int(pairs[:self.n_gates].max().item()))
        return self._pairs

    def fits(self):
        """The sector vector (+ gate table) must fit one workgroup's LDS twice (adjoint sweep)."""
        words = self.na + self.nb + 2 * (1 << self.ncas) + self.Dc
        return ((2 * self.Dc + 2 * self.n_gates + 16 + self.n_theta) * 8 + 40 * self.n_gates
                + 4 * words) <= 160 * 1024

    def _tabs(self):
        i32 = torch.int32
        return (dptr(self.unrank_a, i32), dptr(self.unrank_b, i32), dptr(self.rank_a, i32),
                dptr(self.rank_b, i32), self.na, self.nb)

    def work(self, batch):
        if batch not in self._work:
            n = self.lib.oovqe_sector_work_size(self.ncas, self.na, self.nb, batch)
            self._work[batch] = torch.empty(n, dtype=F64, device=self.device)
        return self._work[batch]

    def state(self, theta, dense=False):
        """theta [batch, n_theta] -> psi_c [batch, Dc] (and dense [batch, 2^n] if requested)."""
        batch = theta.shape[0]
        psi_c = torch.empty((batch, self.Dc), dtype=F64, device=self.device)
        psi = (torch.empty((batch, 1 << (2 * self.ncas)), dtype=F64, device=self.device)
               if dense else None)
        pairs, max_pairs = self.pair_lists()
        check(self.lib.oovqe_sector_state_pl(dptr(theta), self.n_theta, dptr(self.gates_dev, torch.uint8),
                                             self.n_gates, self.ncas, ctypes.c_uint32(self.init_index),
                                             *self._tabs(), batch, dptr(pairs, torch.int32), max_pairs, dptr(psi_c),
                                             dptr(psi), stream_ptr()),
              "oovqe_sector_state_pl")
        return (psi_c, psi) if dense else psi_c

    def rdms(self, psi_c):
        batch = psi_c.shape[0]
        a = self.ncas
        gamma = torch.empty((batch, a, a), dtype=F64, device=self.device)
        Gamma = torch.empty((batch, a, a, a, a), dtype=F64, device=self.device)
        self.pair_lists()          # (the per-circuit block also holds the sector's excitation tables)
        check(self.lib.oovqe_sector_rdms_tb(dptr(psi_c), a, *self._tabs(), batch, self._tables_ptr, dptr(gamma),
                                            dptr(Gamma), dptr(self.work(batch)), stream_ptr()),
              "oovqe_sector_rdms_tb")
        return gamma, Gamma

    def adjoint(self, theta, psi_c, c1, c2):
        """d/dtheta (c1.gamma + c2.Gamma) for each batch element; call after rdms(psi_c)."""
        batch = theta.shape[0]
        dth = torch.empty((batch, self.n_theta), dtype=F64, device=self.device)
        pairs, max_pairs = self.pair_lists()
        check(self.lib.oovqe_sector_adjoint_pl(dptr(theta), self.n_theta,
                                               dptr(self.gates_dev, torch.uint8), self.n_gates,
                                               self.ncas, *self._tabs(), batch, dptr(psi_c),
                                               dptr(c1.contiguous()), dptr(c2.contiguous()),
                                               dptr(pairs, torch.int32), max_pairs, self._tables_ptr,
                                               dptr(self.work(batch)), dptr(dth), stream_ptr()),
              "oovqe_sector_adjoint_pl")
        return dth
