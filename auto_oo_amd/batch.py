"""Batched evaluation over molecular geometries (the Berry-phase-loop batch of the north star).

The reference evaluates one geometry at a time (examples/Tutorial_Berry_phase.ipynb builds a new
``OO_pqc`` per point).  At cc-pVDZ size one evaluation is only ~27 MB of HBM traffic, far too
little to fill an MI355X, so ``OO_pqc_batch`` stacks the per-geometry tensors of G ``OO_pqc``-like
problems (same basis size, same active space, same circuit) and evaluates all of them with ONE
call of ``oovqe_oo_eval_batch`` -- 5 kernel launches in which the geometry index is a grid
dimension.  Per geometry the arithmetic is exactly that of ``OO_pqc.energy_and_gradient``.

Circuits whose state lives in the (N_alpha, N_beta)-sector engine (more than 10 qubits: kUpCCD CAS(8e,8o), GateFabric
CAS(6e,6o), ...) take the same entry points (round 5): states, RDMs and derivative RDMs of ALL geometries from the
sector kernels with the geometry as their batch index, the CAS path of all geometries in one ``oovqe_cas_eval_batch``
call, the orbital Hessians in one ``oovqe_orbital_hessian_batch`` call, directions and line search in lockstep; only
the stages that take ONE set of CAS coefficients per call (the reverse sweep's operator, the theta-theta block) run
geometry by geometry.
"""
import ctypes
import time

import numpy as np
import torch

from . import _lib, excitations as X, ops
from ._lib import check, dptr, stream_ptr
from .oo_energy import mo_ao_to_mo_oao, non_redundant_indices

F64 = torch.float64


class OO_pqc_batch:
    def __init__(self, pqc, mols, ncas, nelecas, oao_mo_coeffs=None, freeze_active=False):
        """
        Args:
            pqc: Parameterized_circuit shared by all geometries
            mols: sequence of Moldata (same nao, same electron count)
            ncas, nelecas: active space
            oao_mo_coeffs: sequence of [N,N] OAO->MO coefficients (default: from mol.hf.mo_coeff)
            freeze_active: freeze active-active rotations (oo_energy.py:139-140)
        """
        self.lib = _lib.load()
        self.device = _lib.require_device()
        self.pqc = pqc
        self.G = len(mols)
        if self.G < 1:
            raise ValueError("need at least one geometry")
        self.nao = mols[0].nao
        self.ncas, self.nelecas = ncas, nelecas
        for m in mols:
            if m.nao != self.nao or m.nelectron != mols[0].nelectron:
                raise ValueError("all geometries of a batch must share nao and the electron count")
        self.occ_idx, self.act_idx, self.virt_idx = mols[0].get_active_space_idx(ncas, nelecas)
        self._n_occ = len(self.occ_idx)
        self.params_idx = non_redundant_indices(self.occ_idx, self.act_idx, self.virt_idx,
                                                freeze_active)
        self.n_kappa = len(self.params_idx)
        rows, cols = X.tril_tables(self.nao, self.params_idx)
        self._kap_row = torch.as_tensor(rows).to(self.device)
        self._kap_col = torch.as_tensor(cols).to(self.device)
        self.n_theta = int(np.prod(pqc.theta_shape))

        N = self.nao
        self.int2e_ao = torch.empty((self.G, N, N, N, N), dtype=F64, device=self.device)
        self.int1e_ao = torch.empty((self.G, N, N), dtype=F64, device=self.device)
        self.oao_coeff = torch.empty((self.G, N, N), dtype=F64, device=self.device)
        self.oao_mo_coeff = torch.empty((self.G, N, N), dtype=F64, device=self.device)
        self.mo_coeff = torch.empty((self.G, N, N), dtype=F64, device=self.device)
        self.nuc = torch.empty(self.G, dtype=F64, device=self.device)
        nuc_host = np.empty(self.G)
        for g, m in enumerate(mols):
            self.int2e_ao[g].copy_(ops.as_device(m.int2e_ao, self.device))
            self.int1e_ao[g].copy_(ops.as_device(m.int1e_ao, self.device))
            self.oao_coeff[g].copy_(ops.as_device(m.oao_coeff, self.device))
            if oao_mo_coeffs is None:
                m.run_rhf()
                c = mo_ao_to_mo_oao(m.hf.mo_coeff, m.overlap)
            else:
                c = oao_mo_coeffs[g]
            self.set_oao_mo_coeff(g, c)
            nuc_host[g] = m.nuc
        self.nuc.copy_(torch.as_tensor(nuc_host))
        # exact p<->q / r<->s symmetry of every geometry's integrals (true for PySCF's int2e), verified
        # bit for bit per geometry, and the packed resident copy the batched N^4 pass streams: ONE pass over the
        # stack (oovqe_eri_ingest).  The batch runs on the flags ALL its geometries share.  int2e_ao must
        # not be modified in place afterwards (set_molecule / reverify_integrals are the ways in).
        self._eri_packed = None
        self._ingest()
        self._plans = {}
        self._flat0 = None
        self._trial_orbitals = None
        self._step_args = None
        self._all_pd_last_step = False
        self.step_by_calls = False

    def set_oao_mo_coeff(self, g, oao_mo_coeff):
        """Replace the orbitals of geometry g and refresh mo_coeff[g] = S^-1/2 C_oao
        (oo_energy.py:173-176)."""
        c = ops.as_device(oao_mo_coeff, self.device)
        self.oao_mo_coeff[g].copy_(c)
        self.mo_coeff[g].copy_(ops.matmul_nn(self.oao_coeff[g].contiguous(), c))

    def reverify_integrals(self):
        """Call after writing into ``int2e_ao`` (or ``int1e_ao`` / ``oao_coeff`` / ``nuc``) in place,
        e.g. when the integrals of many geometries are produced on the device: re-checks the
        symmetry flags of the whole stack bit for bit, rebuilds the packed resident copy (or drops
        it) -- one pass over the stack -- and refreshes ``mo_coeff = S^-1/2 C_oao`` of every geometry."""
        self._ingest()
        self.refresh_mo_coeff()

    def set_molecule(self, g, mol, oao_mo_coeff=None):
        """Replace geometry g of the batch (the next point of a Berry-phase loop, say): integrals,
        OAO basis, nuclear repulsion and orbitals.  The symmetry flags of the batch are re-verified
        for the new integrals and its slice of the packed copy is rebuilt (one pass over the new tensor) -- never
        write into ``int2e_ao`` directly, the flags and the packed copy would go stale."""
        if mol.nao != self.nao:
            raise ValueError("all geometries of a batch must share nao")
        self.int2e_ao[g].copy_(ops.as_device(mol.int2e_ao, self.device))
        self.int1e_ao[g].copy_(ops.as_device(mol.int1e_ao, self.device))
        self.oao_coeff[g].copy_(ops.as_device(mol.oao_coeff, self.device))
        self.nuc[g] = float(mol.nuc)
        if oao_mo_coeff is None:
            mol.run_rhf()
            oao_mo_coeff = mo_ao_to_mo_oao(mol.hf.mo_coeff, mol.overlap)
        self.set_oao_mo_coeff(g, oao_mo_coeff)
        self._ingest(g)

    def _ingest(self, g=None):
        """Symmetry flags and packed copy of geometry ``g`` (None: of the whole stack) from one pass over the
        integrals (``ops.eri_ingest``).  eri_flags = the symmetries every geometry of the stack has (a geometry
        replaced by a symmetric one can RESTORE a flag, not only drop it); the packed copy (slabs p <= q, upper
        triangle of each slab: about a quarter of the tensor) exists while every geometry carries both."""
        both = ops.ERI_PQ_SYMMETRIC | ops.ERI_RS_SYMMETRIC
        psz = int(self.lib.oovqe_eri_packed_size(self.nao))
        want = psz > 0 and not (self.nao <= 48 and self._n_occ + self.ncas > 16)
        had = self._eri_packed is not None
        if want and not had:
            self._eri_packed = torch.empty((self.G, psz), dtype=F64, device=self.device)
        if g is None:
            self._flags_g, _ = ops.eri_ingest(self.int2e_ao, pack=want, out=self._eri_packed)
        else:
            (self._flags_g[g],), _ = ops.eri_ingest(self.int2e_ao[g], pack=want,
                                                    out=self._eri_packed[g] if want else None)
        flags = both
        for f in self._flags_g:
            flags &= f
        self.eri_flags = flags
        if flags != both:
            self._eri_packed = None
        elif want and not had and g is not None:
            # the new geometry restored the last missing flag: the other slices of the copy are made now
            check(self.lib.oovqe_eri_pack(dptr(self.int2e_ao), self.nao, self.G, dptr(self._eri_packed),
                                          stream_ptr()), "oovqe_eri_pack")

    def refresh_mo_coeff(self):
        """mo_coeff[g] = S^-1/2[g] C_oao[g] for the whole stack (oo_energy.py:173-176), one launch."""
        ops.matmul_nn_batch(self.oao_coeff, self.oao_mo_coeff, out=self.mo_coeff)

    def _plan(self, derivatives, slot=0):
        key = (bool(derivatives), slot)
        if key not in self._plans:
            pqc = self.pqc
            wsz = self.lib.oovqe_oo_eval_work_size(self.n_theta, pqc._n_gates, pqc.n_qubits, self.nao,
                                                   self._n_occ, self.ncas, int(key[0]))
            osz = self.lib.oovqe_oo_eval_out_size(self.n_theta, self.n_kappa, self.ncas, int(key[0]))
            work = torch.empty(self.G * wsz, dtype=F64, device=self.device)
            self._plans[key] = (work, int(osz))
        return self._plans[key]

    def evaluate(self, thetas, derivatives=True, count=None, slot=0, mo_coeff=None):
        """thetas [G, n_theta] (device, fp64) -> packed outputs [G, out_size]
        ([c0 | E | dE/dtheta | gvec rows | c1 | c2] per geometry, include/oovqe.h).
        ``count``: evaluate only the first ``count`` geometries of the batch.
        ``slot``: workspace slot -- calls issued on different HIP streams must use different slots
        (their kernels then overlap: the HBM-bound N^4 sweep of one call runs beside the
        latency-bound tail kernels of the other).
        ``mo_coeff``: [G, N, N] orbitals to evaluate at instead of ``self.mo_coeff`` (rotated trial
        orbitals of a line search)."""
        G = self.G if count is None else int(count)
        if not 1 <= G <= self.G:
            raise ValueError(f"count must be in 1..{self.G}")
        thetas = ops.as_device(thetas, self.device).reshape(-1, self.n_theta)[:G]
        if getattr(self.pqc, "_use_sector", False):
            return self._evaluate_sector(thetas, derivatives, G, slot, mo_coeff)
        work, osz = self._plan(derivatives, slot)
        out = torch.empty((G, osz), dtype=F64, device=self.device)
        pqc = self.pqc
        check(self.lib.oovqe_oo_eval_batch(
            dptr(thetas), self.n_theta, dptr(pqc._gates_dev, torch.uint8), pqc._n_gates,
            pqc.n_qubits, ctypes.c_uint32(pqc._init_index), dptr(self.int2e_ao),
            dptr(self.int1e_ao), dptr(self.mo_coeff if mo_coeff is None else mo_coeff), dptr(self.nuc),
            self.nao, self._n_occ,
            self.ncas, dptr(self._kap_row, torch.int32), dptr(self._kap_col, torch.int32),
            self.n_kappa, int(bool(derivatives)), G, dptr(work), dptr(out), int(self.eri_flags),
            dptr(self._eri_packed) if self.eri_flags == 3 else None, stream_ptr()),
            "oovqe_oo_eval_batch")
        return out

    # ---- circuits in the sector engine ---------------------------------------------------------------
    def _cas_batch(self, gamma, Gamma, G, slot=0, mo_coeff=None, want_fock=False):
        """The CAS path of the first G geometries from RDM sets gamma [G, nrdm, a, a], Gamma [G, nrdm, a,a,a,a]
        (``oovqe_cas_eval_batch``) -> (packed outputs [G, osz] in the layout of ``evaluate``, n_t, fock or None)."""
        nrdm = int(gamma.shape[1])
        key = ("cas", nrdm, slot)
        if key not in self._plans:
            wsz = self.lib.oovqe_cas_eval_work_size(self.nao, self._n_occ, self.ncas, nrdm)
            osz = self.lib.oovqe_oo_eval_out_size(nrdm - 1, self.n_kappa, self.ncas, int(nrdm > 1))
            self._plans[key] = (torch.empty(self.G * wsz, dtype=F64, device=self.device), int(osz))
        work, osz = self._plans[key]
        out = torch.empty((G, osz), dtype=F64, device=self.device)
        fock = torch.empty((G, self.nao, self.nao), dtype=F64, device=self.device) if want_fock else None
        gamma = gamma if gamma.is_contiguous() else gamma.contiguous()
        Gamma = Gamma if Gamma.is_contiguous() else Gamma.contiguous()
        check(self.lib.oovqe_cas_eval_batch(
            dptr(self.int2e_ao), dptr(self.int1e_ao), dptr(self.mo_coeff if mo_coeff is None else mo_coeff),
            dptr(gamma), dptr(Gamma), nrdm, dptr(self.nuc), self.nao, self._n_occ, self.ncas,
            dptr(self._kap_row, torch.int32), dptr(self._kap_col, torch.int32), self.n_kappa, G, dptr(work), dptr(out),
            dptr(fock), int(self.eri_flags), dptr(self._eri_packed) if self.eri_flags == 3 else None, stream_ptr()),
            "oovqe_cas_eval_batch")
        return out, (nrdm - 1 if nrdm > 1 else 1), fock

    def _evaluate_sector(self, thetas, derivatives, G, slot, mo_coeff):
        """``evaluate`` for a circuit in the sector engine: the states and RDMs of all geometries in one launch each
        (the geometry is the sector kernels' batch index), the CAS path of all geometries in one call, dE/dtheta by
        the reverse sweep with each geometry's (c1, c2) as cotangents (oo_pqc.py:86-95).  The packed layout is that
        of the dense path; the rows d gvec / d theta_k (the kappa-theta block, which the reverse sweep does not
        produce) are NaN here -- ``energy_gradient_hessian`` delivers them."""
        pqc, eng = self.pqc, self.pqc._sector
        nt, nk, a = self.n_theta, self.n_kappa, self.ncas
        psi = eng.state(thetas)
        g1, g2 = eng.rdms(psi)
        cas, _, _ = self._cas_batch(g1[:, None], g2[:, None], G, slot, mo_coeff)
        if not derivatives:
            return cas
        osz = int(self.lib.oovqe_oo_eval_out_size(nt, nk, a, 1))
        out = torch.full((G, osz), float("nan"), dtype=F64, device=self.device)
        out[:, 0:2] = cas[:, 0:2]                                    # c0, E
        out[:, 2 + nt:2 + nt + nk] = cas[:, 3:3 + nk]                # gvec row 0 = dE/dkappa
        c12 = cas[:, 3 + nk:]
        out[:, 2 + nt + (1 + nt) * nk:] = c12                        # c1 | c2
        if eng.geometry_coefficients_ok():
            # every geometry's reverse sweep in one launch sequence, its (c1 | c2) read where the CAS path left them
            out[:, 2:2 + nt] = eng.adjoint_geometries(thetas, psi, c12, cas.stride(0))
        else:
            c1 = c12[:, :a * a].reshape(G, a, a)
            c2 = c12[:, a * a:].reshape(G, a, a, a, a)
            for g in range(G):      # (the operator of the reverse sweep takes ONE set of coefficients per call)
                out[g, 2:2 + nt] = eng.adjoint(thetas[g:g + 1], psi[g:g + 1], c1[g], c2[g])[0]
        return out

    def _energy_gradient_hessian_sector(self, thetas):
        pqc, eng = self.pqc, self.pqc._sector
        G, nt, nk, a = self.G, self.n_theta, self.n_kappa, self.ncas
        n = nt + nk
        st = eng.tangent_states(thetas, pqc._gates, second=True)                 # [G, 1 + nt + n_pairs, Dc]
        psi, tau = st[:, 0:1], st[:, 1:1 + nt]
        # derivative RDMs by polarisation of the plain RDM kernel (exact: the RDMs are quadratic forms of the real
        # state), all geometries in one list of sector vectors
        vecs = torch.cat((psi, psi + tau, psi - tau), dim=1).reshape(G * (2 * nt + 1), eng.Dc)
        r1, r2 = eng.rdms_chunked(vecs)
        r1 = r1.reshape(G, 2 * nt + 1, a, a)
        r2 = r2.reshape(G, 2 * nt + 1, a, a, a, a)
        gamma = torch.cat((r1[:, 0:1], 0.5 * (r1[:, 1:1 + nt] - r1[:, 1 + nt:])), dim=1)
        Gamma = torch.cat((r2[:, 0:1], 0.5 * (r2[:, 1:1 + nt] - r2[:, 1 + nt:])), dim=1)
        out, _, fock = self._cas_batch(gamma, Gamma, G, want_fock=True)
        E, dE = out[:, 1], out[:, 2:2 + nt]
        gv = out[:, 2 + nt:2 + nt + (1 + nt) * nk].reshape(G, 1 + nt, nk)
        c12 = out[:, 2 + nt + (1 + nt) * nk:]
        H = torch.empty((G, n, n), dtype=F64, device=self.device)
        # kappa-kappa blocks of all geometries in one call
        key = ("orbital_hessian", 0)
        if key not in self._plans:
            wsz = self.lib.oovqe_orbital_hessian_work_size(self.nao, self._n_occ, self.ncas)
            self._plans[key] = (torch.empty(G * wsz, dtype=F64, device=self.device), 0)
        Hkk = torch.empty((G, nk, nk), dtype=F64, device=self.device)
        check(self.lib.oovqe_orbital_hessian_batch(
            dptr(self.int2e_ao), dptr(self.int1e_ao), dptr(self.mo_coeff), dptr(gamma[:, 0].contiguous()),
            dptr(Gamma[:, 0].contiguous()), dptr(fock), self.nao, self._n_occ, self.ncas,
            dptr(self._kap_row, torch.int32), dptr(self._kap_col, torch.int32), nk, G, dptr(self._plans[key][0]),
            dptr(Hkk), int(self.eri_flags), stream_ptr()), "oovqe_orbital_hessian_batch")
        H[:, nt:, nt:] = Hkk
        H[:, nt:, :nt] = gv[:, 1:].transpose(1, 2)
        H[:, :nt, nt:] = gv[:, 1:]
        if eng.geometry_coefficients_ok():
            # theta-theta blocks: H_jk = tau_jk . lam(psi) + tau_j . lam(tau_k), lam = the operator of EACH geometry
            # applied to its psi and first tangents -- one launch sequence for all geometries
            ja, ka, _ = eng.hessian_pair_tables(pqc._gates)
            lam = eng.lam_geometries(st[:, :1 + nt].reshape(G * (1 + nt), eng.Dc), 1 + nt, c12,
                                     out.stride(0)).reshape(G, 1 + nt, eng.Dc)
            first = (st[:, 1 + nt:] * lam[:, 0:1]).sum(dim=2)                           # [G, n_pairs]
            # tau_j . lam(tau_k): scalar products over the sector dimension, geometry by geometry (elementwise multiply
            # + sum, as SectorEngine.circuit_hessian_from_states: no vendor GEMM on this path)
            second = torch.stack([(st[g, 1:1 + nt, None, :] * lam[g, None, 1:, :]).sum(dim=2) for g in range(G)])
            val = first + second[:, ja, ka]
            Htt = torch.zeros((G, nt, nt), dtype=F64, device=self.device)
            Htt[:, ja, ka] = val
            Htt[:, ka, ja] = val
            H[:, :nt, :nt] = Htt
        else:
            c1 = c12[:, :a * a].reshape(G, a, a)
            c2 = c12[:, a * a:].reshape(G, a, a, a, a)
            for g in range(G):      # (one set of CAS coefficients per application of the operator)
                H[g, :nt, :nt] = eng.circuit_hessian_from_states(st[g], pqc._gates, c1[g].contiguous(),
                                                                 c2[g].contiguous())
        grad = torch.cat((dE, gv[:, 0]), dim=1)
        return E, grad, H

    def evaluate_deferred(self, thetas, derivatives=True, count=None, mo_coeff=None, view=None):
        """``evaluate`` enqueued on one of the library's two side streams (taken in turn, each with a workspace of its
        own) -> ``ops.PendingTensor``; ``.result()`` joins it to the stream that is current then.  For INDEPENDENT calls
        issued back to back (a scan over parameter sets, the points of several loops, a benchmark's steps): the
        latency-bound tail of one call (q -> x / p -> n, Fock panels, assembly: a quarter of a 256-geometry call,
        during which HBM idles) runs under the N^4 sweep of the next call instead of in front of it.  The inputs
        must be complete on the current stream when this is called (the side stream is forked from it here);
        per call the launches and the arithmetic are those of ``evaluate``: the same bits."""
        if getattr(self.pqc, "_use_sector", False):
            # the sector engine keeps ONE workspace per batch size (not per stream): its calls stay in order on the
            # current stream; the result is complete in stream order, the handle has nothing to wait for
            out = self.evaluate(thetas, derivatives=derivatives, count=count, mo_coeff=mo_coeff)
            return ops.PendingTensor(out, None, view)
        side_streams = ops.side_streams(self.device)
        k = self._defer_next = (getattr(self, "_defer_next", 1) + 1) & 1
        side = side_streams[k]
        side.wait_stream(torch.cuda.current_stream())
        thetas = ops.as_device(thetas, self.device)
        with torch.cuda.stream(side):
            out = self.evaluate(thetas, derivatives=derivatives, count=count, slot=1 + k, mo_coeff=mo_coeff)
            event = torch.cuda.Event()
            event.record(side)
        thetas.record_stream(side)
        if mo_coeff is not None:
            mo_coeff.record_stream(side)
        return ops.PendingTensor(out, event, view)

    def energy_and_gradient(self, thetas, count=None, slot=0, defer=False):
        """-> [G, 1 + n_theta + n_kappa]: column 0 = E, then dE/dtheta, then dE/dkappa.
        ``defer``: -> ``ops.PendingTensor`` of the same (``evaluate_deferred``)."""
        n_out = 2 + self.n_theta + self.n_kappa
        if defer:
            return self.evaluate_deferred(thetas, derivatives=True, count=count, view=lambda o: o[:, 1:n_out])
        out = self.evaluate(thetas, derivatives=True, count=count, slot=slot)
        return out[:, 1:n_out]

    def energy(self, thetas, kappas=None):
        """-> [G] energies (OO_pqc.energy_from_parameters(theta, kappa) per geometry, oo_pqc.py:64-84).
        ``kappas`` [G, n_kappa]: evaluate at the rotated orbitals mo_coeff[g] expm(-K(kappa[g])) --
        one extra launch for all geometries (oovqe_rotate_orbitals_batch)."""
        if kappas is None:
            return self.evaluate(thetas, derivatives=False)[:, 1]
        return self.evaluate(thetas, derivatives=False, mo_coeff=self.rotated_mo_coeff(kappas))[:, 1]

    # ---- orbital rotations of the whole stack -------------------------------------------------------
    def _rotate(self, C, kappas, out):
        kappas = ops.as_device(kappas, self.device).reshape(self.G, self.n_kappa)
        N = self.nao
        work = None
        if N > 48:
            work = torch.empty((self.G + 7) * N * N, dtype=F64, device=self.device)
        check(self.lib.oovqe_rotate_orbitals_batch(
            dptr(kappas), dptr(self._kap_row, torch.int32), dptr(self._kap_col, torch.int32),
            self.n_kappa, N, self.G, dptr(C), dptr(out), None, dptr(work), stream_ptr()),
            "oovqe_rotate_orbitals_batch")
        return out

    def rotated_mo_coeff(self, kappas):
        """[G, N, N]: mo_coeff[g] @ expm(-K(kappas[g])) (OO_energy.get_transformed_mo per geometry)."""
        return self._rotate(self.mo_coeff, kappas, torch.empty_like(self.mo_coeff))

    def rotate_(self, kappas):
        """oao_mo_coeff[g] <- oao_mo_coeff[g] @ expm(-K(kappas[g])) for every geometry (the update of
        oo_pqc.py:191) and mo_coeff = S^-1/2 C_oao refreshed."""
        if self.nao <= 48:      # (one workgroup per geometry, all in LDS: the result may overwrite its input)
            self._rotate(self.oao_mo_coeff, kappas, self.oao_mo_coeff)
        else:
            new = self._rotate(self.oao_mo_coeff, kappas, torch.empty_like(self.oao_mo_coeff))
            self.oao_mo_coeff.copy_(new)
        self.refresh_mo_coeff()

    # ---- configs[3]'s unit of work, batched -----------------------------------------------------------
    def energy_gradient_hessian(self, thetas):
        """E [G], full gradient [G, n] and full Hessian [G, n, n] (n = n_theta + n_kappa) of every
        geometry from ONE library call (``oovqe_oo_hessian_batch``): OO_pqc.full_gradient +
        OO_pqc.full_hessian (oo_pqc.py:132-148) with the geometry index as a grid dimension of every
        launch.  Block layout of the Hessian as the reference's: [[tt, (kt)^T], [kt, kk]]."""
        G, nt, nk = self.G, self.n_theta, self.n_kappa
        n = nt + nk
        thetas = ops.as_device(thetas, self.device).reshape(G, nt)
        pqc = self.pqc
        if getattr(pqc, "_use_sector", False):
            return self._energy_gradient_hessian_sector(thetas)
        pairs_dev, _, _ = ops._hessian_pair_tables(nt, self.device)
        n_pairs = pairs_dev.shape[0]
        key = ("hessian", 0)
        if key not in self._plans:
            wsz = self.lib.oovqe_oo_hessian_work_size(nt, pqc._n_gates, pqc.n_qubits, self.nao, self._n_occ,
                                                      self.ncas, n_pairs)
            osz = self.lib.oovqe_oo_eval_out_size(nt, nk, self.ncas, 1)
            self._plans[key] = (torch.empty(G * wsz, dtype=F64, device=self.device), int(osz))
        work, osz = self._plans[key]
        out = torch.empty((G, osz), dtype=F64, device=self.device)
        H = torch.empty((G, n, n), dtype=F64, device=self.device)
        check(self.lib.oovqe_oo_hessian_batch(
            dptr(thetas), nt, dptr(pqc._gates_dev, torch.uint8), pqc._n_gates, pqc.n_qubits,
            ctypes.c_uint32(pqc._init_index), dptr(self.int2e_ao), dptr(self.int1e_ao), dptr(self.mo_coeff),
            dptr(self.nuc), self.nao, self._n_occ, self.ncas, dptr(self._kap_row, torch.int32),
            dptr(self._kap_col, torch.int32), nk, dptr(pairs_dev, torch.int32), n_pairs, G, dptr(work),
            dptr(out), dptr(H), int(self.eri_flags),
            dptr(self._eri_packed) if self.eri_flags == 3 else None, stream_ptr()), "oovqe_oo_hessian_batch")
        return out[:, 1], out[:, 2:2 + n], H

    def full_gradient(self, thetas):
        """-> [G, n_theta + n_kappa] (OO_pqc.full_gradient per geometry, oo_pqc.py:132-134)."""
        return self.energy_and_gradient(thetas)[:, 1:]

    def full_hessian(self, thetas):
        """-> [G, n, n] (OO_pqc.full_hessian per geometry, oo_pqc.py:136-148)."""
        return self.energy_gradient_hessian(thetas)[2]

    def damped_newton_step(self, thetas, opt=None, defer_lowest=False):
        """One damped Newton step on (theta, kappa) of EVERY geometry in lockstep -- the body of
        OO_pqc.full_optimization / of the Berry-phase loop (oo_pqc.py:172-196), per geometry the
        arithmetic of NewtonStep.damped_newton_step: gradient + Hessian, the G directions, a line search
        whose trials evaluate all geometries at once, then the orbitals of every geometry rotated.
        Returns (new thetas [G, n_theta], energies at the new parameters [G], lowest Hessian eigenvalues [G]).
        ``defer_lowest``: the eigenvalues come as an ``ops.PendingLowest`` -- they are a diagnostic
        (hess_eig_l of the reference's loops) computed on a side stream beside the line search; ``.result()``
        joins them.

        The whole step up to the first verdict of the line search is ONE library call
        (``oovqe_oo_newton_step_batch``: ~45 launches back to back) and one 32-byte readback; only further trials
        (rare) are driven from here.  ``step_by_calls=True`` (attribute; for tests and measurements) drives every
        stage through its own entry point instead -- the same launches, the same bits."""
        from .newton_raphson import BatchedNewtonStep
        if opt is None:
            opt = BatchedNewtonStep(verbose=0)
        nt = self.n_theta
        thetas = ops.as_device(thetas, self.device).reshape(self.G, nt)
        thetas = thetas if thetas.is_contiguous() else thetas.contiguous()
        if self._trial_orbitals is None:
            self._trial_orbitals = (torch.empty_like(self.oao_mo_coeff), torch.empty_like(self.mo_coeff))

        def trial(pa, pb):
            # the orbitals of a trial are formed as an accepted step forms them -- C_oao expm(-K), then
            # S^-1/2 (C_oao U) (oo_pqc.py:191, oo_energy.py:173-176) -- so the accepted trial's orbitals ARE the new
            # ones: the step adopts them by exchanging buffers, nothing is launched behind the last readback
            t_oao, t_mo = self._trial_orbitals
            self._rotate(self.oao_mo_coeff, pb, t_oao)
            ops.matmul_nn_batch(self.oao_coeff, t_oao, out=t_mo)
            return self.evaluate(pa, derivatives=False, mo_coeff=t_mo)[:, 1]

        n = nt + self.n_kappa
        if (self.step_by_calls or n > self.lib.oovqe_newton_direction_max_n() or self.G > 32767
                or getattr(self.pqc, "_use_sector", False)):
            E, grad, H = self.energy_gradient_hessian(thetas)
            flat = self._flat0
            if flat is None:
                flat = self._flat0 = torch.zeros((self.G, n), dtype=F64, device=self.device)
            flat[:, :nt] = thetas                    # (the kappa part stays zero: steps start at the current orbitals)
            # the energy at the accepted trial point IS the energy at the new parameters, so the loop body's closing
            # evaluation (oo_pqc.py:195) is not repeated
            (new_thetas, new_kappas), low, energies = opt.damped_newton_steps_flat(
                trial, flat, grad, H, energy0=E, defer_lowest=True, split=nt, return_energy=True)
        else:
            new_thetas, new_kappas, low, energies = self._newton_step_one_call(thetas, opt, trial)
        if opt.last_search_gave_up:
            self.rotate_(new_kappas)                 # (some problems went back to their old parameters)
        else:
            # every trial evaluates ALL geometries at their current points, the accepted ones at their accepted
            # point: the last trial's orbitals are the new orbitals of the whole stack.  Like the reference's
            # `self.oao_mo_coeff = ...` the attributes are rebound, not written in place.
            t_oao, t_mo = self._trial_orbitals
            self._trial_orbitals = (self.oao_mo_coeff, self.mo_coeff)
            self.oao_mo_coeff, self.mo_coeff = t_oao, t_mo
        return new_thetas, energies, (low if defer_lowest else low.checked())

    def _step_block(self):
        """The argument block of oovqe_oo_newton_step_batch with everything that does not change from step to
        step filled in (made once per batch object)."""
        if self._step_args is not None:
            return self._step_args
        lib, pqc, G, N = self.lib, self.pqc, self.G, self.nao
        nt, nk = self.n_theta, self.n_kappa
        n = nt + nk
        pairs_dev, _, _ = ops._hessian_pair_tables(nt, self.device)
        n_pairs = int(pairs_dev.shape[0])
        osz1 = int(lib.oovqe_oo_eval_out_size(nt, nk, self.ncas, 1))
        osz0 = int(lib.oovqe_oo_eval_out_size(nt, nk, self.ncas, 0))
        key = ("hessian", 0)
        if key not in self._plans:
            wsz = lib.oovqe_oo_hessian_work_size(nt, pqc._n_gates, pqc.n_qubits, N, self._n_occ, self.ncas, n_pairs)
            self._plans[key] = (torch.empty(G * wsz, dtype=F64, device=self.device), osz1)
        work_eval, _ = self._plan(False, 0)
        b = _lib.NewtonStepT()
        keep = {"pairs": pairs_dev, "trial_out": torch.empty((G, osz0), dtype=F64, device=self.device),
                "flat": torch.empty((G, n), dtype=F64, device=self.device)}
        # the first trial's verdict goes straight into pinned host memory (the library's last kernel of the step writes
        # its four flags through the device-visible pointer; the host polls them instead of a 32-byte memcpy + stream
        # synchronisation: ~12 us of every step); later trials (rare) keep a device tensor and a readback
        keep["flags_host"] = torch.full((4,), float("nan"), dtype=F64).pin_memory()
        keep["flags_np"] = keep["flags_host"].numpy()
        keep["flags_dev"] = torch.empty(4, dtype=F64, device=self.device)
        has_pd = bool(lib.oovqe_newton_direction_has_pd(n, 1))
        if has_pd:
            keep["work_pd"] = torch.empty(int(lib.oovqe_newton_direction_pd_work_size(n, G)), dtype=F64,
                                          device=self.device)
        rest = int(lib.oovqe_newton_direction_rest_work_size(n, G))
        keep["work_rest"] = torch.empty(rest, dtype=F64, device=self.device)
        # one workspace per side stream: the eigenvalue routes of consecutive steps run beside each other
        keep["work_side"] = [torch.empty(rest, dtype=F64, device=self.device) for _ in range(2)]
        if N > 48:
            keep["work_rotate"] = torch.empty((G + 7) * N * N, dtype=F64, device=self.device)
        b.gates = pqc._gates_dev.data_ptr()
        b.kap_row, b.kap_col = self._kap_row.data_ptr(), self._kap_col.data_ptr()
        b.pairs = pairs_dev.data_ptr()
        b.work_hessian = self._plans[key][0].data_ptr()
        b.work_eval = work_eval.data_ptr()
        b.work_pd = keep["work_pd"].data_ptr() if has_pd else None
        b.work_rest = keep["work_rest"].data_ptr()
        b.work_rotate = keep["work_rotate"].data_ptr() if N > 48 else None
        b.flat = keep["flat"].data_ptr()
        b.trial_out = keep["trial_out"].data_ptr()
        b.n_theta, b.n_gates, b.n_qubits, b.N, b.n_occ, b.ncas = nt, pqc._n_gates, pqc.n_qubits, N, self._n_occ, self.ncas
        b.n_kappa, b.n_pairs, b.batch = nk, n_pairs, G
        b.init_index = pqc._init_index
        # the eigenvalue route beside the line search keeps to a quarter of the chip (half for larger stacks):
        # ops.newton_direction
        b.side_wg = max(1, 64 // G) if G <= 16 else max(1, 128 // G)
        # per-step slab (doubles): out | H | grad | energy | dp | low | nu | info | t | state | flags | pa | pb
        sizes = [("out", G * osz1), ("hessian", G * n * n), ("grad", G * n), ("energy", G), ("dp", G * n),
                 ("lowest", G), ("shift", G), ("info", G), ("t", G), ("state", 3 * G), ("flags", 4),
                 ("points_a", G * nt), ("points_b", G * nk)]
        offs, o = {}, 0
        for k, sz in sizes:
            offs[k] = (o, sz)
            o += (sz + 1) & ~1                      # (16-byte aligned pieces)
        self._step_args = (b, keep, offs, o, osz0)
        return self._step_args

    def _newton_step_one_call(self, thetas, opt, trial):
        from .newton_raphson import LockstepSearch
        b, keep, offs, total, osz0 = self._step_block()
        G, nt, nk = self.G, self.n_theta, self.n_kappa
        n = nt + nk
        slab = torch.empty(total, dtype=F64, device=self.device)
        base = slab.data_ptr()
        for k, (o, _) in offs.items():
            setattr(b, k, base + 8 * o)
        flags_np = keep["flags_np"]
        flags_np[:] = np.nan
        b.flags = keep["flags_host"].data_ptr()

        def view(k, *shape):
            o, sz = offs[k]
            return slab[o:o + sz].view(*shape)

        t_oao, t_mo = self._trial_orbitals
        b.theta = thetas.data_ptr()
        b.g_ao, b.h_ao, b.nuc = self.int2e_ao.data_ptr(), self.int1e_ao.data_ptr(), self.nuc.data_ptr()
        b.g_packed = (self._eri_packed.data_ptr() if (self.eri_flags == 3 and self._eri_packed is not None)
                      else None)        # (no packed copy exists for M > 16 at N <= 48: _refresh_flags)
        b.eri_flags = int(self.eri_flags)
        b.oao_coeff, b.oao_mo_coeff, b.mo_coeff = (self.oao_coeff.data_ptr(), self.oao_mo_coeff.data_ptr(),
                                                   self.mo_coeff.data_ptr())
        b.trial_oao, b.trial_mo = t_oao.data_ptr(), t_mo.data_ptr()
        b.lambda_min, b.mu, b.rho, b.alpha, b.beta = opt.lambda_min, opt.mu, opt.rho, opt.alpha, opt.beta
        b.aug = int(bool(opt.aug))
        # when the previous step of this stack found every Hessian positive definite, the band route of "the
        # others" is not waited for (it runs beside the trial; flags[1] says whether that was right)
        spec = bool(self._all_pd_last_step)
        b.speculate = int(spec)
        side = ops._side_stream(self.device)
        b.work_rest_side = keep["work_side"][ops.side_streams(self.device).index(side)].data_ptr()
        refused = False
        try:
            check(self.lib.oovqe_oo_newton_step_batch(ctypes.byref(b), stream_ptr(), ctypes.c_void_p(side.cuda_stream)),
                  "oovqe_oo_newton_step_batch")
        except _lib.OovqeError:
            # beyond N = 48 the trial's orbital rotation fails loudly on a direction the library refused (dp = NaN):
            # the search below repeats that direction without inter-workgroup waits / through eigh, or raises
            torch.cuda.current_stream().wait_stream(side)
            if min(slab[offs["info"][0]:offs["info"][0] + G].tolist()) >= 0:
                raise
            refused = True
        event = torch.cuda.Event()
        event.record(side)
        # the side stream's route reads and writes this slab: it stays referenced here until that route's event has
        # completed (a host-side query per step) -- record_stream() on a megabyte block costs the allocator ~35 us at the
        # NEXT torch.empty (event bookkeeping on the critical path of the next step)
        held = self.__dict__.setdefault("_slabs_in_flight", [])
        held[:] = [(sl, ev) for sl, ev in held if not ev.query()]
        held.append((slab, event))
        # (the views are made while the device works through the step, the readback comes last)
        state = view("state", 3, G)
        lam_args = (float(opt.lambda_min), float(opt.mu), float(opt.rho), int(bool(opt.aug)))

        def retry_lowest():
            # eigenvalues the side route could not deliver: again on the calling stream, one workgroup per problem
            check(self.lib.oovqe_newton_direction_rest(
                dptr(view("hessian", G, n, n)), dptr(view("grad", G, n)), n, G, *lam_args, dptr(view("info", G)), 2, 1,
                dptr(keep["work_rest"]), dptr(view("dp", G, n)), dptr(view("lowest", G)), dptr(view("shift", G)),
                stream_ptr()), "oovqe_newton_direction_rest")

        s = LockstepSearch(flat=keep["flat"], g=view("grad", G, n), H=view("hessian", G, n, n), dp=view("dp", G, n),
                           low=ops.PendingLowest(view("lowest", G), event, retry_lowest), nu=view("shift", G),
                           info=view("info", G), energy=view("energy", G), t=view("t", G), active=state[0],
                           best=state[1], slope=state[2], flags=keep["flags_dev"], pa=view("points_a", G, nt),
                           pb=view("points_b", G, nk))
        if refused:
            fl = [1.0, -1.0, 0.0, 0.0]
        else:
            # the verdict of the first trial: polled in pinned memory (falls back to a stream synchronisation should
            # the flags not arrive within seconds -- a failed launch surfaces there)
            t_poll = time.perf_counter()
            while np.isnan(flags_np).any():
                if time.perf_counter() - t_poll > 5.0:
                    torch.cuda.current_stream().synchronize()
                    if np.isnan(flags_np).any():
                        raise _lib.OovqeError("oovqe_oo_newton_step_batch: the step's verdict never arrived")
            fl = flags_np.tolist()
        self._all_pd_last_step = fl[1] >= 1.0
        if refused:
            s.t.fill_(1.0)
            fl = None
        elif spec and fl[1] < 1.0:
            # some Hessian was not positive definite after all: its direction is still on the side stream.  Join it
            # and search from the start.
            torch.cuda.current_stream().wait_event(event)
            s.t.fill_(1.0)
            fl = None
        opt.run_search(s, trial, split=nt, first_flags=fl)
        return s.pa, s.pb, s.low, s.best

