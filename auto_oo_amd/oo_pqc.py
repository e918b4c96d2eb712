"""OO_pqc: hybrid orbital-optimized VQE cost function (energy, composite gradients, Hessians).

Drop-in for the reference's ``auto_oo.OO_pqc`` (src/auto_oo/oo_pqc.py:30-207).  Where the
reference obtains theta-derivatives by back-propagating torch autograd through ~500 simulator ops
(oo_pqc.py:86-95,103-125), this engine propagates tangent states through the Givens passes and
feeds the derivative RDMs into the same Fock kernel -- the numbers are the same derivatives.
"""
import numpy as np
import torch

from . import ops
from .autodiff import differentiable_scalar, kernel_scope, needs_autodiff, unwrap
from .newton_raphson import NewtonStep
from .oo_energy import OO_energy, _OrbitalRotationRule, _is_zero, _matvec

F64 = torch.float64


class _ParametersEnergyModel(_OrbitalRotationRule):
    """E(theta, kappa) = OO_pqc.energy_from_parameters as torch sees it (oo_pqc.py:64-95): value,
    gradient (dE/dtheta from the tangent-state kernels; dE/dkappa through C0 expm(-K), see
    _OrbitalRotationRule) and, at kappa = 0, the second derivatives = the three analytic Hessian
    blocks (oo_pqc.py:103-148), which is where test/test_oo_pqc.py:113-125 takes them."""

    def value(self, theta, kappa):
        C, _, _ = self.rotated(kappa)
        return self.oo._evaluate(theta, C, derivatives=False)["E"].reshape(())

    def grad(self, theta, kappa):
        oo = self.oo
        C, U, K = self.rotated(kappa)
        if U is None:                       # kappa = 0: one fused pass gives both blocks
            res = oo._evaluate(theta, C)
            return res["dE"].reshape(theta.shape), res["gvec"][0].reshape(kappa.shape)
        if getattr(oo.pqc, "_use_sector", False):
            dE = oo._evaluate_adjoint(theta, C)["dE"]
            g1, g2 = oo.pqc.get_rdms(theta)
            res = oo._cas_eval(C, g1[None].contiguous(), g2[None].contiguous(), want_matrices=True)
        else:
            res = oo._evaluate(theta, C, derivatives=True, want_matrices=True)
            dE = res["dE"]
        gk = self.kappa_gradient(res["fock"], res["gvec"][0], U, K)
        return dE.reshape(theta.shape), gk.reshape(kappa.shape)

    def hvp(self, xs, vs, needs=None):
        theta, kappa = (unwrap(x) for x in xs)
        if not _is_zero(kappa):
            return self._hvp_rotated(theta, kappa, vs)
        H = getattr(self, "_H", None)
        if H is None:
            with kernel_scope():
                H = self._H = self.oo.full_hessian(theta)
        nt = theta.numel()
        vt, vk = vs
        out_t = out_k = None
        if vt is not None:
            out_t, out_k = _matvec(H[:nt, :nt], vt), _matvec(H[nt:, :nt], vt)
        if vk is not None:
            bt, bk = _matvec(H[:nt, nt:], vk), _matvec(H[nt:, nt:], vk)
            out_t = bt if out_t is None else out_t + bt
            out_k = bk if out_k is None else out_k + bk
        return (None if out_t is None else out_t.reshape(theta.shape),
                None if out_k is None else out_k.reshape(kappa.shape))


    def _hvp_rotated(self, theta, kappa, vs):
        """Second derivatives at kappa != 0 (oo_pqc.py:103-125 differentiates at any point): the three
        analytic blocks at the rotated orbitals over ALL rotation pairs, pulled back through the map
        between kappa and the local rotation parameters (_OrbitalRotationRule)."""
        oo = self.oo
        pqc = oo.pqc
        cache = getattr(self, "_rot", None)
        if cache is None:
            with kernel_scope():
                C, U, K = self.rotated(kappa)
                C = oo._t(C)
                tr, tc = self._full_pairs()
                gamma, Gamma = pqc.rdms_with_derivatives(theta)
                res = ops.cas_eval(oo.int2e_ao, oo.int1e_ao, C, gamma, Gamma, oo.nuc, oo._n_occ, oo.ncas, tr, tc,
                                   want_matrices=True, eri_flags=oo._eri_flags())
                Hxt = res["gvec"][1:].T.contiguous()                       # [P, n_theta]
                nt = oo._n_theta()
                if getattr(pqc, "_use_sector", False):     # large registers: inside the (N_alpha, N_beta) sector
                    Htt = pqc._sector.circuit_hessian(pqc._theta2d(theta), pqc._gates, res["c1"], res["c2"]).reshape(nt, nt)
                else:
                    Htt = ops.circuit_hessian(pqc._theta2d(theta).reshape(-1), pqc._gates_dev, pqc._n_gates,
                                              pqc.n_qubits, oo.ncas, pqc._init_index, res["c1"], res["c2"]).reshape(nt, nt)
                Hxx = ops.orbital_hessian(oo.int2e_ao, oo.int1e_ao, C, gamma[0].contiguous(), Gamma[0].contiguous(),
                                          res["fock"], oo._n_occ, oo.ncas, tr, tc, want_matrix=True)[0]
                cache = self._rot = (U, K, res["gmat"], Hxx, Hxt, Htt)
        U, K, Gm, Hxx, Hxt, Htt = cache
        vt, vk = vs
        for v in (vt, vk):
            if (v is not None and torch._C._functorch.is_functorch_wrapped_tensor(v)
                    and torch._C._functorch.is_batchedtensor(v)):
                raise NotImplementedError("batched tangents at kappa != 0: use "
                                          "torch.autograd.functional.hessian (vectorize=False)")
        with kernel_scope():
            out_t = out_x = curv = None
            if vt is not None:
                t = unwrap(vt).reshape(-1)
                out_t, out_x = _matvec(Htt, t), _matvec(Hxt, t)
            if vk is not None:
                v = unwrap(vk).reshape(-1)
                xv, B = self.local_push(U, K, v)
                bt, bx = _matvec(Hxt.T, xv), _matvec(Hxx, xv)
                out_t = bt if out_t is None else out_t + bt
                out_x = bx if out_x is None else out_x + bx
                curv = self.local_curvature(U, K, Gm, v, B)
            out_k = None
            if out_x is not None:
                out_k = self.local_pull(U, K, out_x)
                if curv is not None:
                    out_k = out_k + curv
        return (None if out_t is None else out_t.reshape(theta.shape),
                None if out_k is None else out_k.reshape(kappa.shape))


class OO_pqc(OO_energy):
    def __init__(self, pqc, mol, ncas, nelecas, oao_mo_coeff=None, freeze_active=False,
                 interface='torch'):
        """oo_pqc.py:36-62"""
        super().__init__(mol, ncas, nelecas, oao_mo_coeff=oao_mo_coeff,
                         freeze_active=freeze_active, interface=interface)
        self.pqc = pqc

    def _n_theta(self):
        return int(np.prod(self.pqc.theta_shape))

    # ---- fused evaluation (one pass: state + tangents, RDMs + derivative RDMs, one N^4 sweep) -----
    def _evaluate_adjoint(self, theta, mo_coeff):
        """Large circuits (sector engine): RDMs -> CAS path (E, orbital gradient, c1, c2) ->
        reverse-mode theta-gradient with c1, c2 as cotangents (3 + 5 + 5 launches)."""
        pqc = self.pqc
        eng = pqc._sector
        th = pqc._theta2d(theta)
        psi_c = eng.state(th)
        gamma, Gamma = eng.rdms(psi_c)
        res = self._cas_eval(mo_coeff, gamma, Gamma)
        res["dE"] = eng.adjoint(th, psi_c, res["c1"], res["c2"])[0]
        res["packed"] = torch.cat((res["E"], res["dE"], res["gvec"][0]))
        return res

    def _evaluate(self, theta, mo_coeff=None, derivatives=True, want_matrices=False):
        if mo_coeff is None:
            mo_coeff = self.mo_coeff
        if derivatives == "adjoint" or (derivatives is True and getattr(self.pqc, "_use_sector", False)
                                        and not want_matrices):
            return self._evaluate_adjoint(theta, mo_coeff)
        if not want_matrices and not getattr(self.pqc, "_use_sector", False):
            # one C call: circuit (+tangents) -> RDM sets -> CAS path, persistent workspace
            plans = self.__dict__.setdefault("_plans2", {})
            # the plan bakes in the integrals' symmetry flags: it is valid for one STATE of the
            # tensors (object + in-place version counter); an edited int2e_ao gets a fresh plan
            state = (id(self.pqc), id(self.int2e_ao), self.int2e_ao._version, id(self.int1e_ao),
                     self.int1e_ao._version)
            hit = plans.get(bool(derivatives))
            if hit is None or hit[0] != state:
                pqc = self.pqc
                plan = ops.OoEvalPlan(pqc._gates_dev, pqc._n_gates, self._n_theta(), pqc.n_qubits,
                                      pqc._init_index, self.int2e_ao, self.int1e_ao, self.nuc,
                                      self._n_occ, self.ncas, self._kap_row, self._kap_col,
                                      derivatives=derivatives, eri_flags=self._eri_flags(),
                                      g_packed=self._eri_packed())
                plans[bool(derivatives)] = hit = (state, plan)
            plan = hit[1]
            th = self.pqc._theta2d(theta).reshape(-1)
            return plan.unpack(plan(th, self._t(mo_coeff)))
        if derivatives:
            gamma, Gamma = self.pqc.rdms_with_derivatives(theta)
        else:
            g1, g2 = self.pqc.get_rdms(theta)
            gamma, Gamma = g1[None], g2[None]
        return self._cas_eval(mo_coeff, gamma, Gamma, want_matrices=want_matrices)

    def energy_and_gradient(self, theta):
        """Extension: E and the full gradient (circuit part, then orbital part) from ONE fused pass
        (the reference needs three simulations and three N^5 transforms for the same numbers,
        oo_pqc.py:64-101,132-134)."""
        res = self._evaluate(theta)
        packed = res["packed"]            # [E | dE/dtheta | dE/dkappa], written by one kernel
        return packed[0], packed[1:]

    # ---- reference API ------------------------------------------------------------------------------
    def energy_from_parameters(self, theta, kappa=None):
        """oo_pqc.py:64-84.  Differentiable by torch (autograd.functional.jacobian / hessian,
        torch.func.jacrev / hessian): first order in (theta, kappa) anywhere, second order at
        kappa = 0 -- the derivative rules call the analytic kernels (_ParametersEnergyModel)."""
        if needs_autodiff(theta, kappa):
            theta_d, = self._diff_args(theta)
            if kappa is None:
                kappa_d = torch.zeros(self.n_kappa, dtype=F64, device=self.device)
            else:
                kappa_d, = self._diff_args(kappa)
            return differentiable_scalar(_ParametersEnergyModel(self), theta_d, kappa_d)
        if kappa is None:
            mo_coeff = self.mo_coeff
        else:
            mo_coeff = self.get_transformed_mo(self.mo_coeff, kappa)
        return self._evaluate(theta, mo_coeff, derivatives=False)["E"].reshape(())

    def circuit_gradient(self, theta):
        """oo_pqc.py:86-95: dE/dtheta"""
        return self._evaluate(theta)["dE"].reshape(self._n_theta())

    def orbital_gradient(self, theta):
        """oo_pqc.py:97-101: analytic orbital gradient, flattened to the non-redundant kappa."""
        return self._evaluate(theta, derivatives=False)["gvec"][0]

    def orbital_circuit_hessian(self, theta):
        """oo_pqc.py:113-125: d(orbital gradient)/d theta, shape [n_kappa, n_theta] (forward-mode
        tangent RDMs through the linear Fock map)."""
        gamma, Gamma = self.pqc.rdms_with_derivatives(theta)
        res = self._cas_eval(self.mo_coeff, gamma, Gamma)
        return res["gvec"][1:].T.contiguous()

    def circuit_circuit_hessian(self, theta):
        """oo_pqc.py:103-111: d^2E/dtheta^2 (second tangent states + transition RDMs contracted
        with the CAS coefficients, ``oovqe_circuit_second_tangents`` / ``_hessian_assemble``)."""
        pqc = self.pqc
        c0, c1, c2 = self.get_active_integrals(self.mo_coeff)
        n = self._n_theta()
        if getattr(pqc, "_use_sector", False):
            # large registers (kUpCCD CAS(8e,8o)): tangents and quadratic forms inside the (N_alpha, N_beta) sector
            return pqc._sector.circuit_hessian(pqc._theta2d(theta), pqc._gates, self._t(c1), self._t(c2)).reshape(n, n)
        th = pqc._theta2d(theta).reshape(-1)
        return ops.circuit_hessian(th, pqc._gates_dev, pqc._n_gates, pqc.n_qubits, self.ncas,
                                   pqc._init_index, c1, c2).reshape(n, n)

    def orbital_orbital_hessian(self, theta):
        """oo_pqc.py:127-130: orbital-orbital Hessian on the non-redundant kappa pairs."""
        one_rdm, two_rdm = self.pqc.get_rdms(theta)
        return self.analytic_hessian_matrix(one_rdm, two_rdm)

    def full_hessian(self, theta):
        """oo_pqc.py:136-148: [[theta-theta, (kappa-theta)^T], [kappa-theta, kappa-kappa]].  The
        three blocks share ONE pass over the integrals: the RDMs and their theta-derivatives go
        through the CAS path together (set 0 yields c1, c2 and the Fock matrix, sets k >= 1 the
        kappa-theta columns), then the circuit block and the orbital block follow from those."""
        pqc = self.pqc
        C = self._t(self.mo_coeff)
        if (not getattr(pqc, "_use_sector", False) and not getattr(self, "hessian_by_blocks", False)
                and self.n_kappa >= 1 and self.nao <= 48):
            # ONE library call: the batched entry point on a stack of one geometry (this object's own tensors, no
            # copy) -- the same launches as the three calls below without the host work between them and without the
            # concatenations: 184 -> 140 us at the cc-pVDZ shape, the same bits (`hessian_by_blocks = True` on the
            # object keeps the three calls; beyond N = 48 the two forms pick different kernels and agree to rounding
            # only: the three calls stay)
            return self._full_hessian_one_call(theta, C)
        gamma, Gamma = pqc.rdms_with_derivatives(theta)
        res = self._cas_eval(C, gamma, Gamma, want_matrices=True)
        hessian_vqe_oo = res["gvec"][1:].T
        n = self._n_theta()
        if getattr(pqc, "_use_sector", False):
            hessian_vqe_vqe = pqc._sector.circuit_hessian(pqc._theta2d(theta), pqc._gates, res["c1"],
                                                          res["c2"]).reshape(n, n)
        else:
            hessian_vqe_vqe = ops.circuit_hessian(pqc._theta2d(theta).reshape(-1), pqc._gates_dev, pqc._n_gates,
                                                  pqc.n_qubits, self.ncas, pqc._init_index, res["c1"],
                                                  res["c2"]).reshape(n, n)
        hessian_oo_oo = ops.orbital_hessian(self.int2e_ao, self.int1e_ao, C, gamma[0].contiguous(),
                                            Gamma[0].contiguous(), res["fock"], self._n_occ, self.ncas,
                                            self._kap_row, self._kap_col, want_matrix=True,
                                            want_full=False)[0]
        return torch.cat((torch.cat((hessian_vqe_vqe, hessian_vqe_oo.T), dim=1),
                          torch.cat((hessian_vqe_oo, hessian_oo_oo), dim=1)), dim=0)

    def _full_hessian_one_call(self, theta, C):
        import ctypes
        from ._lib import check, dptr, stream_ptr
        lib = ops._lib.load()
        pqc = self.pqc
        nt, nk, N = self._n_theta(), self.n_kappa, self.nao
        n = nt + nk
        th = pqc._theta2d(theta).reshape(1, nt).contiguous()
        pairs_dev, _, _ = ops._hessian_pair_tables(nt, self.device)
        n_pairs = int(pairs_dev.shape[0])
        # scratch per (object, STREAM): calls on different HIP streams must not share it
        key = torch.cuda.current_stream().cuda_stream
        plan = self.__dict__.setdefault("_hess1_plans", {}).get(key)
        if plan is None:
            if len(self._hess1_plans) >= 4:
                self._hess1_plans.clear()
            wsz = lib.oovqe_oo_hessian_work_size(nt, pqc._n_gates, pqc.n_qubits, N, self._n_occ, self.ncas, n_pairs)
            osz = lib.oovqe_oo_eval_out_size(nt, nk, self.ncas, 1)
            plan = self._hess1_plans[key] = (torch.empty(int(wsz), dtype=F64, device=self.device), int(osz),
                                             torch.tensor([float(self.nuc)], dtype=F64, device=self.device))
        work, osz, nuc = plan
        out = torch.empty((1, osz), dtype=F64, device=self.device)
        H = torch.empty((1, n, n), dtype=F64, device=self.device)
        flags = int(self._eri_flags())
        packed = self._eri_packed()
        check(lib.oovqe_oo_hessian_batch(
            dptr(th), nt, dptr(pqc._gates_dev, torch.uint8), pqc._n_gates, pqc.n_qubits,
            ctypes.c_uint32(pqc._init_index), dptr(self.int2e_ao), dptr(self.int1e_ao), dptr(C), dptr(nuc), N,
            self._n_occ, self.ncas, dptr(self._kap_row, torch.int32), dptr(self._kap_col, torch.int32), nk,
            dptr(pairs_dev, torch.int32), n_pairs, 1, dptr(work), dptr(out), dptr(H), flags,
            dptr(packed) if packed is not None else None, stream_ptr()), "oovqe_oo_hessian_batch")
        return H[0]

    def full_gradient(self, theta):
        """oo_pqc.py:132-134"""
        return self.energy_and_gradient(theta)[1]

    def full_circuit_hessian_to_matrix(self, full_circuit_hessian):
        """oo_pqc.py:150-153"""
        size = self._n_theta()
        return full_circuit_hessian.reshape(size, size)

    def _one_geometry_stack(self):
        """This geometry as an ``OO_pqc_batch`` of one (its own resident copy of the integrals, kept per state of
        ``int2e_ao`` / ``int1e_ao``), for the one-call Newton iteration of ``full_optimization`` -- or None where that
        path does not apply (sector circuits, N > 64 [the copy], more parameters than the direction kernels
        take)."""
        from .batch import OO_pqc_batch
        pqc = self.pqc
        n = int(np.prod(pqc.theta_shape)) + self.n_kappa
        if (getattr(pqc, "_use_sector", False) or self.nao > 64 or self.n_kappa < 1
                or n > ops._lib.load().oovqe_newton_direction_max_n()):
            return None
        g, h = self.int2e_ao, self.int1e_ao
        # the stack holds its own copy of the integrals: it belongs to ONE state of these tensors.  The key keeps the
        # tensors themselves (compared with `is`: an id() can be reused by a new tensor once the old one is freed)
        key = (g, g._version, h, h._version, self.oao_coeff)
        hit = self.__dict__.get("_stack1")
        if hit is None or not (hit[0][0] is g and hit[0][1] == g._version and hit[0][2] is h
                               and hit[0][3] == h._version and hit[0][4] is self.oao_coeff):
            oo = self

            class _Mol:                      # what OO_pqc_batch reads of a Moldata
                nao, nuc = oo.nao, oo.nuc
                nelectron = 2 * oo._n_occ + oo.nelecas
                int1e_ao, int2e_ao, oao_coeff = h, g, oo.oao_coeff

                @staticmethod
                def get_active_space_idx(ncas, nelecas):
                    return oo.occ_idx, oo.act_idx, oo.virt_idx
            stack = OO_pqc_batch(pqc, [_Mol], self.ncas, self.nelecas, oao_mo_coeffs=[self._t(self.oao_mo_coeff)],
                                 freeze_active=self._freeze_active)
            hit = self.__dict__["_stack1"] = (key, stack)
        return hit[1]

    def full_optimization(self, theta_init, max_iterations=50, conv_tol=1e-10, verbose=0,
                          flush=True, **kwargs):
        """oo_pqc.py:155-207: Newton-Raphson on circuit and orbital parameters together.  The
        reference's quirks are kept: ``kappa_l`` collects theta (oo_pqc.py:189) and convergence is
        only tested for n > 1 (oo_pqc.py:200)."""
        opt = NewtonStep(verbose=verbose, **kwargs)
        theta_init = self._t(theta_init)
        energy_init = self.energy_from_parameters(theta_init).item()
        if verbose is not None:
            print(f"iter = 000, energy = {energy_init:.12f}", flush=flush)
        theta_l, kappa_l, oao_mo_coeff_l, energy_l, hess_eig_l = [], [], [], [], []
        theta = theta_init
        # (`optimization_by_calls = True` on the object keeps the loop below: tests and measurements)
        stack = None if (verbose or getattr(self, "optimization_by_calls", False)) else self._one_geometry_stack()
        if stack is not None:
            # One library call per iteration (oovqe_oo_newton_step_batch on a stack of one geometry: gradient,
            # Hessian, direction, the line search's first trial and its verdict back to back) instead of a
            # dozen calls with host work between them; the accepted trial's energy IS the closing energy of
            # oo_pqc.py:195.  Same arithmetic per geometry (tests: lockstep == sequential).
            from .newton_raphson import BatchedNewtonStep
            bopt = BatchedNewtonStep(verbose=0, **kwargs)
            stack.oao_mo_coeff[0].copy_(self._t(self.oao_mo_coeff))
            stack.refresh_mo_coeff()
            th = theta.reshape(1, -1)
            for n in range(max_iterations):
                th, e_new, low = stack.damped_newton_step(th, bopt, defer_lowest=True)
                hess_eig_l.append(low)
                theta = th[0].reshape(self.pqc.theta_shape)
                theta_l.append(theta)
                kappa_l.append(theta)
                self.oao_mo_coeff = stack.oao_mo_coeff[0].clone()
                oao_mo_coeff_l.append(self.oao_mo_coeff)
                energy = e_new[0].item()
                energy_l.append(energy)
                if verbose is not None:
                    print(f"iter = {n+1:03}, energy = {energy:.12f}")
                if n > 1:
                    if abs(energy_l[-1] - energy_l[-2]) < conv_tol:
                        if verbose is not None:
                            print("optimization finished.")
                            print("E_fin =", energy_l[-1])
                        break
            max_iterations = 0
        for n in range(max_iterations):
            kappa = torch.zeros(self.n_kappa, dtype=F64, device=self.device)
            grad = self.full_gradient(theta)
            hess = self.full_hessian(theta)
            # (the lowest eigenvalue is only collected: it is joined when the loop is over, so an iteration
            # does not wait for the band route that computes it beside the line search)
            new_theta_kappa, hess_eig = opt.damped_newton_step(
                self.energy_from_parameters, (theta, kappa), grad, hess, defer_lowest=True)
            hess_eig_l.append(hess_eig)
            theta = new_theta_kappa[0].reshape(self.pqc.theta_shape)
            kappa = new_theta_kappa[1]
            theta_l.append(theta)
            kappa_l.append(theta)
            self.oao_mo_coeff = ops.matmul_nn(self._t(self.oao_mo_coeff),
                                              self.kappa_to_mo_coeff(kappa))
            oao_mo_coeff_l.append(self.oao_mo_coeff)
            energy = self.energy_from_parameters(theta).item()
            energy_l.append(energy)
            if verbose is not None:
                print(f"iter = {n+1:03}, energy = {energy:.12f}")
            if n > 1:
                if abs(energy_l[-1] - energy_l[-2]) < conv_tol:
                    if verbose is not None:
                        print("optimization finished.")
                        print("E_fin =", energy_l[-1])
                    break
        hess_eig_l = [e.item() if isinstance(e, ops.PendingLowest) else e for e in hess_eig_l]
        return energy_l, theta_l, kappa_l, oao_mo_coeff_l, hess_eig_l
