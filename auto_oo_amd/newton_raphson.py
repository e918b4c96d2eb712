"""Damped Newton step with augmented Hessian and backtracking line search, on the device.

Drop-in for the reference's ``NewtonStep`` (src/auto_oo/utils/newton_raphson.py:47-211: same
constructor, same method names and return values, same acceptance rule), built differently:

* the direction comes from ``oovqe_newton_direction`` (reduction of the Hessian to a band of half-width
  8 on the matrix cores, several workgroups per problem; lowest eigenvalue by multisection on a band
  LDL^T, one shift per lane; the reference's level shift ``mu + rho |lambda_low|``; band solve) instead
  of two ``eigh`` calls and an explicit inverse (newton_raphson.py:78-129);
* the line search keeps energies on the device and reads back once per trial (one small tensor:
  old energy, trial energy, Armijo slope, lowest eigenvalue) instead of once per comparison
  (newton_raphson.py:131-192).

``BatchedNewtonStep`` (an extension) takes the same step for G independent problems in lockstep:
one launch with G workgroups for the directions, one readback per line-search trial for all of them.
"""
import ctypes

import torch

from . import _lib, ops


def wolfe(t, grad, dp, alpha=1e-4):
    """Armijo term alpha * t * <grad, dp> (newton_raphson.py:12-13)."""
    return alpha * t * torch.dot(grad, dp)


def split_list_shapes(parameters, paramshapes):
    """Cut a flat parameter vector back into tensors of the given shapes (newton_raphson.py:214-224)."""
    out, start = [], 0
    for shape in paramshapes:
        count = 1
        for extent in shape:
            count *= int(extent)
        out.append(parameters[start:start + count].reshape(shape))
        start += count
    return out


def _flatten(parameters):
    shapes = [tuple(p.shape) for p in parameters]
    return torch.cat([p.reshape(-1) for p in parameters]), shapes


class NewtonStep():
    def __init__(self, alpha=0.0001, beta=.5, mu=1e-6, rho=1.1, lmax=20, lambda_min=1e-6,
                 aug=True, verbose=1):
        """Hyper-parameters of newton_raphson.py:47-77: Armijo constant ``alpha``, step reduction
        ``beta``, level shift ``mu + rho |lambda_low|`` applied when the lowest Hessian eigenvalue
        is below ``lambda_min`` (``aug``), at most ``lmax`` reductions of the step."""
        self.alpha = alpha
        self.beta = beta
        self.mu = mu
        self.rho = rho
        self.lmax = lmax
        self.lambda_min = lambda_min
        self.aug = aug
        self.verbose = verbose
        self.last_search_gave_up = False       # (set by damped_newton_steps_flat: the search ended in a give-up)

    # ---- direction ----------------------------------------------------------------------------
    def _eigh_direction(self, H, g):
        """The reference's own algebra on the device (newton_raphson.py:105-128) -- only beyond the library's
        kernels: n > 5128 (no configuration of the reference's workloads gets there: N = 200, CAS(6e,6o) has
        n_kappa = 4 659), or an indefinite Hessian that is to be inverted as it stands (aug = False) at
        n > 480."""
        vals, vecs = torch.linalg.eigh(H)
        low = vals[..., 0]
        nu = torch.zeros_like(low)
        if self.aug:
            nu = torch.where(low < self.lambda_min, self.mu + self.rho * low.abs(), nu)
        proj = torch.einsum("...ji,...j->...i", vecs, g) / (vals + nu[..., None])
        return -torch.einsum("...ij,...j->...i", vecs, proj), low, nu

    def _direction(self, gradient, hessian, batched=False, defer_lowest=False):
        """(dp, lowest eigenvalue, shift, info) as device tensors; no host synchronisation.  info [G]: the
        library's per-problem verdict (ops.NEWTON_INFO; negative = it failed loudly and dp is NaN), checked on
        the host together with the line search's first readback (``_check_direction``).  ``defer_lowest``: the
        eigenvalue as an ``ops.PendingLowest`` (still being computed beside whatever follows)."""
        dev = _lib.require_device()
        g = ops.as_device(gradient, dev)
        H = ops.as_device(hessian, dev)
        if not batched:
            g = g.reshape(-1)
        if g.shape[-1] <= _lib.load().oovqe_newton_direction_max_n():
            return ops.newton_direction(H, g, self.lambda_min, self.mu, self.rho, self.aug, want_info=True,
                                        defer_lowest=defer_lowest)
        dp, low, nu = self._eigh_direction(H, g)
        if defer_lowest:
            low = ops.PendingLowest(low, None)
        return dp, low, nu, torch.zeros(dp.shape[0] if batched else 1, dtype=dp.dtype, device=dp.device)

    def _check_direction(self, gradient, hessian, dp, low, nu, info_host, batched=False):
        """Act on negative entries of the library's info (host values): a timed-out hand-off between the
        workgroups of a problem is repeated with one workgroup per problem (no inter-workgroup wait left to
        time out); an indefinite Hessian without level shift beyond the pivoted kernel goes through eigh.
        Anything else negative, or a second failure, raises -- a NaN direction is never handed on."""
        codes = [int(c) for c in info_host]
        if min(codes) >= 0:
            return dp, low, nu
        dev = _lib.require_device()
        g = ops.as_device(gradient, dev)
        H = ops.as_device(hessian, dev)
        if not batched:
            g = g.reshape(-1)
        if -2 in codes and min(codes) >= -2:
            return self._eigh_direction(H, g)
        if all(c >= -1 for c in codes):
            dp, low, nu, info = ops.newton_direction(H, g, self.lambda_min, self.mu, self.rho, self.aug,
                                                     want_info=True, max_wg=1)
            again = [int(c) for c in info.tolist()]
            if min(again) >= 0:
                return dp, low, nu
            codes = again
        bad = sorted(set(c for c in codes if c < 0))
        raise _lib.OovqeError("oovqe_newton_direction failed: " +
                              "; ".join(f"{c}: {ops.NEWTON_INFO.get(c, 'unknown')}" for c in bad))

    def newton_step(self, gradient, hessian):
        """newton_raphson.py:78-129 -> (dp, lowest Hessian eigenvalue as a float)."""
        dp, low, nu, info = self._direction(gradient, hessian)
        lowest_eigenvalue, shift, code = torch.stack((low.reshape(()), nu.reshape(()), info.reshape(()))).tolist()
        if code < 0:
            dp, low, nu = self._check_direction(gradient, hessian, dp, low, nu, [code])
            lowest_eigenvalue, shift = torch.stack((low.reshape(()), nu.reshape(()))).tolist()
        if self.verbose:
            print("lowest eigval hessian =", lowest_eigenvalue)
            if shift != 0.0:
                print("augmenting hessian... lowest eigenvalue of augmented hessian:",
                      lowest_eigenvalue + shift)
        return dp, lowest_eigenvalue

    # ---- line search --------------------------------------------------------------------------
    def _search(self, objective_fn, parameters, dp, gradient, extra=None):
        """Backtracking with the acceptance rule of newton_raphson.py:146-183.  Returns
        (new parameters, new energy, host values of `extra`)."""
        flat, shapes = _flatten(parameters)
        dp = dp.to(flat.device)
        slope = self.alpha * torch.dot(ops.as_device(gradient, flat.device).reshape(-1), dp)
        at = lambda step: objective_fn(*split_list_shapes(flat + step * dp, shapes)).reshape(())  # noqa: E731
        e_old = objective_fn(*parameters).reshape(())
        e_try = at(1.0)
        pack = [e_old.to(flat.device), e_try.to(flat.device), slope]
        if extra is not None:
            pack += [x.reshape(()) for x in extra]
        host = torch.stack(pack).tolist()                      # the one readback of the common case
        old, trial, slope_h = host[0], host[1], host[2]
        if slope_h != slope_h:
            raise _lib.OovqeError("Newton direction is not finite (NaN in <gradient, dp>)")
        step, reductions = 1.0, 0
        # (a NaN trial energy is never accepted: it counts as a failed Armijo test)
        if not trial <= old + step * slope_h:
            if not slope_h < 0:
                raise AssertionError("Newton direction is not a descent direction")
            if self.verbose:
                print("test_energy:", trial, "... old energy:", old)
                print("do backtracking line search...")
            while not trial <= old + step * slope_h:
                step *= self.beta
                reductions += 1
                if self.verbose:
                    print("t =", step)
                trial = at(step).item()
                if reductions > self.lmax:
                    # newton_raphson.py:177-183: after lmax + 1 reductions the search gives up and
                    # returns the old parameters (whatever the last trial said)
                    step, trial = 0.0, old
                    if self.verbose:
                        print("Warning: line search failed. Output previous parameters.")
                    break
        if self.verbose:
            print("new energy:", trial)
            print("old energy:", old)
        new_flat = flat + step * dp
        new = tuple(split_list_shapes(new_flat, shapes)) if len(parameters) > 1 else new_flat
        return new, trial, host[3:]

    def backtracking(self, objective_fn, parameters, dp, gradient):
        """newton_raphson.py:131-192 -> (new parameters, new energy)."""
        new, energy, _ = self._search(objective_fn, parameters, dp, gradient)
        return new, energy

    def damped_newton_step(self, objective_fn, parameters, gradient, hessian, defer_lowest=False):
        """newton_raphson.py:194-211 -> (new parameters, lowest Hessian eigenvalue).  The lowest eigenvalue of
        a positive definite Hessian is computed BESIDE the line search (side stream) and read back after it:
        the direction does not depend on it (newton_raphson.py:105-128).  ``defer_lowest`` (an extension; ignored
        when verbose): the eigenvalue comes back as an ``ops.PendingLowest`` (``.item()`` / ``float()`` join it) --
        a loop of steps that only COLLECTS the eigenvalues (``hess_eig_l`` of ``OO_pqc.full_optimization``) then
        never waits for the band route (1.1 ms at n = 331 where the step itself takes 0.4 ms)."""
        dp, low, nu, info = self._direction(gradient, hessian, defer_lowest=True)
        try:
            new, _, (shift, code) = self._search(objective_fn, parameters, dp, gradient, extra=(nu, info))
        except _lib.OovqeError:
            code = info.reshape(()).item()
            if code >= 0:
                raise
        if code < 0:            # the library refused loudly: repeat / fall back, then search again
            dp, low_t, nu = self._check_direction(gradient, hessian, dp, None, nu, [code])
            low = ops.PendingLowest(low_t, None)
            new, _, (shift,) = self._search(objective_fn, parameters, dp, gradient, extra=(nu,))
        if defer_lowest and not self.verbose:
            return new, low
        lowest_eigenvalue = low.result().reshape(()).item()
        if self.verbose:
            print("lowest eigval hessian =", lowest_eigenvalue)
            if shift != 0.0:
                print("hessian was augmented by", shift)
        return new, lowest_eigenvalue


class BatchedNewtonStep(NewtonStep):
    """Extension (not in the reference): the same damped Newton step for G INDEPENDENT problems at
    once -- e.g. the geometries of a rank's shard, or independent Berry-phase loops in lockstep.
    The G directions are one launch of G workgroups, and the line search synchronises with the
    host once per trial instead of once per problem.  Per problem the arithmetic is that of
    ``NewtonStep`` (newton_raphson.py:78-211)."""

    def newton_steps(self, gradients, hessians):
        """gradients [G, n], hessians [G, n, n] -> (dp [G, n], lowest eigenvalues [G])"""
        dp, low, nu, info = self._direction(gradients, hessians, batched=True)
        codes = info.tolist()
        if min(codes) < 0:
            dp, low, nu = self._check_direction(gradients, hessians, dp, low, nu, codes, batched=True)
        return dp, low

    def damped_newton_steps(self, objective_fns, parameters, gradients, hessians):
        """objective_fns[g](*parameters[g]) -> 0-d tensor; parameters[g] = tuple of tensors.
        Returns (list of new parameter tuples, lowest Hessian eigenvalues [G])."""
        G = len(objective_fns)
        shapes = [[tuple(p.shape) for p in ps] for ps in parameters]
        flat = torch.stack([torch.cat([p.flatten() for p in ps]) for ps in parameters])

        def evaluate(points):
            return torch.stack([objective_fns[g](*split_list_shapes(points[g], shapes[g])).reshape(())
                                for g in range(G)])

        newp, low = self.damped_newton_steps_flat(evaluate, flat, gradients, hessians)
        return [tuple(split_list_shapes(newp[g], shapes[g])) for g in range(G)], low

    def damped_newton_steps_flat(self, objective, flat, gradients, hessians, energy0=None, defer_lowest=False,
                                 split=None, return_energy=False):
        """The same with ONE objective for all problems: objective(points [G, n]) -> energies [G]
        (e.g. ``OO_pqc_batch.energy``: every line-search trial is one batched evaluation).
        flat [G, n] = the current parameters; energy0 [G] = objective(flat) when the caller has it.
        ``split`` = n_a: the objective takes the first n_a parameters and the rest as two contiguous tensors,
        objective(points_a [G, n_a], points_b [G, n - n_a]), and so are the new parameters returned.
        Returns (new parameters [G, n] -- or (new_a, new_b) -- , lowest Hessian eigenvalues [G]
        [, energies at the new parameters [G]]); with ``defer_lowest`` the eigenvalues come as an
        ``ops.PendingLowest`` (they are computed beside the line search on a side stream and nothing in the
        step reads them).

        Per problem the acceptance rule is the reference's (newton_raphson.py:146-183); the book-keeping of a
        trial is two small launches (``oovqe_linesearch_points`` / ``_update``) and one 32-byte readback."""
        lib = _lib.load()
        dev = _lib.require_device()
        g = ops.as_device(gradients, dev)
        H = ops.as_device(hessians, dev)
        flat = ops.as_device(flat, dev)
        g = g if g.is_contiguous() else g.contiguous()
        flat = flat if flat.is_contiguous() else flat.contiguous()
        G, n = flat.shape
        n_a = n if split is None else int(split)
        if g.shape[-1] <= lib.oovqe_newton_direction_max_n():
            dp, low, nu, info = ops.newton_direction(H, g, self.lambda_min, self.mu, self.rho, self.aug,
                                                     defer_lowest=True, want_info=True)
        else:
            dp, low, nu = self._eigh_direction(H, g)
            low, info = ops.PendingLowest(low, None), None
        if energy0 is None:
            energy0 = (objective(flat) if split is None
                       else objective(flat[:, :n_a].contiguous(), flat[:, n_a:].contiguous()))
        kw = dict(dtype=flat.dtype, device=dev)
        state = torch.empty((3, G), **kw)            # active | best energy | slope
        s = LockstepSearch(flat=flat, g=g, H=H, dp=dp, low=low, nu=nu, info=info,
                           energy=energy0 if energy0.is_contiguous() else energy0.contiguous(),
                           t=torch.ones(G, **kw), active=state[0], best=state[1], slope=state[2],
                           flags=torch.empty(4, **kw), pa=torch.empty((G, n_a), **kw),
                           pb=torch.empty((G, n - n_a), **kw) if n_a < n else None)
        self.run_search(s, objective, split)
        new = s.pa if split is None else (s.pa, s.pb)
        out = (new, s.low if defer_lowest else s.low.result())
        return out + (s.best,) if return_energy else out

    def run_search(self, s, objective, split=None, first_flags=None):
        """The backtracking line search of G problems in lockstep on the state ``s`` (a ``LockstepSearch``: the
        directions are in it).  ``first_flags``: the host values of ``s.flags`` after a first trial (t = 1 for every
        problem) that has been made already -- by ``oovqe_oo_newton_step_batch``, which enqueues a whole step up to
        that verdict; None: the first trial is made here.  On return s.pa / s.pb are the new parameters, s.best the
        energies there."""
        lib = _lib.load()
        G, n = s.flat.shape
        n_a = s.pa.shape[1]
        sp = ops.stream_ptr

        def points(with_slope):
            ops.check(lib.oovqe_linesearch_points(ops.dptr(s.flat), ops.dptr(s.dp), ops.dptr(s.t), ops.dptr(s.g),
                                                  float(self.alpha), n, n_a, G, ops.dptr(s.pa),
                                                  ops.dptr(s.pb) if s.pb is not None else None,
                                                  ops.dptr(s.slope) if with_slope else None, sp()),
                      "oovqe_linesearch_points")

        def update(trial, first, give_up, with_info):
            if trial is not None and (trial.dtype != s.flat.dtype or trial.device != s.flat.device or trial.dim() != 1):
                raise _lib.OovqeError("the objective must return a 1-d fp64 device tensor of energies")
            ops.check(lib.oovqe_linesearch_update(
                # (a strided view is fine: e.g. column 1 of the packed outputs of a batched evaluation)
                ctypes.c_void_p(trial.data_ptr()) if trial is not None else None,
                trial.stride(0) if trial is not None else 1,
                ops.dptr(s.energy), ops.dptr(s.slope),
                ops.dptr(s.info) if (with_info and s.info is not None) else None,
                float(self.beta), int(first), int(give_up), G, ops.dptr(s.t), ops.dptr(s.active), ops.dptr(s.best),
                ops.dptr(s.flags), sp()), "oovqe_linesearch_update")
            return s.flags.tolist()                   # the one readback of a trial

        def trial_energies():
            e = objective(s.pa) if split is None else objective(s.pa, s.pb)
            return e.reshape(G)

        self.last_search_gave_up = False
        fl = first_flags
        if fl is None:
            points(True)
            try:
                fl = update(trial_energies(), 1, 0, True)
            except _lib.OovqeError:
                # a direction the library refused (info < 0: dp = NaN) makes the trial's orbital rotation itself fail
                # loudly beyond N = 48 (expm checks its input): that is the refusal, not a new error
                if s.info is None or min(s.info.tolist()) >= 0:
                    raise
                fl = [1.0, -1.0, 0.0, 0.0]
        if fl[1] < 0:
            # the library refused a problem loudly: repeat without inter-workgroup waits / eigh fallback (or
            # raise), then search from scratch
            dp, low_t, s.nu = self._check_direction(s.g, s.H, s.dp, None, s.nu, s.info.tolist(), batched=True)
            s.dp = dp if dp.is_contiguous() else dp.contiguous()
            s.low = ops.PendingLowest(low_t, None)
            s.t.fill_(1.0)
            points(True)
            fl = update(trial_energies(), 1, 0, False)
        if fl[2] != 0.0:
            raise _lib.OovqeError("Newton direction is not finite (NaN in <gradient, dp>)")
        num = 0
        while fl[0] != 0.0:
            if fl[3] != 0.0:
                raise AssertionError("Newton direction is not a descent direction")
            num += 1
            if num > self.lmax:
                # newton_raphson.py:177-183: give up on the problems still searching
                update(None, 0, 1, False)
                points(False)
                self.last_search_gave_up = True       # (the returned points are no trial's: some are the old ones)
                if self.verbose:
                    print("Warning: line search failed. Output previous parameters.")
                break
            points(False)
            fl = update(trial_energies(), 0, 0, False)
        return fl


class LockstepSearch:
    """State of the line search of G problems in lockstep (device tensors): flat [G, n] current parameters,
    g [G, n], H [G, n, n], dp [G, n], low (``ops.PendingLowest``), nu [G], info [G] or None, energy [G] at flat,
    t [G] step lengths, active / best / slope [G], flags [4], pa [G, n_a] / pb [G, n - n_a] trial points."""
    __slots__ = ("flat", "g", "H", "dp", "low", "nu", "info", "energy", "t", "active", "best", "slope", "flags",
                 "pa", "pb")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw[k])
