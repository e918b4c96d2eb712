"""Damped Newton step with augmented Hessian and backtracking line search.

Mirror of the reference's ``NewtonStep`` (src/auto_oo/utils/newton_raphson.py:12-224): same
hyper-parameters, same control flow, same printed messages.  It is a host-side driver over the
cost function; the (n_theta + n_kappa)-sized ``eigh`` runs through torch on the device (SURVEY.md
section 2 row 8: out of scope as a kernel).
"""
import torch


def wolfe(t, grad, dp, alpha=1e-4):
    """newton_raphson.py:12-13"""
    return alpha * t * torch.dot(grad, dp)


def split_list_shapes(parameters, paramshapes):
    """newton_raphson.py:214-224"""
    chunks = []
    num = 0
    for shape in paramshapes:
        shapesize = 1
        for s in shape:
            shapesize *= int(s)
        chunks.append(parameters[num:num + shapesize].reshape(shape))
        num += shapesize
    return chunks


class NewtonStep():
    def __init__(self, alpha=0.0001, beta=.5, mu=1e-6, rho=1.1, lmax=20, lambda_min=1e-6,
                 aug=True, verbose=1):
        """newton_raphson.py:47-77"""
        self.alpha = alpha
        self.beta = beta
        self.mu = mu
        self.rho = rho
        self.lmax = lmax
        self.lambda_min = lambda_min
        self.aug = aug
        self.verbose = verbose

    def newton_step(self, gradient, hessian):
        """newton_raphson.py:78-129"""
        vhessian, whessian = torch.linalg.eigh(hessian)
        lowest_eigenvalue = vhessian[0].item()
        if self.verbose:
            print("lowest eigval hessian =", lowest_eigenvalue)
        if lowest_eigenvalue < self.lambda_min and self.aug:
            if self.verbose:
                print("augmenting hessian...")
            # The reference diagonalises H + c*1 again (newton_raphson.py:116-120); its eigenvectors
            # are those of H and its eigenvalues are shifted by c, so the second eigh (7 ms for the
            # 331 x 331 Hessian of configs[3], half of the whole step) is replaced by the shift.
            vhessian = vhessian + (self.mu + self.rho * abs(lowest_eigenvalue))
            if self.verbose:
                print("Lowest eigenvalue of augmented hessian:", vhessian[0].item())
        hessian_inv = whessian @ torch.diag(1 / vhessian) @ whessian.T
        dp = - (hessian_inv @ gradient)
        return dp, lowest_eigenvalue

    def backtracking(self, objective_fn, parameters, dp, gradient):
        """newton_raphson.py:131-192"""
        nargs = len(parameters)
        t = 1.
        energy = objective_fn(*parameters).item()
        parameters_tot = torch.cat([parameter.flatten() for parameter in parameters])
        paramshapes = [tuple(parameter.shape) for parameter in parameters]
        newp = parameters_tot + (t * dp)
        test_energy = objective_fn(*split_list_shapes(newp, paramshapes))
        if test_energy > energy + wolfe(t, gradient, dp, alpha=self.alpha):
            assert (wolfe(t, gradient, dp, alpha=self.alpha) < 0)
            num = 0
            if self.verbose:
                print("test_energy:", test_energy.item(), "... old energy:", energy)
                print("do backtracking line search...")
            while test_energy > energy + wolfe(t, gradient, dp, alpha=self.alpha):
                t = self.beta * t
                if self.verbose:
                    print("t =", t)
                newp = parameters_tot + (t * dp)
                test_energy = objective_fn(*split_list_shapes(newp, paramshapes))
                num += 1
                if num > self.lmax:
                    t = 0.
                    test_energy = objective_fn(*parameters)
                    if self.verbose:
                        print("Warning: line search failed. Output previous parameters.")
                    break
        new_energy = test_energy.item()
        newp = parameters_tot + (t * dp)
        if self.verbose:
            print("new energy:", new_energy)
            print("old energy:", energy)
        if nargs > 1:
            new_parameters = tuple(split_list_shapes(newp, paramshapes))
        else:
            new_parameters = newp
        return new_parameters, new_energy

    def damped_newton_step(self, objective_fn, parameters, gradient, hessian):
        """newton_raphson.py:194-211"""
        dp, lowest_eigenvalue = self.newton_step(gradient, hessian)
        new_parameters, new_energy = self.backtracking(objective_fn, parameters, dp, gradient)
        return new_parameters, lowest_eigenvalue


class BatchedNewtonStep(NewtonStep):
    """Extension (not in the reference): the same damped Newton step for G INDEPENDENT problems at
    once -- e.g. the geometries of a rank's shard, or independent Berry-phase loops in lockstep.
    One batched ``eigh`` replaces G sequential ones (331 x 331 on MI355X: 16 ms for 64 matrices
    against 7.3 ms each), and the line search synchronises with the host once per trial instead of
    once per problem.  Per problem the arithmetic is that of ``NewtonStep`` (newton_raphson.py:78-211)."""

    def newton_steps(self, gradients, hessians):
        """gradients [G, n], hessians [G, n, n] -> (dp [G, n], lowest eigenvalues [G])"""
        v, w = torch.linalg.eigh(hessians)
        low = v[:, 0]
        if self.aug:
            shift = torch.where(low < self.lambda_min, self.mu + self.rho * low.abs(),
                                torch.zeros_like(low))
            v = v + shift[:, None]
        proj = torch.einsum("gji,gj->gi", w, gradients)          # w^T g
        dp = -torch.einsum("gij,gj->gi", w, proj / v)
        return dp, low

    def damped_newton_steps(self, objective_fns, parameters, gradients, hessians):
        """objective_fns[g](*parameters[g]) -> 0-d tensor; parameters[g] = tuple of tensors.
        Returns (list of new parameter tuples, lowest Hessian eigenvalues [G])."""
        G = len(objective_fns)
        dp, low = self.newton_steps(gradients, hessians)
        shapes = [[tuple(p.shape) for p in ps] for ps in parameters]
        flat = torch.stack([torch.cat([p.flatten() for p in ps]) for ps in parameters])

        def evaluate(points):
            return torch.stack([objective_fns[g](*split_list_shapes(points[g], shapes[g])).reshape(())
                                for g in range(G)])

        energy = evaluate(flat)
        slope = self.alpha * (gradients * dp).sum(dim=1)          # wolfe(t) = t * slope
        t = torch.ones(G, dtype=flat.dtype, device=flat.device)
        test = evaluate(flat + t[:, None] * dp)
        active = test > energy + t * slope
        num = 0
        while bool(active.any()):
            if bool((slope[active] >= 0).any()):
                raise AssertionError("Newton direction is not a descent direction")
            t = torch.where(active, self.beta * t, t)
            num += 1
            if num > self.lmax:
                # newton_raphson.py:177-183: give up on the problems still failing
                t = torch.where(active, torch.zeros_like(t), t)
                if self.verbose:
                    print("Warning: line search failed. Output previous parameters.")
                break
            trial = evaluate(flat + t[:, None] * dp)
            test = torch.where(active, trial, test)
            active = active & (test > energy + t * slope)
        newp = flat + t[:, None] * dp
        return [tuple(split_list_shapes(newp[g], shapes[g])) for g in range(G)], low
