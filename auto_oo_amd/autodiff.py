"""torch as the autodiff container of the cost functions.

The reference defines its derivatives by differentiating ``energy_from_parameters(theta, kappa)``
and ``energy_from_kappa(kappa, one_rdm, two_rdm)`` with torch autograd
(src/auto_oo/oo_pqc.py:86-95,103-125; test/test_oo_pqc.py:113-125; test/test_oo_energy.py:930-943).
Here the numbers come from HIP kernels, so the cost functions are presented to torch as
``torch.autograd.Function``s whose derivative rules call the analytic kernels:

    ScalarValue   E(x_1..x_n)                   backward / jvp -> ScalarGradient
    ScalarGradient (dE/dx_1 .. dE/dx_n)         backward / jvp -> model.hvp (second derivatives)

Both carry ``setup_context`` / ``jvp`` / ``vmap`` rules, so ``torch.autograd.functional.jacobian``
/ ``hessian`` and ``torch.func.jacrev`` / ``jacfwd`` / ``hessian`` all work through them.  This
module is pure host logic (no kernel calls): a *model* object supplies

    model.value(*xs)      -> 0-d tensor
    model.grad(*xs)       -> tuple, one tensor per input (None where not differentiable)
    model.hvp(xs, vs, needs=None) -> tuple, row i = sum_j (d^2E/dx_i dx_j) . v_j  (vs[j] may be
                             None = zero; needs[i] tells which rows the caller will use)

The kernel-backed models live next to the classes that own the kernels (oo_energy.py, oo_pqc.py).
"""
import contextlib

import torch
from torch._functorch.pyfunctorch import temporarily_clear_interpreter_stack


@contextlib.contextmanager
def kernel_scope():
    """Run kernel-calling code on plain tensors: inside a torch.func transform every torch op --
    even a reshape of a plain tensor -- returns a tensor wrapped at the current level, which has no
    data pointer to hand to the C ABI.  This scope suspends the transform stack (and grad mode);
    derivative rules compute their kernel results inside it and combine them with the (wrapped)
    tangents outside."""
    with temporarily_clear_interpreter_stack(), torch.no_grad():
        yield


def needs_autodiff(*tensors):
    """True when torch is tracking derivatives through any of the arguments: a requires_grad tensor
    under grad mode, or any torch.func transform (their wrapped tensors carry no data pointer)."""
    for t in tensors:
        if isinstance(t, torch.Tensor):
            if torch._C._functorch.is_functorch_wrapped_tensor(t):
                return True
            if t.requires_grad and torch.is_grad_enabled():
                return True
    return False


def unwrap(t):
    """The plain tensor under torch.func's wrappers (saved tensors reach a derivative rule wrapped
    once per enclosing transform; the kernels need the storage).  Only for values that are not
    batched -- the cost functions are differentiated at one parameter point at a time."""
    while isinstance(t, torch.Tensor) and torch._C._functorch.is_functorch_wrapped_tensor(t):
        if torch._C._functorch.is_batchedtensor(t):
            raise NotImplementedError("the analytic derivative rules take one parameter point at a "
                                      "time (a batched primal reached a second-derivative rule)")
        t = torch._C._functorch.get_unwrapped(t)
    return t


def is_zero_tangent(t):
    """True for None and for a tangent / cotangent that is identically zero in every batch entry
    (torch.func hands zeros, not None, to a jvp rule for inputs that are not differentiated).
    Looks at the raw data under the wrappers, so it is safe under vmap."""
    if t is None:
        return True
    while isinstance(t, torch.Tensor) and torch._C._functorch.is_functorch_wrapped_tensor(t):
        t = torch._C._functorch.get_unwrapped(t)
    return not bool((t != 0).any())


def _loop_vmap(apply, n_out):
    """vmap rule by looping over the batch (the kernels take one parameter point per call)."""
    def vmap(info, in_dims, model, *xs):
        dims = in_dims[1:]
        outs = []
        for b in range(info.batch_size):
            args = [x if d is None else x.select(d, b) for x, d in zip(xs, dims)]
            outs.append(apply(model, *args))
        if n_out is None:
            return torch.stack(outs), 0
        cols = list(zip(*outs))
        return tuple(None if c[0] is None else torch.stack(c) for c in cols), \
            tuple(None if c[0] is None else 0 for c in cols)
    return vmap


def _dot(a, b):
    return (a * b).sum()


class ScalarGradient(torch.autograd.Function):
    """(x_1..x_n) -> (dE/dx_1 .. dE/dx_n) with the model's analytic gradient; its own derivative
    is the model's Hessian-vector product."""

    @staticmethod
    def forward(model, *xs):
        with kernel_scope():
            gs = model.grad(*xs)
        return tuple(torch.zeros_like(x) if g is None else g for x, g in zip(xs, gs))

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.model = inputs[0]
        ctx.n = len(inputs) - 1
        ctx.save_for_backward(*inputs[1:])
        ctx.save_for_forward(*inputs[1:])

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *vs):
        xs = ctx.saved_tensors
        # the Hessian is symmetric: vjp == jvp
        hv = ctx.model.hvp(xs, vs, needs=tuple(ctx.needs_input_grad[1:]))
        return (None,) + tuple(h if need else None
                               for h, need in zip(hv, ctx.needs_input_grad[1:]))

    @staticmethod
    def jvp(ctx, _model_tangent, *ts):
        xs = ctx.saved_tensors
        hv = ctx.model.hvp(xs, ts)
        return tuple(torch.zeros_like(x) if h is None else h for x, h in zip(xs, hv))


ScalarGradient.vmap = staticmethod(_loop_vmap(lambda m, *a: ScalarGradient.apply(m, *a), 0))


class ScalarValue(torch.autograd.Function):
    """(x_1..x_n) -> E with the model's value; backward and jvp go through ScalarGradient, so
    second derivatives are reachable (create_graph=True, jacfwd over jacrev, ...)."""

    @staticmethod
    def forward(model, *xs):
        with kernel_scope():
            return model.value(*xs)

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.model = inputs[0]
        ctx.save_for_backward(*inputs[1:])
        ctx.save_for_forward(*inputs[1:])

    @staticmethod
    def backward(ctx, gE):
        xs = ctx.saved_tensors
        gs = ScalarGradient.apply(ctx.model, *xs)
        return (None,) + tuple(gE * g if need else None
                               for g, need in zip(gs, ctx.needs_input_grad[1:]))

    @staticmethod
    def jvp(ctx, _model_tangent, *ts):
        xs = ctx.saved_tensors
        gs = ScalarGradient.apply(ctx.model, *xs)
        out = None
        for g, t in zip(gs, ts):
            if t is not None:
                term = _dot(g, t)
                out = term if out is None else out + term
        return out if out is not None else torch.zeros((), dtype=xs[0].dtype, device=xs[0].device)


ScalarValue.vmap = staticmethod(_loop_vmap(lambda m, *a: ScalarValue.apply(m, *a), None))


def differentiable_scalar(model, *xs):
    """E = model.value(*xs) as a node of torch's autodiff graph (see module docstring)."""
    return ScalarValue.apply(model, *xs)


class VectorWithJacobian(torch.autograd.Function):
    """x -> (y_1..y_m) with analytic Jacobians: ``fn(x)`` returns ``(ys, jacs)`` where
    ``jacs[i]`` has shape ``x.shape + ys[i].shape`` flattened as [x.numel(), *ys[i].shape]
    (first order only; used for the circuit's RDMs as functions of theta)."""

    @staticmethod
    def forward(fn, x):
        with kernel_scope():
            ys, jacs = fn(x)
        return tuple(ys) + tuple(jacs)

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.m = len(output) // 2
        ctx.shape = inputs[1].shape
        ctx.save_for_backward(*output[ctx.m:])
        ctx.save_for_forward(*output[ctx.m:])
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*output[ctx.m:])

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        jacs = ctx.saved_tensors
        total = None
        for g, J in zip(gs[:ctx.m], jacs):
            if g is None:
                continue
            term = (J.reshape(J.shape[0], -1) * g.reshape(1, -1)).sum(dim=1)
            total = term if total is None else total + term
        if total is None:
            return None, None
        return None, total.reshape(ctx.shape)

    @staticmethod
    def jvp(ctx, _fn_tangent, t):
        jacs = ctx.saved_tensors
        tv = t.reshape(-1)
        outs = tuple(torch.tensordot(tv, J, dims=1) for J in jacs)
        return outs + tuple(None for J in jacs)


def _vector_vmap(info, in_dims, fn, x):
    d = in_dims[1]
    outs = [VectorWithJacobian.apply(fn, x.select(d, b)) for b in range(info.batch_size)]
    cols = list(zip(*outs))
    return tuple(torch.stack(c) for c in cols), tuple(0 for _ in cols)


VectorWithJacobian.vmap = staticmethod(_vector_vmap)
