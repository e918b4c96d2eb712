"""Host-logic test of the autodiff container (auto_oo_amd/autodiff.py) on CPU: a toy analytic
model stands in for the kernels; every transform the reference's tests apply to its cost functions
(torch.autograd.functional.jacobian / hessian, test/test_oo_pqc.py:113-125; torch.func.jacrev /
hessian, test/test_oo_energy.py:930-943) must reproduce plain-torch autodiff of the same formula."""
import torch
from torch.autograd.functional import hessian as thessian, jacobian as tjacobian

from auto_oo_amd.autodiff import VectorWithJacobian, differentiable_scalar, needs_autodiff

torch.set_default_dtype(torch.float64)


def plain(x, y):
    return (torch.sin(x) * x.flip(0)).sum() * torch.cos(y).sum() + (x ** 2).sum() * y[0] * y[1]


class ToyModel:
    """value / grad / hvp written out by hand (no autograd inside, like the kernels)."""
    calls = {"value": 0, "grad": 0, "hvp": 0}

    def value(self, x, y):
        self.calls["value"] += 1
        assert x.reshape(-1).data_ptr() and y.detach().data_ptr()
        with torch.no_grad():
            return plain(x, y)

    def _pieces(self, x, y):
        xf = x.flip(0)
        s = (torch.sin(x) * xf).sum()
        ds = torch.cos(x) * xf + torch.sin(xf)            # d s / d x
        c = torch.cos(y).sum()
        dc = -torch.sin(y)
        return s, ds, c, dc

    def grad(self, x, y):
        self.calls["grad"] += 1
        assert x.reshape(-1).data_ptr() and y.detach().data_ptr()
        s, ds, c, dc = self._pieces(x, y)
        yy = torch.stack((y[1], y[0]))
        return ds * c + 2 * x * y[0] * y[1], s * dc + (x ** 2).sum() * yy

    def hvp(self, xs, vs, needs=None):
        self.calls["hvp"] += 1
        from auto_oo_amd.autodiff import unwrap
        from auto_oo_amd.autodiff import kernel_scope
        x, y = (unwrap(t) for t in xs)
        with kernel_scope():
            # what a kernel call needs: real storage, also after a view op (which a transform
            # level would wrap again outside kernel_scope)
            assert x.data_ptr() and y.reshape(1, -1).data_ptr() and x.detach().data_ptr()
        vx = torch.zeros_like(x) if vs[0] is None else vs[0]
        vy = torch.zeros_like(y) if vs[1] is None else vs[1]
        n = x.numel()
        s, ds, c, dc = self._pieces(x, y)
        xf = x.flip(0)
        # d ds_i / d x_j
        Hs = torch.diag(-torch.sin(x) * xf)
        P = torch.eye(n).flip(0)
        Hs = Hs + torch.cos(x)[:, None] * P + P * torch.cos(xf)[:, None]
        Hxx = Hs * c + 2 * torch.eye(n) * y[0] * y[1]
        yy = torch.stack((y[1], y[0]))
        Hxy = ds[:, None] * dc[None, :] + 2 * x[:, None] * yy[None, :]
        Hyy = torch.diag(-torch.cos(y) * s) + (x ** 2).sum() * torch.tensor([[0., 1.], [1., 0.]])
        return Hxx @ vx + Hxy @ vy, Hxy.T @ vx + Hyy @ vy


def f(x, y):
    return differentiable_scalar(ToyModel(), x, y)


def _pt():
    g = torch.Generator().manual_seed(3)
    return torch.randn(3, generator=g), torch.randn(2, generator=g)


def test_toy_model_is_consistent():
    x, y = _pt()
    m = ToyModel()
    gx, gy = tjacobian(plain, (x, y))
    mx, my = m.grad(x, y)
    assert torch.allclose(gx, mx, atol=1e-13) and torch.allclose(gy, my, atol=1e-13)
    H = thessian(plain, (x, y))
    vx, vy = torch.randn(3), torch.randn(2)
    hx, hy = m.hvp((x, y), (vx, vy))
    assert torch.allclose(hx, H[0][0] @ vx + H[0][1] @ vy, atol=1e-12)
    assert torch.allclose(hy, H[1][0] @ vx + H[1][1] @ vy, atol=1e-12)


def test_autograd_functional_jacobian_and_hessian():
    x, y = _pt()
    ga = tjacobian(f, (x, y))
    gp = tjacobian(plain, (x, y))
    for a, p in zip(ga, gp):
        assert torch.allclose(a, p, atol=1e-13)
    Ha = thessian(f, (x, y))
    Hp = thessian(plain, (x, y))
    for i in range(2):
        for j in range(2):
            assert torch.allclose(Ha[i][j], Hp[i][j], atol=1e-12)
    # vectorised variants go through the vmap rule
    gv = tjacobian(f, (x, y), vectorize=True)
    assert torch.allclose(gv[0], gp[0], atol=1e-13)


def test_backward_and_double_backward():
    x, y = _pt()
    x = x.clone().requires_grad_(True)
    y = y.clone().requires_grad_(True)
    e = f(x, y)
    assert torch.allclose(e, plain(x, y))
    gx, gy = torch.autograd.grad(e, (x, y), create_graph=True)
    (gx.sum() + 2 * gy.sum()).backward()
    x2 = x.detach().clone().requires_grad_(True)
    y2 = y.detach().clone().requires_grad_(True)
    px, py = torch.autograd.grad(plain(x2, y2), (x2, y2), create_graph=True)
    (px.sum() + 2 * py.sum()).backward()
    assert torch.allclose(x.grad, x2.grad, atol=1e-12) and torch.allclose(y.grad, y2.grad, atol=1e-12)


def test_torch_func_transforms():
    x, y = _pt()
    for argnums in (0, 1, (0, 1)):
        ja = torch.func.jacrev(f, argnums=argnums)(x, y)
        jp = torch.func.jacrev(plain, argnums=argnums)(x, y)
        ja, jp = (ja, jp) if isinstance(ja, tuple) else ((ja,), (jp,))
        for a, p in zip(ja, jp):
            assert torch.allclose(a, p, atol=1e-13)
    assert torch.allclose(torch.func.jacfwd(f, argnums=0)(x, y), torch.func.jacfwd(plain, argnums=0)(x, y),
                          atol=1e-13)
    assert torch.allclose(torch.func.grad(f)(x, y), torch.func.grad(plain)(x, y), atol=1e-13)
    Ha = torch.func.hessian(f, argnums=0)(x, y)          # jacfwd(jacrev): test_oo_energy.py:937-940
    Hp = torch.func.hessian(plain, argnums=0)(x, y)
    assert torch.allclose(Ha, Hp, atol=1e-12)
    Hb = torch.func.jacrev(torch.func.jacrev(f, argnums=1), argnums=0)(x, y)
    Hq = torch.func.jacrev(torch.func.jacrev(plain, argnums=1), argnums=0)(x, y)
    assert torch.allclose(Hb, Hq, atol=1e-12)


def test_no_graph_when_nothing_requires_grad():
    x, y = _pt()
    assert not needs_autodiff(x, y, None, 3.0)
    assert needs_autodiff(x.clone().requires_grad_(True))
    with torch.no_grad():
        assert not needs_autodiff(x.clone().requires_grad_(True))
    seen = []
    torch.func.jacrev(lambda a: (seen.append(needs_autodiff(a)), a.sum())[1])(x)
    assert seen == [True]


def test_vector_with_jacobian():
    A = torch.randn(4, 3)
    B = torch.randn(2, 2, 3)

    def fn(t):
        with torch.no_grad():
            s = torch.sin(t)
            y1 = A @ s
            y2 = B @ (t ** 2)
            J1 = (A * torch.cos(t)[None, :]).T.contiguous()               # [3, 4]
            J2 = (B * (2 * t)[None, None, :]).permute(2, 0, 1).contiguous()  # [3, 2, 2]
        return (y1, y2), (J1, J2)

    def wrapped(t):
        out = VectorWithJacobian.apply(fn, t)
        return out[0], out[1]

    def ref(t):
        return A @ torch.sin(t), B @ (t ** 2)

    t = torch.randn(3)
    ja = tjacobian(wrapped, t)
    jp = tjacobian(ref, t)
    for a, p in zip(ja, jp):
        assert torch.allclose(a, p, atol=1e-13)
    jf = torch.func.jacfwd(wrapped)(t)
    for a, p in zip(jf, jp):
        assert torch.allclose(a, p, atol=1e-13)


def test_is_zero_tangent_under_vmap():
    from auto_oo_amd.autodiff import is_zero_tangent
    seen = []

    def f(t):
        seen.append(is_zero_tangent(t))
        return t.sum()
    torch.func.vmap(f)(torch.zeros(3, 2))
    torch.func.vmap(f)(torch.eye(3))
    assert seen == [True, False] and is_zero_tangent(None)
