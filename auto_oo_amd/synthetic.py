"""Synthetic AO-basis tensors "of the named shape"  for benchmarks and
tests: PySCF is not available, so formaldimine/cc-pVDZ is realised as random tensors with the
physical symmetries (8-fold symmetric, positive semi-definite two-electron integrals; symmetric
core Hamiltonian; SPD overlap; orthogonal OAO->MO coefficients).  ``synthetic_problem`` is numpy
on the host (the seeds the parity tests and the CPU baseline share); ``synthetic_problem_device``
builds tensors of the same construction directly on the GPU (a different random stream) for large
batches of benchmark geometries."""
import numpy as np


def synthetic_problem(nao, seed, n_aux=None, enuc=31.0):
    """Returns dict(int1e_ao, int2e_ao, overlap, oao_mo_coeff, nuc) as numpy fp64 arrays.
    g_ao = (1/N_aux) sum_L B_Lpq B_Lrs with B symmetric in (p,q)."""
    rng = np.random.default_rng(seed)
    n = nao
    n_aux = n if n_aux is None else n_aux
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = rng.uniform(0.3, 1.7, size=n)
    S = (Q * lam) @ Q.T
    A = rng.standard_normal((n, n))
    h = (A + A.T) / (2 * np.sqrt(n)) - np.diag(np.linspace(3.0, 0.0, n))
    B = rng.standard_normal((n_aux, n, n))
    B = 0.5 * (B + B.transpose(0, 2, 1))
    g = np.einsum('Lpq,Lrs->pqrs', B, B, optimize=True) / n_aux
    # bit-for-bit p<->q and r<->s symmetry, as integral packages deliver it (a no-op wherever the
    # GEMM above already produced identical values)
    g = 0.5 * (g + g.transpose(1, 0, 2, 3))
    g = 0.5 * (g + g.transpose(0, 1, 3, 2))
    Qc, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return dict(int1e_ao=h, int2e_ao=g, overlap=S, oao_mo_coeff=Qc, nuc=float(enuc))


def synthetic_problem_device(nao, seed, device, n_aux=None, enuc=31.0):
    """The same construction as ``synthetic_problem`` with torch on ``device`` (values from torch's
    generator, not numpy's): dict(int1e_ao, int2e_ao, overlap, oao_coeff = S^-1/2, oao_mo_coeff,
    nuc) as fp64 device tensors.  1024 geometries of N = 43 take seconds instead of minutes and
    never exist on the host."""
    import torch
    gen = torch.Generator(device=device).manual_seed(int(seed))
    n = nao
    n_aux = n if n_aux is None else n_aux
    kw = dict(generator=gen, dtype=torch.float64, device=device)
    Q, _ = torch.linalg.qr(torch.randn((n, n), **kw))
    lam = 0.3 + 1.4 * torch.rand(n, **kw)
    S = (Q * lam) @ Q.T
    S = 0.5 * (S + S.T)
    A = torch.randn((n, n), **kw)
    h = (A + A.T) / (2 * n ** 0.5) - torch.diag(torch.linspace(3.0, 0.0, n, dtype=torch.float64, device=device))
    B = torch.randn((n_aux, n, n), **kw)
    B = 0.5 * (B + B.transpose(1, 2))
    g = torch.einsum('Lpq,Lrs->pqrs', B, B) / n_aux
    g = 0.5 * (g + g.transpose(0, 1))
    g = 0.5 * (g + g.transpose(2, 3))
    Qc, _ = torch.linalg.qr(torch.randn((n, n), **kw))
    # S^-1/2 from the eigen-decomposition (moldata.ao_to_oao)
    w, V = torch.linalg.eigh(S)
    oao = (V * w.rsqrt()) @ V.T
    return dict(int1e_ao=h.contiguous(), int2e_ao=g.contiguous(), overlap=S, oao_coeff=oao.contiguous(),
                oao_mo_coeff=Qc.contiguous(), nuc=float(enuc))


def synthetic_loop(nao, seed, n_geom, eps=0.01, n_aux=None, enuc=31.0):
    """A closed loop of ``n_geom`` nearby geometries around a base problem -- the setting of the
    Berry-phase loop (examples/Tutorial_Berry_phase.ipynb: every loop point is a small displacement of
    its neighbour, and the state is tracked by ONE Newton step per point from the previous point's
    solution).  Geometry g displaces the three factors the synthetic integrals are built from by
    ``eps (cos phi_g D1 + sin phi_g D2)``, phi_g = 2 pi g / n_geom: the auxiliary factors B of
    g_ao = (1/N_aux) sum_L B_Lpq B_Lrs (so g_ao stays 8-fold symmetric and positive semi-definite), the
    symmetric part of the core Hamiltonian and the spectrum of the overlap (its eigenvectors are kept).
    Returns (base problem, list of n_geom problems), each a dict like ``synthetic_problem``'s; all share
    the base's ``oao_mo_coeff`` as a starting guess."""
    rng = np.random.default_rng(seed)
    n = nao
    n_aux = n if n_aux is None else n_aux
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = rng.uniform(0.3, 1.7, size=n)
    A = rng.standard_normal((n, n))
    B = rng.standard_normal((n_aux, n, n))
    Qc, _ = np.linalg.qr(rng.standard_normal((n, n)))
    dlam = [rng.uniform(-0.3, 0.3, size=n) for _ in range(2)]
    dA = [rng.standard_normal((n, n)) for _ in range(2)]
    dB = [rng.standard_normal((n_aux, n, n)) for _ in range(2)]

    def build(c1, c2):
        lam_g = lam + eps * (c1 * dlam[0] + c2 * dlam[1])
        S = (Q * lam_g) @ Q.T
        S = 0.5 * (S + S.T)
        A_g = A + eps * (c1 * dA[0] + c2 * dA[1])
        h = (A_g + A_g.T) / (2 * np.sqrt(n)) - np.diag(np.linspace(3.0, 0.0, n))
        B_g = B + eps * (c1 * dB[0] + c2 * dB[1])
        B_g = 0.5 * (B_g + B_g.transpose(0, 2, 1))
        g = np.einsum('Lpq,Lrs->pqrs', B_g, B_g, optimize=True) / n_aux
        g = 0.5 * (g + g.transpose(1, 0, 2, 3))
        g = 0.5 * (g + g.transpose(0, 1, 3, 2))
        return dict(int1e_ao=h, int2e_ao=g, overlap=S, oao_mo_coeff=Qc, nuc=float(enuc))

    base = build(0.0, 0.0)
    loop = [build(np.cos(2 * np.pi * k / n_geom), np.sin(2 * np.pi * k / n_geom)) for k in range(n_geom)]
    return base, loop
