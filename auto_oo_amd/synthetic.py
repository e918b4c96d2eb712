"""Synthetic AO-basis tensors "of the named shape"  for benchmarks and
tests: PySCF is not available, so formaldimine/cc-pVDZ is realised as random tensors with the
physical symmetries (8-fold symmetric, positive semi-definite two-electron integrals; symmetric
core Hamiltonian; SPD overlap; orthogonal OAO->MO coefficients).  numpy only, host side."""
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
