"""AO integrals and RHF orbitals for s/p Gaussian basis sets (STO-3G), on the host.

SURVEY.md section 8(f) rank 3: the reference's ``Moldata_pyscf`` (src/auto_oo/moldata_pyscf.py:19-61)
gets its arrays from PySCF / libcint -- ``int1e_kin + int1e_nuc``, ``int2e``, ``int1e_ovlp``,
``energy_nuc()`` (:28-35), RHF orbitals (:58-61) -- which this container does not have.  For the
basis set of every STO-3G case in the reference's tests and notebooks (formaldimine: N, C, H) the
same quantities are computed here from scratch: contracted s/p Cartesian Gaussians, McMurchie-
Davidson Hermite expansion (overlap, kinetic, nuclear attraction, electron repulsion; Boys function
through the confluent hypergeometric function), a DIIS-accelerated RHF, and PySCF's Z-matrix ->
Cartesian convention.  Third-party algorithm restated: PySCF ``gto.mole.from_zmatrix`` and libcint
conventions (atoms in input order; per atom 1s, 2s, 2px, 2py, 2pz; contracted functions normalised;
1 Bohr = 0.52917721092 Angstrom).  Pinned by the reference's own literals: ``S^-1/2`` of
formaldimine at (alpha, phi) = (140, 80) (test/test_moldata_pyscf.py:21-85), the RHF energy
-92.66372193556138 (test/test_oo_energy.py:396) and the CAS(2,2) energy -92.74923236954386 at the
literal orbitals (test/test_oo_energy.py:298).

This is AO-integral preparation -- outside the hot path, once per geometry, numpy on the host like
the reference's libcint calls; the hot path starts at the arrays it returns.  cc-pVDZ (d shells,
general contractions) is out of scope.
"""
import itertools
from types import SimpleNamespace

import numpy as np
from scipy.special import hyp1f1

from .moldata import Moldata

BOHR = 0.52917721092        # Angstrom; the value PySCF converts with

# STO-3G (EMSL / Basis Set Exchange): exponents, s coefficients, p coefficients per shell
_STO3G_1S_COEF = (0.15432897, 0.53532814, 0.44463454)
_STO3G_2S_COEF = (-0.09996723, 0.39951283, 0.70011547)
_STO3G_2P_COEF = (0.15591627, 0.60768372, 0.39195739)
_STO3G = {
    "H": {"Z": 1, "1s": (3.42525091, 0.62391373, 0.16885540)},
    "C": {"Z": 6, "1s": (71.6168370, 13.0450960, 3.5305122), "2sp": (2.9412494, 0.6834831, 0.2222899)},
    "N": {"Z": 7, "1s": (99.1061690, 18.0523120, 4.8856602), "2sp": (3.7804559, 0.8784966, 0.2857144)},
    "O": {"Z": 8, "1s": (130.7093200, 23.8088610, 6.4436083), "2sp": (5.0331513, 1.1695961, 0.3803890)},
}


# ---------------------------------------------------------------------------------------------
# geometry
# ---------------------------------------------------------------------------------------------
def _rotation(axis, angle):
    """Rotation matrix about `axis` by `angle` (Rodrigues)."""
    axis = np.asarray(axis, dtype=float)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def zmatrix_to_cartesian(zmat):
    """Z-matrix text (one atom per line: ``sym [ref dist [ref angle [ref dihedral]]]``, 1-based
    references, Angstrom / degrees) -> (symbols, coordinates [natm, 3] in Angstrom), laid out as
    PySCF does: first atom at the origin, second on +x, third rotated about z x (bond vector)."""
    symbols, coord = [], []
    for line in zmat.replace(";", "\n").replace(",", " ").splitlines():
        tok = line.split()
        if not tok or tok[0].startswith("#"):
            continue
        symbols.append(tok[0])
        if len(tok) < 3:
            coord.append(np.zeros(3))
        elif len(tok) == 3:
            coord.append(np.array([float(tok[2]), 0.0, 0.0]))
        else:
            bonda, bond = int(tok[1]) - 1, float(tok[2])
            anga, ang = int(tok[3]) - 1, float(tok[4]) / 180.0 * np.pi
            v1 = coord[anga] - coord[bonda]
            if len(tok) == 5:
                vecn = np.cross(v1, [0.0, 0.0, 1.0]) if not np.allclose(v1[:2], 0) else np.array([0.0, 0.0, 1.0])
                c = _rotation(vecn, ang) @ v1 * (bond / np.linalg.norm(v1))
            else:
                v1 = v1 / np.linalg.norm(v1)
                if ang < 1e-7:
                    c = v1 * bond
                elif np.pi - ang < 1e-7:
                    c = -v1 * bond
                else:
                    diha, dih = int(tok[5]) - 1, float(tok[6]) / 180.0 * np.pi
                    v2 = coord[diha] - coord[anga]
                    vecn = np.cross(v2, -v1)
                    norm = np.linalg.norm(vecn)
                    if norm < 1e-7:
                        vecn = (np.cross(v1, [0.0, 0.0, 1.0]) if not np.allclose(v1[:2], 0)
                                else np.array([0.0, 0.0, 1.0]))
                        c = _rotation(vecn, ang) @ v1 * bond
                    else:
                        vecn = _rotation(v1, -dih) @ vecn / norm
                        c = _rotation(vecn, ang) @ v1 * bond
            coord.append(coord[bonda] + c)
    return symbols, np.array(coord)


# ---------------------------------------------------------------------------------------------
# basis functions
# ---------------------------------------------------------------------------------------------
class _Shell:
    """One contracted Cartesian Gaussian x^l y^m z^n sum_k c_k exp(-a_k r^2) on a centre."""

    def __init__(self, center, lmn, exps, coefs):
        self.center = np.asarray(center, dtype=float)
        self.lmn = tuple(lmn)
        self.exps = np.asarray(exps, dtype=float)
        L = sum(lmn)
        prim_norm = (2 * self.exps / np.pi) ** 0.75 * (4 * self.exps) ** (L / 2.0)    # l, m, n <= 1
        self.coefs = np.asarray(coefs, dtype=float) * prim_norm
        # normalise the contraction
        a, b = self.exps[:, None], self.exps[None, :]
        s = (np.pi / (a + b)) ** 1.5 / (2 * (a + b)) ** L
        self.coefs = self.coefs / np.sqrt(self.coefs @ s @ self.coefs)


def sto3g_basis(symbols, coords_bohr):
    shells = []
    for sym, R in zip(symbols, coords_bohr):
        key = sym.capitalize()
        if key not in _STO3G:
            raise ValueError(f"no STO-3G parameters for element {sym!r} (H, C, N, O are built in)")
        par = _STO3G[key]
        shells.append(_Shell(R, (0, 0, 0), par["1s"], _STO3G_1S_COEF))
        if "2sp" in par:
            shells.append(_Shell(R, (0, 0, 0), par["2sp"], _STO3G_2S_COEF))
            for lmn in ((1, 0, 0), (0, 1, 0), (0, 0, 1)):
                shells.append(_Shell(R, lmn, par["2sp"], _STO3G_2P_COEF))
    return shells


# ---------------------------------------------------------------------------------------------
# McMurchie-Davidson machinery (arrays broadcast over primitives)
# ---------------------------------------------------------------------------------------------
def _E(i, j, t, Q, a, b):
    """Hermite expansion coefficient E_t^{ij} of the product of two 1-D Gaussians, Q = A - B."""
    p = a + b
    mu = a * b / p
    if t < 0 or t > i + j:
        return np.zeros(np.broadcast(Q, a, b).shape)
    if i == j == t == 0:
        return np.exp(-mu * Q * Q) * np.ones(np.broadcast(Q, a, b).shape)
    if j == 0:
        return (_E(i - 1, j, t - 1, Q, a, b) / (2 * p) - (mu * Q / a) * _E(i - 1, j, t, Q, a, b)
                + (t + 1) * _E(i - 1, j, t + 1, Q, a, b))
    return (_E(i, j - 1, t - 1, Q, a, b) / (2 * p) + (mu * Q / b) * _E(i, j - 1, t, Q, a, b)
            + (t + 1) * _E(i, j - 1, t + 1, Q, a, b))


def _boys(n, x):
    return hyp1f1(n + 0.5, n + 1.5, -x) / (2.0 * n + 1.0)


def _R(t, u, v, n, p, PC, T, cache):
    """Hermite Coulomb integral R^n_{tuv}(p, P - C), T = p |PC|^2."""
    key = (t, u, v, n)
    if key in cache:
        return cache[key]
    if t < 0 or u < 0 or v < 0:
        val = 0.0
    elif t == u == v == 0:
        val = (-2.0 * p) ** n * _boys(n, T)
    elif t == u == 0:
        val = PC[2] * _R(t, u, v - 1, n + 1, p, PC, T, cache)
        if v > 1:
            val = val + (v - 1) * _R(t, u, v - 2, n + 1, p, PC, T, cache)
    elif t == 0:
        val = PC[1] * _R(t, u - 1, v, n + 1, p, PC, T, cache)
        if u > 1:
            val = val + (u - 1) * _R(t, u - 2, v, n + 1, p, PC, T, cache)
    else:
        val = PC[0] * _R(t - 1, u, v, n + 1, p, PC, T, cache)
        if t > 1:
            val = val + (t - 1) * _R(t - 2, u, v, n + 1, p, PC, T, cache)
    cache[key] = val
    return val


class _Pair:
    """Primitive-pair data of two shells: exponent sums, product centres, Hermite coefficients."""

    def __init__(self, A, B):
        a, b = A.exps[:, None], B.exps[None, :]
        self.p = a + b
        self.P = (a[..., None] * A.center + b[..., None] * B.center) / self.p[..., None]   # [na, nb, 3]
        self.cc = A.coefs[:, None] * B.coefs[None, :]
        Q = A.center - B.center
        self.E = []          # per dimension: list over t of arrays [na, nb]
        for d in range(3):
            i, j = A.lmn[d], B.lmn[d]
            self.E.append([_E(i, j, t, Q[d], a, b) for t in range(i + j + 1)])
        self.tuv = [(t, u, v) for t in range(len(self.E[0])) for u in range(len(self.E[1]))
                    for v in range(len(self.E[2]))]

    def herm(self, t, u, v):
        return self.E[0][t] * self.E[1][u] * self.E[2][v]


def _overlap_1d(i, j, Q, a, b):
    return _E(i, j, 0, Q, a, b) * np.sqrt(np.pi / (a + b))


def one_electron_integrals(shells, charges, centers):
    """-> (S, T, V): overlap, kinetic energy, nuclear attraction, [nao, nao]."""
    n = len(shells)
    S, T, V = np.zeros((n, n)), np.zeros((n, n)), np.zeros((n, n))
    for ia, A in enumerate(shells):
        for ib in range(ia + 1):
            B = shells[ib]
            a, b = A.exps[:, None], B.exps[None, :]
            cc = A.coefs[:, None] * B.coefs[None, :]
            Q = A.center - B.center
            s1 = [_overlap_1d(A.lmn[d], B.lmn[d], Q[d], a, b) for d in range(3)]
            S[ia, ib] = S[ib, ia] = np.sum(cc * s1[0] * s1[1] * s1[2])
            kin = 0.0
            for d in range(3):
                i, j = A.lmn[d], B.lmn[d]
                td = -2.0 * b * b * _overlap_1d(i, j + 2, Q[d], a, b) + b * (2 * j + 1) * s1[d]
                if j >= 2:
                    td = td - 0.5 * j * (j - 1) * _overlap_1d(i, j - 2, Q[d], a, b)
                others = [s1[e] for e in range(3) if e != d]
                kin = kin + td * others[0] * others[1]
            T[ia, ib] = T[ib, ia] = np.sum(cc * kin)
            pair = _Pair(A, B)
            acc = 0.0
            for Z, C in zip(charges, centers):
                PC = [pair.P[..., d] - C[d] for d in range(3)]
                Tt = pair.p * (PC[0] ** 2 + PC[1] ** 2 + PC[2] ** 2)
                cache = {}
                term = 0.0
                for (t, u, v) in pair.tuv:
                    term = term + pair.herm(t, u, v) * _R(t, u, v, 0, pair.p, PC, Tt, cache)
                acc = acc - Z * np.sum(pair.cc * term * 2.0 * np.pi / pair.p)
            V[ia, ib] = V[ib, ia] = acc
    return S, T, V


def electron_repulsion_integrals(shells):
    """(pq|rs) in chemist order, [nao]*4, 8-fold symmetry filled in exactly (bit-for-bit)."""
    n = len(shells)
    pairs = {(i, j): _Pair(shells[i], shells[j]) for i in range(n) for j in range(i + 1)}
    g = np.zeros((n, n, n, n))
    idx = [(i, j) for i in range(n) for j in range(i + 1)]
    for k1, (i, j) in enumerate(idx):
        ab = pairs[(i, j)]
        p = ab.p[:, :, None, None]
        for (k, l) in idx[:k1 + 1]:
            cd = pairs[(k, l)]
            q = cd.p[None, None, :, :]
            alpha = p * q / (p + q)
            PQ = [ab.P[:, :, None, None, d] - cd.P[None, None, :, :, d] for d in range(3)]
            Tt = alpha * (PQ[0] ** 2 + PQ[1] ** 2 + PQ[2] ** 2)
            cache = {}
            val = 0.0
            for (t, u, v) in ab.tuv:
                e1 = ab.herm(t, u, v)[:, :, None, None]
                for (tt, uu, vv) in cd.tuv:
                    e2 = cd.herm(tt, uu, vv)[None, None, :, :]
                    sign = -1.0 if (tt + uu + vv) % 2 else 1.0
                    val = val + sign * e1 * e2 * _R(t + tt, u + uu, v + vv, 0, alpha, PQ, Tt, cache)
            val = val * 2.0 * np.pi ** 2.5 / (p * q * np.sqrt(p + q))
            x = np.sum(ab.cc[:, :, None, None] * cd.cc[None, None, :, :] * val)
            for (a_, b_, c_, d_) in set(itertools.chain(
                    [(i, j, k, l), (j, i, k, l), (i, j, l, k), (j, i, l, k),
                     (k, l, i, j), (l, k, i, j), (k, l, j, i), (l, k, j, i)])):
                g[a_, b_, c_, d_] = x
    return g


# ---------------------------------------------------------------------------------------------
# molecule
# ---------------------------------------------------------------------------------------------
def rhf(int1e, int2e, overlap, n_occ, conv_tol=1e-12, max_cycle=200):
    """Restricted Hartree-Fock with DIIS from the core-Hamiltonian guess.
    -> (mo_coeff [nao, nao], mo_energy, electronic energy)."""
    s_val, s_vec = np.linalg.eigh(overlap)
    X = s_vec @ np.diag(s_val ** -0.5) @ s_vec.T

    def diag(F):
        e, c = np.linalg.eigh(X.T @ F @ X)
        return e, X @ c
    e, C = diag(int1e)
    D = 2.0 * C[:, :n_occ] @ C[:, :n_occ].T
    fs, errs = [], []
    energy = 0.0
    for _ in range(max_cycle):
        J = np.einsum("pqrs,rs->pq", int2e, D)
        K = np.einsum("prqs,rs->pq", int2e, D)
        F = int1e + J - 0.5 * K
        new_energy = 0.5 * np.sum(D * (int1e + F))
        err = F @ D @ overlap - overlap @ D @ F
        fs.append(F)
        errs.append(err)
        fs, errs = fs[-8:], errs[-8:]
        if len(fs) > 1:
            m = len(fs)
            B = -np.ones((m + 1, m + 1))
            B[m, m] = 0.0
            for a_ in range(m):
                for b_ in range(m):
                    B[a_, b_] = np.sum(errs[a_] * errs[b_])
            rhs = np.zeros(m + 1)
            rhs[m] = -1.0
            try:
                w = np.linalg.solve(B, rhs)[:m]
                F = sum(wi * fi for wi, fi in zip(w, fs))
            except np.linalg.LinAlgError:
                pass
        e, C = diag(F)
        D = 2.0 * C[:, :n_occ] @ C[:, :n_occ].T
        if abs(new_energy - energy) < conv_tol and np.abs(err).max() < 1e-9:
            energy = new_energy
            break
        energy = new_energy
    return C, e, energy


class Moldata_sto3g(Moldata):
    """Stand-in for ``Moldata_pyscf(geometry, 'sto-3g')`` (src/auto_oo/moldata_pyscf.py:19-61) for
    molecules of H, C, N, O: same attributes (``int1e_ao, int2e_ao, overlap, oao_coeff, nuc, nao``),
    ``run_rhf()`` -> ``hf.mo_coeff`` / ``hf.e_tot``.  ``geometry`` is a Z-matrix string (as
    ``get_formal_geo`` returns) or a list of ``(symbol, (x, y, z))`` in Angstrom."""

    def __init__(self, geometry, basis="sto-3g", charge=0):
        if str(basis).lower().replace("-", "") != "sto3g":
            raise ValueError("Moldata_sto3g only builds STO-3G integrals; import others with Moldata.from_npz")
        if isinstance(geometry, str):
            symbols, xyz = zmatrix_to_cartesian(geometry)
        else:
            symbols = [a[0] for a in geometry]
            xyz = np.array([a[1] for a in geometry], dtype=float)
        self.symbols = symbols
        self.coordinates = xyz
        R = xyz / BOHR
        charges = [_STO3G[s.capitalize()]["Z"] for s in symbols]
        shells = sto3g_basis(symbols, R)
        S, T, V = one_electron_integrals(shells, charges, R)
        g = electron_repulsion_integrals(shells)
        nuc = sum(charges[i] * charges[j] / np.linalg.norm(R[i] - R[j])
                  for i in range(len(charges)) for j in range(i))
        super().__init__(T + V, g, S, nuc, sum(charges) - charge)

    def run_rhf(self, verbose=0):
        """moldata_pyscf.py:58-61"""
        if self.hf is None:
            C, e, e_elec = rhf(self.int1e_ao, self.int2e_ao, self.overlap, self.nelectron // 2)
            self.hf = SimpleNamespace(mo_coeff=C, mo_energy=e, e_tot=e_elec + self.nuc)
