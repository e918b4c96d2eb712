"""Array-backed molecule container with the attributes OO_energy consumes.

The reference's ``Moldata_pyscf`` (src/auto_oo/moldata_pyscf.py:19-56) obtains AO integrals from
PySCF/libcint, which is a third-party Gaussian-integral engine outside the hot path (SURVEY.md
section 2, row 10).  ``Moldata`` takes the same arrays from the caller instead and exposes the same
duck type: ``int1e_ao, int2e_ao, overlap, oao_coeff, nuc, nao, hf.mo_coeff, run_rhf(),
get_active_space_idx()``.
"""
from types import SimpleNamespace

import numpy as np


def ao_to_oao(ovlp):
    """Orthogonal atomic orbitals in terms of atomic orbitals, S^(-1/2)
    (moldata_pyscf.py:13-16; host numpy, once per molecule as in the reference)."""
    S_eigval, S_eigvec = np.linalg.eigh(np.asarray(ovlp, dtype=np.float64))
    return S_eigvec @ np.diag(S_eigval ** (-0.5)) @ S_eigvec.T


class Moldata:
    def __init__(self, int1e_ao, int2e_ao, overlap, nuc, nelectron, mo_coeff=None):
        """
        Args:
            int1e_ao: [N,N] core Hamiltonian (kinetic + nuclear attraction) in the AO basis
            int2e_ao: [N,N,N,N] two-electron integrals, chemist order (pq|rs)
            overlap:  [N,N] AO overlap
            nuc:      nuclear repulsion energy
            nelectron: total number of electrons
            mo_coeff: optional [N,N] AO->MO coefficients standing in for ``mol.hf.mo_coeff``
        """
        self.int1e_ao = np.asarray(int1e_ao, dtype=np.float64)
        self.int2e_ao = np.asarray(int2e_ao, dtype=np.float64)
        self.overlap = np.asarray(overlap, dtype=np.float64)
        self.oao_coeff = ao_to_oao(self.overlap)
        self.nuc = float(nuc)
        self.nao = self.overlap.shape[0]
        self.nelectron = int(nelectron)
        self._mo_coeff0 = None if mo_coeff is None else np.asarray(mo_coeff, dtype=np.float64)
        self.hf = None
        self.fci = None
        self.casci = None
        self.casscf = None
        self.sa_casscf = None

    def get_active_space_idx(self, ncas, nelecas):
        """moldata_pyscf.py:42-56"""
        nelecore = self.nelectron - nelecas
        if nelecore % 2 == 1:
            raise ValueError('odd number of core electrons')
        occ_idx = np.arange(nelecore // 2)
        act_idx = (occ_idx[-1] + 1 + np.arange(ncas) if len(occ_idx) > 0 else np.arange(ncas))
        virt_idx = np.arange(act_idx[-1] + 1, self.nao)
        return occ_idx, act_idx, virt_idx

    def run_rhf(self, verbose=0):
        """moldata_pyscf.py:58-61.  No SCF engine ships with this package (PySCF is outside the
        hot path): the starting orbitals must have been supplied as ``mo_coeff``."""
        if self.hf is None:
            if self._mo_coeff0 is None:
                raise RuntimeError(
                    "Moldata has no RHF engine: pass mo_coeff=... to Moldata, or oao_mo_coeff=... "
                    "to OO_energy / OO_pqc")
            self.hf = SimpleNamespace(mo_coeff=self._mo_coeff0)

    # ---- real-molecule inputs -----------------------------------
    NPZ_KEYS = ("int1e_ao", "int2e_ao", "overlap", "nuc", "nelectron")

    def save_npz(self, path):
        """Write the arrays this container holds to ``path`` (numpy .npz, fp64)."""
        extra = {} if self._mo_coeff0 is None else {"mo_coeff": self._mo_coeff0}
        np.savez(path, int1e_ao=self.int1e_ao, int2e_ao=self.int2e_ao, overlap=self.overlap,
                 nuc=np.float64(self.nuc), nelectron=np.int64(self.nelectron), **extra)

    @classmethod
    def from_npz(cls, path):
        """Molecule from integrals produced elsewhere, e.g. with PySCF on a machine that has it
        (the quantities of moldata_pyscf.py:28-35, see ``tools/export_pyscf_npz.py``):
        ``int1e_ao`` = int1e_kin + int1e_nuc, ``int2e_ao`` = int2e (full [N,N,N,N], chemist order),
        ``overlap`` = int1e_ovlp, ``nuc`` = energy_nuc(), ``nelectron``; optional ``mo_coeff``
        (RHF orbitals).  An 8-fold packed ``int2e_ao`` of length npair*(npair+1)/2 is unpacked."""
        with np.load(path) as data:
            missing = [k for k in cls.NPZ_KEYS if k not in data]
            if missing:
                raise KeyError(f"{path}: missing arrays {missing}; expected {cls.NPZ_KEYS}")
            overlap = np.asarray(data["overlap"], dtype=np.float64)
            n = overlap.shape[0]
            g = np.asarray(data["int2e_ao"], dtype=np.float64)
            if g.ndim == 1:
                g = unpack_eri_s8(g, n)
            elif g.shape != (n, n, n, n):
                raise ValueError(f"{path}: int2e_ao has shape {g.shape}, expected {(n, n, n, n)} or "
                                 "the 8-fold packed vector")
            mo = np.asarray(data["mo_coeff"], dtype=np.float64) if "mo_coeff" in data else None
            return cls(data["int1e_ao"], g, overlap, float(data["nuc"]), int(data["nelectron"]),
                       mo_coeff=mo)


def unpack_eri_s8(packed, n):
    """8-fold symmetric two-electron integrals (lower-triangular pairs of lower-triangular pairs,
    the layout of PySCF's ``aosym='s8'``) -> full [n,n,n,n] chemist-order tensor."""
    npair = n * (n + 1) // 2
    packed = np.asarray(packed, dtype=np.float64)
    if packed.shape != (npair * (npair + 1) // 2,):
        raise ValueError(f"packed ERI has {packed.shape[0]} elements, expected "
                         f"{npair * (npair + 1) // 2} for n = {n}")
    tri = np.zeros((npair, npair))
    ij = np.tril_indices(npair)
    tri[ij] = packed
    tri = tri + tri.T - np.diag(np.diag(tri))
    pq = np.tril_indices(n)
    full = np.zeros((n, n, npair))
    full[pq[0], pq[1], :] = tri
    full[pq[1], pq[0], :] = tri
    out = np.zeros((n, n, n, n))
    out[:, :, pq[0], pq[1]] = full
    out[:, :, pq[1], pq[0]] = full
    return out


def get_formal_geo(alpha, phi):
    """Z-matrix of formaldimine H2C=NH with the N-H bond at bending angle ``alpha`` (degrees,
    H-N-C) and dihedral ``phi`` (degrees), the molecule of every reference test and notebook
    (src/auto_oo/utils/miscellaneous.py:34-45: N-C 1.498047, C-H 1.066797, N-H 0.987109 Angstrom,
    H-C-N 118.359375 degrees).  Returns the multi-line Z-matrix string PySCF-style parsers take."""
    r_nc, r_ch, r_nh, a_hcn = 1.498047, 1.066797, 0.987109, 118.359375
    rows = ("N",
            f"C 1 {r_nc}",
            f"H 2 {r_ch}  1 {a_hcn}",
            f"H 2 {r_ch}  1 {a_hcn} 3 180",
            f"H 1 {r_nh}  2 {alpha} 3 {phi}")
    pad = " " * 20
    return "\n" + "".join(pad + r + "\n" for r in rows) + pad
