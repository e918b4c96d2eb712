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
