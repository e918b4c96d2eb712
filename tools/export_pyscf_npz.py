"""Export AO integrals with PySCF (on a machine that has it) for ``auto_oo_amd.Moldata.from_npz``.

    python tools/export_pyscf_npz.py "N 0 0 0; C 0 0 1.27; H 0 0.94 -0.54; H 0.89 0 1.83; H -0.89 0 1.83" cc-pvdz out.npz

Writes the quantities the reference's Moldata_pyscf builds (src/auto_oo/moldata_pyscf.py:28-35):
int1e_ao = int1e_kin + int1e_nuc, int2e_ao = int2e (full tensor), overlap = int1e_ovlp,
nuc = energy_nuc(), nelectron, and the RHF orbitals as mo_coeff.  Not runnable in the build image
(PySCF is not installed there); kept as the documented producer of the .npz format.
"""
import sys

import numpy as np


def main(atom, basis, out):
    from pyscf import gto, scf   # third-party, only needed by this exporter
    mol = gto.M(atom=atom, basis=basis, verbose=0)
    mf = scf.RHF(mol).run()
    np.savez(out,
             int1e_ao=mol.intor("int1e_kin") + mol.intor("int1e_nuc"),
             int2e_ao=mol.intor("int2e"),
             overlap=mol.intor("int1e_ovlp"),
             nuc=np.float64(mol.energy_nuc()),
             nelectron=np.int64(mol.nelectron),
             mo_coeff=mf.mo_coeff)
    print(f"wrote {out}: nao = {mol.nao}, nelectron = {mol.nelectron}, E(RHF) = {mf.e_tot:.12f}")


if __name__ == "__main__":
    if len(sys.argv) != 4:
        raise SystemExit(__doc__)
    main(*sys.argv[1:])
