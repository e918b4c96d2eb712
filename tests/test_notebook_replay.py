"""The reference's recorded notebook runs replayed with the CPU ORACLE (oracle/cpu_ref.py) on
integrals from auto_oo_amd/gaussian.py: the published numbers pin the oracle end to end --
GateFabric circuit, RDMs, integral transforms, CAS coefficients, composite gradient, all three
Hessian blocks, the damped Newton step with augmentation and line search, the Berry-phase overlaps.

  * examples/Tutorial_auto_oo.ipynb (cell 52): 20 energies of OO_pqc.full_optimization on
    formaldimine(140, 80) / STO-3G, CAS(4e,3o), np_fabric 2 layers -> -92.74995368139427;
  * examples/Tutorial_Berry_phase.ipynb (cells 17-32): pre-optimisation, nine single Newton steps
    around the loop, ten state overlaps (the last one -0.9987993361086737: the Berry phase)."""
import numpy as np
import torch

from oracle import cpu_ref as R
from tests import _replay as P

RUNS = P.RUNS


def _oracle_oo(run):
    pqc = R.OraclePQC(run["ncas"], run["nelecas"], run["ansatz"], n_layers=run["n_layers"])

    def make(mol, oao_mo_coeff):
        omol = R.OracleMol(mol.int1e_ao, mol.int2e_ao, mol.overlap, mol.nuc, mol.nelectron)
        return R.OracleOOPQC(pqc, omol, run["ncas"], run["nelecas"], oao_mo_coeff,
                             freeze_active=run["freeze_active"])
    return pqc, make


def test_tutorial_oo_vqe_trajectory_with_the_oracle():
    run = RUNS["tutorial_auto_oo"]
    pqc, make = _oracle_oo(run)
    oo = make(P.sto3g_molecule(*run["formal_geo"]), P.reference_hf_orbitals())
    theta = torch.zeros(pqc.theta_shape, dtype=torch.float64)
    ref = run["energies"]
    assert abs(oo.energy_from_parameters(theta).item() - ref[0]) < 1e-9         # iter = 000: RHF
    energies, _ = P.newton_trajectory(oo, theta, R.OracleNewtonStep(), 50, 1e-10)
    assert len(energies) == len(ref) - 1                                        # stops at the same iteration
    # every iterate of the recorded run (the start orbitals are the reference's 9-digit literal:
    # mid-trajectory iterates feel that rounding at the 1e-7 level, the minimum does not)
    assert np.abs(np.array(energies) - np.array(ref[1:])).max() < 5e-7
    assert np.abs(np.array(energies[-2:]) - np.array(ref[-2:])).max() < 1e-9
    assert abs(energies[-1] - run["E_fin"]) < 1e-9
    assert abs(energies[-1] - run["printed_hf_casci_casscf"][2]) < 1e-6         # == CASSCF(4e,3o)


def test_tutorial_berry_phase_loop_with_the_oracle():
    run = RUNS["tutorial_berry_phase"]
    pqc, make = _oracle_oo(run)
    out = P.berry_loop(make, R.OracleNewtonStep(), run, torch.device("cpu"))
    # pre-optimisation: same converged minimum (the path to it starts from our RHF orbitals,
    # converged tighter than the notebook's PySCF run, so early iterates differ in the 6th digit)
    assert abs(out["preopt"][0] - run["preopt_energies"][0]) < 1e-9
    assert abs(out["preopt"][-1] - run["preopt_E_fin"]) < 1e-9
    assert abs(out["preopt"][-1] - run["preopt_casscf_energy"]) < 1e-8
    assert abs(out["lowest"] - run["preopt_lowest_hessian_eigenvalue"]) < 1e-6
    # one damped Newton step per loop point
    assert np.abs(np.array(out["energies"]) - np.array(run["loop_energies"])).max() < 1e-8
    # overlaps <psi_{i+1}| G_{i -> i+1} |psi_i> with the dense operator definition
    states = [pqc.qnode(t).real.numpy() for t in out["thetas"]]
    n = len(states)
    ovl = []
    for i in range(n):
        j = (i + 1) % n
        mo_atob = (out["orbitals"][i].T @ out["orbitals"][j]).numpy()
        act = out["act_idx"]
        from auto_oo_amd.berry import givens_orthogonal
        U = givens_orthogonal(mo_atob[np.ix_(act, act)])
        G = R.orbital_rotation_operator(U)
        ovl.append(float(states[j] @ (G @ states[i])))
    assert np.abs(np.array(ovl[:-1]) - np.array(run["overlaps"])).max() < 5e-7
    assert abs(ovl[-1] - run["final_overlap"]) < 5e-7
    assert ovl[-1] < -0.99                                                       # the sign flip
