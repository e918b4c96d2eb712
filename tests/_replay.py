"""Replays of the reference's recorded notebook runs (tests/golden/notebook_runs.json) through a
backend: the CPU oracle (tests/test_notebook_replay.py) or the HIP engine
(tests/test_notebook_replay_gpu.py).  The drivers below restate the notebooks' loops
(examples/Tutorial_auto_oo.ipynb cell 52 = OO_pqc.full_optimization, oo_pqc.py:155-207;
examples/Tutorial_Berry_phase.ipynb cells 10-32)."""
import json
import os

import numpy as np
import torch

from auto_oo_amd.gaussian import Moldata_sto3g
from auto_oo_amd.moldata import get_formal_geo
from auto_oo_amd.oo_energy import mo_ao_to_mo_oao

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "notebook_runs.json")) as fh:
    RUNS = json.load(fh)
with open(os.path.join(HERE, "golden", "molecule_cases.json")) as fh:
    _CASES = json.load(fh)

_MOLS = {}


def sto3g_molecule(alpha, phi):
    key = (float(alpha), float(phi))
    if key not in _MOLS:
        _MOLS[key] = Moldata_sto3g(get_formal_geo(alpha, phi))
    return _MOLS[key]


def reference_hf_orbitals():
    """S^1/2 C_HF of formaldimine(140, 80)/STO-3G as the reference's PySCF run produced it
    (test/test_oo_energy.py:30-95, 9-digit literal), made exactly orthogonal (polar factor): the
    starting orbitals of the Tutorial_auto_oo run up to the literal's rounding."""
    lit = np.array([c for c in _CASES if c["test"] == "test_mo_ao_to_oao"][0]["hf_oao_coeff_ref"])
    u, _, vt = np.linalg.svd(lit)
    return u @ vt


def newton_trajectory(oo, theta, newton, max_iterations, conv_tol):
    """oo_pqc.py:172-205 on any backend: -> list of energies after each iteration."""
    energies = []
    for n in range(max_iterations):
        kappa = torch.zeros(oo.n_kappa, dtype=torch.float64, device=theta.device)
        new, _ = newton.damped_newton_step(oo.energy_from_parameters, (theta, kappa),
                                           oo.full_gradient(theta), oo.full_hessian(theta))
        theta, kappa = new
        oo.oao_mo_coeff = oo.oao_mo_coeff @ oo.kappa_to_mo_coeff(kappa)
        energies.append(oo.energy_from_parameters(theta).item())
        if n > 1 and abs(energies[-1] - energies[-2]) < conv_tol:
            break
    return energies, theta


def loop_points(run):
    """Tutorial_Berry_phase cells 8-11."""
    phase = np.pi / run["phase_pi_over"]
    pts = []
    for t in [run["t0"] + dt for dt in np.linspace(0, 1, run["n_points"])]:
        pts.append((run["origin"][0] + run["radius"][0] * np.cos(2 * np.pi * t + phase),
                    run["origin"][1] + run["radius"][1] * np.sin(2 * np.pi * t + phase)))
    return pts


def berry_loop(make_oo, newton, run, device):
    """Tutorial_Berry_phase cells 17-22: pre-optimise at point 0 (full_optimization from theta = 0
    and the RHF orbitals), then ONE damped Newton step per loop point, each starting from the
    previous point's (theta, orbitals).  -> dict(preopt, energies, thetas, orbitals, lowest)."""
    pts = loop_points(run)
    mol0 = sto3g_molecule(*pts[0])
    mol0.run_rhf()
    oo = make_oo(mol0, mo_ao_to_mo_oao(mol0.hf.mo_coeff, mol0.overlap))
    theta = torch.zeros(oo.pqc.theta_shape, dtype=torch.float64, device=device)
    e0 = oo.energy_from_parameters(theta).item()
    pre, theta = newton_trajectory(oo, theta, newton, 50, 1e-10)
    H = oo.full_hessian(theta)
    lowest = torch.linalg.eigvalsh(H.cpu())[0].item()
    thetas, orbitals, energies = [theta], [oo.oao_mo_coeff], []
    for step in range(1, len(pts)):
        oo = make_oo(sto3g_molecule(*pts[step]), orbitals[-1])
        kappa = torch.zeros(oo.n_kappa, dtype=torch.float64, device=device)
        new, _ = newton.damped_newton_step(oo.energy_from_parameters, (thetas[-1], kappa),
                                           oo.full_gradient(thetas[-1]), oo.full_hessian(thetas[-1]))
        coeff = orbitals[-1] @ oo.kappa_to_mo_coeff(new[1])
        oo.oao_mo_coeff = coeff
        thetas.append(new[0])
        orbitals.append(coeff)
        energies.append(oo.energy_from_parameters(new[0]).item())
    return dict(preopt=[e0] + pre, energies=energies, thetas=thetas, orbitals=orbitals, lowest=lowest,
                act_idx=list(oo.act_idx))

