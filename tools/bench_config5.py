#!/usr/bin/env python3
"""BASELINE.json configs[4]: kUpCCD CAS(8e,8o) (16-qubit register, 4 900-determinant sector),
synthetic N=43-shaped integrals: gate-apply rate, state+RDM+theta-gradient evaluations/s over a
batch-size sweep, and the full OO evaluation (E + full gradient)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import auto_oo_amd as aoo                       # noqa: E402
from auto_oo_amd.synthetic import synthetic_problem   # noqa: E402


def timed(fn, warm=3, reps=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    out = {}
    ncas, nelecas, nelec, N = 8, 8, 16, 43
    D = 1 << 16
    for k in (1, 2):
        pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="kupccd", k=k)
        eng = pqc._sector
        n_theta = pqc.theta_shape
        rng = np.random.default_rng(5 + k)
        c1 = torch.tensor(rng.standard_normal((ncas, ncas)), device="cuda")
        c2 = torch.tensor(rng.standard_normal((ncas,) * 4), device="cuda")
        rec = {"n_theta": n_theta, "n_gates": pqc._n_gates, "sector_dim": eng.Dc, "sweep": []}
        for batch in (1, 4, 16, 64, 256, 1024):
            th = torch.tensor(rng.uniform(0, 2 * np.pi, (batch, n_theta)), device="cuda")
            t_state = timed(lambda: eng.state(th))

            def full():
                psi_c = eng.state(th)
                eng.rdms(psi_c)
                return eng.adjoint(th, psi_c, c1, c2)
            t_full = timed(full, warm=2, reps=5)
            dense_bytes = batch * pqc._n_gates * 2.0 * D * 16        # SURVEY 8(d) definition
            sector_bytes = batch * pqc._n_gates * 2.0 * eng.Dc * 8   # what the LDS kernel touches
            rec["sweep"].append({
                "batch": batch, "state_us": t_state * 1e6, "states_per_s": batch / t_state,
                "gate_apply_GBs_dense_complex128_equiv": dense_bytes / t_state / 1e9,
                "gate_apply_GBs_sector_lds": sector_bytes / t_state / 1e9,
                "state_rdm_grad_us": t_full * 1e6, "state_rdm_grad_evals_per_s": batch / t_full})
            print(k, rec["sweep"][-1], flush=True)
        # full OO evaluation on one geometry (CAS path + adjoint)
        P = synthetic_problem(N, 20265)
        mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
        oo = aoo.OO_pqc(pqc, mol, ncas, nelecas, oao_mo_coeff=P["oao_mo_coeff"])
        th1 = torch.tensor(rng.uniform(0, 2 * np.pi, n_theta), device="cuda")
        t_oo = timed(lambda: oo.energy_and_gradient(th1), warm=3, reps=20)
        E, g = oo.energy_and_gradient(th1)
        rec["oo_eval_us"] = t_oo * 1e6
        rec["oo_eval_n_kappa"] = oo.n_kappa
        rec["energy"] = E.item()
        # parity spot check of the reverse-mode gradient: central finite difference on theta_0
        h = 1e-5
        j = int(torch.argmax(g[:n_theta].abs()).item())
        tp, tm = th1.clone(), th1.clone()
        tp[j] += h
        tm[j] -= h
        fd = (oo.energy_from_parameters(tp).item() - oo.energy_from_parameters(tm).item()) / (2 * h)
        rec["fd_check_dtheta0"] = {"index": j, "adjoint": g[j].item(), "finite_difference": fd}
        print(k, "oo_eval_us", rec["oo_eval_us"], rec["fd_check_dtheta0"], flush=True)
        out[f"kUpCCD_k{k}"] = rec
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/config5.json", "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
