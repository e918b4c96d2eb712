"""Time every public cost-function call of the drop-in API on the configs[1] shape (N = 43, CAS(4e,3o)), for the UCCD and
the GateFabric ansatz -- a survey for calls that are much slower than their work."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import auto_oo_amd as aoo
from auto_oo_amd.synthetic import synthetic_problem
from torch.autograd.functional import jacobian, hessian

P = synthetic_problem(43, 20262)
mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 16)


def T(f, n=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for ansatz, kw in (("ucc", {}), ("ucc", {"add_singles": True}), ("np_fabric", {"n_layers": 2})):
    pqc = aoo.Parameterized_circuit(3, 4, None, ansatz=ansatz, **kw)
    oo = aoo.OO_pqc(pqc, mol, 3, 4, oao_mo_coeff=P["oao_mo_coeff"])
    rng = np.random.default_rng(1)
    th = torch.tensor(rng.uniform(0, 2 * np.pi, pqc.theta_shape), device="cuda")
    kap = torch.tensor(rng.standard_normal(oo.n_kappa) * 0.05, device="cuda")
    g1, g2 = pqc.get_rdms(th)
    print(f"--- {ansatz} {kw}: n_theta = {int(np.prod(pqc.theta_shape))}, n_kappa = {oo.n_kappa}")
    calls = [("qnode", lambda: pqc.qnode(th)), ("get_rdms", lambda: pqc.get_rdms(th)),
             ("energy_from_parameters(theta)", lambda: oo.energy_from_parameters(th)),
             ("energy_from_parameters(theta, kappa)", lambda: oo.energy_from_parameters(th, kap)),
             ("circuit_gradient", lambda: oo.circuit_gradient(th)), ("orbital_gradient", lambda: oo.orbital_gradient(th)),
             ("full_gradient", lambda: oo.full_gradient(th)),
             ("circuit_circuit_hessian", lambda: oo.circuit_circuit_hessian(th)),
             ("orbital_circuit_hessian", lambda: oo.orbital_circuit_hessian(th)),
             ("orbital_orbital_hessian", lambda: oo.orbital_orbital_hessian(th)),
             ("full_hessian", lambda: oo.full_hessian(th)),
             ("analytic_gradient(rdms)", lambda: oo.analytic_gradient(g1, g2)),
             ("analytic_hessian(rdms) [N^4]", lambda: oo.analytic_hessian(g1, g2)),
             ("energy_from_kappa", lambda: oo.energy_from_kappa(kap, g1, g2)),
             ("get_active_integrals", lambda: oo.get_active_integrals(oo.mo_coeff)),
             ("int2e_transform (full N^5)", lambda: aoo.int2e_transform(oo.int2e_ao, oo.mo_coeff)),
             ("jacobian(energy_from_parameters)(theta, kappa)",
              lambda: jacobian(oo.energy_from_parameters, (th, kap))),
             ]
    for name, f in calls:
        try:
            print(f"{name:52s} {T(f):10.1f} us", flush=True)
        except Exception as exc:          # noqa: BLE001
            print(f"{name:52s} failed: {type(exc).__name__}: {str(exc)[:80]}", flush=True)
    if ansatz == "ucc" and not kw:
        t0 = time.perf_counter()
        H = hessian(oo.energy_from_parameters, (th, kap))
        torch.cuda.synchronize()
        print(f"{'hessian(energy_from_parameters)(theta, kappa) once':52s} {(time.perf_counter() - t0) * 1e6:10.1f} us")
        t0 = time.perf_counter()
        e = oo.orbital_optimization(g1, g2, max_iterations=3, verbose=None)
        torch.cuda.synchronize()
        print(f"{'orbital_optimization, 3 iterations':52s} {(time.perf_counter() - t0) * 1e6:10.1f} us")
