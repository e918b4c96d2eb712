#!/usr/bin/env python3
"""Contract benchmark: OO-VQE energy + full-gradient evaluations per second on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], SURVEY.md section 8(d) "C2"): formaldimine CAS(4e,3o)/cc-pVDZ
SHAPE on synthetic tensors -- N=43 AOs, 6 doubly occupied, 3 active orbitals, UCCD (4 thetas),
327 non-redundant kappas.  One EVALUATION is E = energy_from_parameters(theta) and the full
gradient (dE/dtheta, dE/dkappa) for one molecular geometry whose AO integrals are already resident
in HBM.  A STEP is one pass of the hot path over the rank's batch: ONE batched call
(OO_pqc_batch: the geometry index is a grid dimension of every kernel) evaluating all
G = --geoms (default 256) synthetic geometries of the rank's shard (geometry g lives on rank
g mod N: the Berry-phase-loop partition of the north star), each reading its own 27 MB integral
tensor.  `--steps K` therefore times exactly K full batched calls = K x G evaluations per GPU;
`value` = n_gpus x K x G / elapsed (evaluations per second), `config.evals_per_step` = G.
Per-GPU work is fixed ("scaling": "weak"); the only collective is one all_gather of the
per-geometry energies/gradients at the end of the timed region.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself (a
torch.distributed.run child process, started before anything touches the GPU); under torchrun the
ranks read RANK / LOCAL_RANK / WORLD_SIZE as usual.

The JSON line also carries
  roofline      -- dominant kernel (the N^4 half-transform sweep) against the HBM roofline,
                   timed with HIP events on the launch stream inside the timed region;
  cpu_baseline  -- the CPU oracle (plain-torch restatement of the reference algorithm, "port")
                   timed on this box's host cores on a bounded sample (rank 0, N=1 only);
  transform     -- the second half of the headline metric: full (pq|rs)->(ij|kl) transform at
                   N=200 (BASELINE.json configs[2]) in fp64 TFLOP/s against the 78.6 TF peak.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = ("OO-VQE energy+grad evals/sec (formaldimine CAS(4e,3o)/cc-pVDZ); "
          "2e-transform fp64 TFLOP/s vs roofline")
NAO, NELEC, NCAS, NELECAS = 43, 16, 3, 4
N_GEOM = 256         # geometries per GPU (7 GB of g_ao + 1.9 GB packed copy in 288 GB of HBM; --geoms 1024:
                     # 36 GB, +2...8 % evaluations/s depending on the box)
N_GEOM_HOST = 4      # of them generated with numpy on the host (geometry 0 is the parity / CPU-baseline anchor)
N_GEOM_BERRY = 64    # geometries per GPU of the Berry-loop extra (each holds its own OO_pqc object)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # AMD spec, vector = matrix fp64


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400,
                    help="timed steps; one step = one batched call over the rank's --geoms geometries")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--geoms", type=int, default=N_GEOM,
                    help="molecular geometries PER GPU (weak scaling: the job holds geoms x n_gpus)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--general-eri", action="store_true",
                    help="treat g_ao as a general tensor (eri_flags = 0: every slab is read) in the "
                         "headline run, for comparison with the default symmetric-integral path")
    ap.add_argument("--deferred", action="store_true",
                    help="headline steps as deferred calls over the library's two side streams "
                         "(OO_pqc_batch.evaluate_deferred) instead of in-order calls on the current stream: the tail of "
                         "one call under the N^4 sweep of the next.  Measured (round 5): +3 ... 6 % in short bursts, "
                         "nothing in the sustained regime the headline is taken in (0.5 s of priming: the chip is "
                         "power-limited there and the sweep slows by what the tails use) -- so not the default")
    ap.add_argument("--no-transform", action="store_true")
    ap.add_argument("--no-kupccd", action="store_true", help="skip the configs[4] kUpCCD CAS(8e,8o) extra")
    ap.add_argument("--no-berry", action="store_true",
                    help="skip the configs[3] extra (energy+gradient+Hessian+Newton step per geometry)")
    ap.add_argument("--transform-n", type=int, default=200)
    ap.add_argument("--prime-seconds", type=float, default=0.5,
                    help="set-up time spent keeping the GPU busy before the W warm-up steps")
    ap.add_argument("--berry-geoms", type=int, default=N_GEOM_BERRY,
                    help="geometries of the configs[3] Berry-phase-loop extra: this many IN TOTAL for the "
                         "strong-scaling figure (split over the GPUs) and this many PER GPU for the weak one")
    ap.add_argument("--master-port", type=int, default=0,
                    help="rendezvous port when bench.py starts the ranks itself (0 = pick a free one)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend: nccl (= RCCL, default) or gloo (rehearsal of the "
                         "N>1 path on a one-GPU box: all ranks share cuda:0)")
    return ap.parse_args()


def build_geometries(my_geoms):
    """The rank's batch of synthetic geometries.  The first N_GEOM_HOST come from the numpy generator
    (the seeds the CPU oracle uses: geometry 0 is the parity anchor); the others are generated on
    the GPU with the same construction (auto_oo_amd.synthetic.synthetic_problem_device) and written
    straight into the batch's stacked tensors -- every geometry has its own distinct 27 MB of
    integrals in HBM, none of it ever on the host."""
    import auto_oo_amd as aoo
    from auto_oo_amd.synthetic import synthetic_problem, synthetic_problem_device
    pqc = aoo.Parameterized_circuit(NCAS, NELECAS, None, ansatz="ucc")
    n_host = min(N_GEOM_HOST, len(my_geoms))
    mols, coeffs, thetas = [], [], []
    for g in my_geoms[:n_host]:
        P = synthetic_problem(NAO, 20260 + 2 + 1000 * g)
        mols.append(aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], NELEC))
        coeffs.append(P["oao_mo_coeff"])
    for g in my_geoms:
        rng = np.random.default_rng(777 + g)
        thetas.append(rng.uniform(0, 2 * np.pi, pqc.theta_shape))
    pad = len(my_geoms) - n_host                     # slots filled on the device below
    batch = aoo.OO_pqc_batch(pqc, mols + [mols[0]] * pad, NCAS, NELECAS, oao_mo_coeffs=coeffs + [coeffs[0]] * pad)
    for slot in range(n_host, len(my_geoms)):
        P = synthetic_problem_device(NAO, 20260 + 2 + 1000 * my_geoms[slot], "cuda")
        batch.int2e_ao[slot].copy_(P["int2e_ao"])
        batch.int1e_ao[slot].copy_(P["int1e_ao"])
        batch.oao_coeff[slot].copy_(P["oao_coeff"])
        batch.oao_mo_coeff[slot].copy_(P["oao_mo_coeff"])
        batch.nuc[slot] = P["nuc"]
    if pad:
        batch.reverify_integrals()
    single = aoo.OO_pqc(pqc, mols[0], NCAS, NELECAS, oao_mo_coeff=coeffs[0])
    thetas = torch.tensor(np.stack(thetas), device="cuda")
    return pqc, batch, single, thetas


def cpu_baseline(seconds_budget=24.0):
    """Reference algorithm (3 simulations + 3 full N^5 transforms + autograd jacobian per
    evaluation, oo_pqc.py:64-101,132-134) restated in plain torch, on the host cores: SURVEY.md
    section 8(d) -- all host threads (os.cpu_count()) and one thread, warm-up 3, median of >= 20
    evaluations (fewer only if the time budget runs out; the sample says how many)."""
    from oracle import cpu_ref as R
    P = R.synthetic_problem(NAO, 20260 + 2)
    mol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], NELEC)
    pqc = R.OraclePQC(NCAS, NELECAS, "ucc")
    oo = R.OracleOOPQC(pqc, mol, NCAS, NELECAS, P["oao_mo_coeff"])
    theta = torch.tensor(np.random.default_rng(777).uniform(0, 2 * np.pi, pqc.theta_shape))

    def one():
        e = oo.energy_from_parameters(theta)
        g = oo.full_gradient(theta)
        return e, g

    def median_rate(threads, budget):
        torch.set_num_threads(threads)
        for _ in range(3):
            one()
        ts = []
        t_start = time.perf_counter()
        while len(ts) < 20 or (time.perf_counter() - t_start < budget and len(ts) < 60):
            t0 = time.perf_counter()
            one()
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget and len(ts) >= 5:
                break
        ts.sort()
        return 1.0 / ts[len(ts) // 2], len(ts)

    n_threads_before = torch.get_num_threads()
    # SURVEY.md section 8(d) asks for os.cpu_count() threads; a 1-GPU box shows all 256 host CPUs but
    # owns a 16-core share of them, and at 256 threads the many tiny torch ops of the reference
    # algorithm collapse under oversubscription (measured: 0.058 evaluations/s against 4.4 on ONE
    # thread, profiles/r02_a_bench_driver_cmd.json) -- so "all threads" is capped at that share
    cores = min(os.cpu_count() or 1, 16)
    v_all, n_all = median_rate(cores, seconds_budget / 2)
    v_one, n_one = median_rate(1, seconds_budget / 2)
    torch.set_num_threads(n_threads_before)
    e, g = one()
    return dict(value=v_all, unit="evals/s", cores=cores, kind="port",
                one_thread_value=v_one,
                sample=f"median of {n_all} ({cores} threads = this box's CPU share) / {n_one} (1 thread) energy+full-"
                       f"gradient evaluations of geometry 0 (N={NAO}) after 3 warm-up evaluations, "
                       f"torch {torch.__version__} CPU"), float(e), g


def berry_loop_extra(my_geoms, n_geom, dist, world, backend="nccl", mode="weak", regime="cold"):
    """BASELINE.json configs[3]: for every geometry of this rank's shard, from a shared (theta0, C0):
    energy + full gradient + full (n_theta+n_kappa)^2 Hessian + one damped Newton step
    (oo_pqc.py:172-196); one all_gather of the new energies at the end.  Returns geometries/s.

    regime "cold": independent random geometries and an arbitrary shared start (theta = 0.1, random
    orthogonal orbitals) -- far from any minimum, the Hessians are indefinite and every step is level-shifted
    (newton_raphson.py:107-120).  regime "tracking": what a step of the Berry-phase loop is in the reference's
    notebook (examples/Tutorial_Berry_phase.ipynb raw 404-441: freeze_active = True, ONE Newton step per loop
    point from the previous point's solution): the geometries are small displacements around a base problem
    (synthetic_loop) and the shared start is the converged optimum of the base -- the Hessians are positive
    definite and no step is shifted."""
    import contextlib
    import auto_oo_amd as aoo
    from auto_oo_amd import ops as ops_mod
    from auto_oo_amd.synthetic import synthetic_problem, synthetic_loop
    from auto_oo_amd.parallel import gather_results
    pqc = aoo.Parameterized_circuit(NCAS, NELECAS, None, ansatz="ucc")
    objs = []
    if regime == "tracking":
        freeze = True
        base, loop = synthetic_loop(NAO, 20263, n_geom, eps=0.01)
        bmol = aoo.Moldata(base["int1e_ao"], base["int2e_ao"], base["overlap"], base["nuc"], NELEC)
        boo = aoo.OO_pqc(pqc, bmol, NCAS, NELECAS, oao_mo_coeff=base["oao_mo_coeff"], freeze_active=True)
        th_start = torch.full((pqc.theta_shape,), 0.1, dtype=torch.float64, device="cuda")
        c_start = boo.oao_mo_coeff.detach().clone()
        per_iter = {}
        # OO_pqc.full_optimization (oo_pqc.py:155-207) of the base problem, timed both ways (second run of each):
        # the reference's sequence of calls per iteration, and one library call per iteration (the default)
        for by_calls in (True, False):
            boo.optimization_by_calls = by_calls
            for rep in range(2):
                boo.oao_mo_coeff = c_start.clone()
                torch.cuda.synchronize()
                t_opt = time.perf_counter()
                with contextlib.redirect_stdout(sys.stderr):
                    e_l, th_l, _, _, eig_l = boo.full_optimization(th_start, max_iterations=80, conv_tol=1e-11,
                                                                   verbose=None)
                torch.cuda.synchronize()
                t_opt = time.perf_counter() - t_opt
            per_iter["by_calls" if by_calls else "one_call"] = t_opt / len(e_l) * 1e3
        theta0 = th_l[-1].detach().clone()
        c_star = boo.oao_mo_coeff.detach().clone()
        setup = {"base_optimisation_iterations": len(e_l), "base_energy": float(e_l[-1]),
                 "base_lowest_hessian_eigenvalue": float(eig_l[-1]), "displacement_eps": 0.01,
                 "freeze_active": True, "full_optimization_ms_per_iteration": per_iter}
        for g in my_geoms:
            P = loop[g]
            mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], NELEC)
            objs.append(aoo.OO_pqc(pqc, mol, NCAS, NELECAS, oao_mo_coeff=c_star, freeze_active=True))
            objs[-1]._mol_ref = mol
        del boo, bmol, base, loop
    else:
        freeze = False
        setup = {"freeze_active": False}
        for g in my_geoms:
            P = synthetic_problem(NAO, 20260 + 2 + 1000 * g)
            mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], NELEC)
            objs.append(aoo.OO_pqc(pqc, mol, NCAS, NELECAS, oao_mo_coeff=P["oao_mo_coeff"]))
            objs[-1]._mol_ref = mol
        theta0 = torch.full((pqc.theta_shape,), 0.1, dtype=torch.float64, device="cuda")
    opt = aoo.NewtonStep(verbose=0)

    eigs = []

    def one(oo):
        kappa = torch.zeros(oo.n_kappa, dtype=torch.float64, device="cuda")
        grad = oo.full_gradient(theta0)
        hess = oo.full_hessian(theta0)
        # the lowest Hessian eigenvalue (hess_eig_l of the reference's loops) is collected, as OO_pqc.full_optimization
        # collects it: joined once, before the gather -- a geometry's step does not wait for the band route that
        # computes it beside the line search when the Hessian is positive definite
        new, eig = opt.damped_newton_step(oo.energy_from_parameters, (theta0, kappa), grad, hess, defer_lowest=True)
        eigs.append(eig)
        return oo.energy_from_parameters(new[0], new[1])

    # set-up, untimed: every geometry's object verifies its integrals' symmetry flags (one pass + a readback) and
    # builds its evaluation plan on first use -- the loop body proper is what is timed
    for oo in objs:
        oo.full_gradient(theta0)
    one(objs[0])                                   # warm-up
    eigs.clear()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    res = torch.stack([one(oo) for oo in objs]).reshape(-1, 1)
    eig_host = [float(e) for e in eigs]            # (joins the side streams)
    full = gather_results(res, my_geoms, n_geom, dist)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0

    # the same steps for all geometries of the shard in lockstep on the batched device path
    # (OO_pqc_batch.damped_newton_step): ONE library call for energy + gradient + full Hessian of all
    # geometries, one launch for all Newton directions, every line-search trial one batched evaluation,
    # one launch to rotate all orbitals
    batch = aoo.OO_pqc_batch(pqc, [oo._mol_ref for oo in objs], NCAS, NELECAS,
                             oao_mo_coeffs=[oo.oao_mo_coeff for oo in objs], freeze_active=freeze)
    thetas0 = theta0.reshape(1, -1).repeat(len(objs), 1).contiguous()
    c_saved = batch.oao_mo_coeff.clone()
    bopt = aoo.BatchedNewtonStep(verbose=0)

    def restore():
        batch.oao_mo_coeff.copy_(c_saved)
        batch.refresh_mo_coeff()

    def lockstep():
        # (the lowest Hessian eigenvalues are joined: everything the step returns is complete when it is timed)
        return batch.damped_newton_step(thetas0, bopt)[1]

    t_w = time.perf_counter()                      # warm-up (workspaces, code objects, clocks)
    while time.perf_counter() - t_w < 0.1:
        lockstep()
        restore()
    reps = 5
    times = []
    res_b = None
    for _ in range(reps):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t1 = time.perf_counter()
        res_b = lockstep().reshape(-1, 1)
        gather_results(res_b, my_geoms, n_geom, dist)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        times.append(time.perf_counter() - t1)
        restore()
    el_b = sorted(times)[len(times) // 2]
    agree = float((res_b - res).abs().max().item())
    # ingest of the shard's geometries (untimed above: SURVEY.md section 8(d) fixes g_ao): in the reference's loop every
    # point is a NEW molecule (Tutorial_Berry_phase.ipynb raw 408-418), so its integrals are verified and packed
    # once per step -- one pass over the stack (oovqe_eri_ingest); per object of the sequential path: the symmetry
    # test alone (single geometries keep no packed copy at this size).  Host -> device copies are not in it.
    t_ing = []
    for _ in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        batch.reverify_integrals()
        torch.cuda.synchronize()
        t_ing.append(time.perf_counter() - t1)
    ing_b = min(t_ing)
    restore()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for oo in objs:
        ops_mod.eri_flags(oo.int2e_ao)
    torch.cuda.synchronize()
    ing_seq = time.perf_counter() - t1
    # where the step's time goes (untimed extra pass): gradient + Hessian call, direction launch
    def timed(fn, n_rep=5):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n_rep):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n_rep * 1e6
    egh_us = timed(lambda: batch.energy_gradient_hessian(thetas0))
    _, g_b, h_b = batch.energy_gradient_hessian(thetas0)
    dir_us = timed(lambda: bopt.newton_steps(g_b, h_b))
    from auto_oo_amd import ops as _ops
    _, low_b, _, info_b = _ops.newton_direction(h_b, g_b, want_info=True)
    fast_fraction = float((info_b == 1.0).double().mean().item())
    low_range = [float(low_b.min().item()), float(low_b.max().item())]
    trial_us = timed(lambda: batch.energy(thetas0, g_b[:, batch.n_theta:] * 1e-3))
    if dist is not None:
        tmax = torch.tensor([el, el_b], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el, el_b = float(tmax[0].item()), float(tmax[1].item())
    return {"regime": regime, "setup": setup,
            "cholesky_fast_path_fraction": fast_fraction, "lowest_hessian_eigenvalue_range": low_range,
            "geometries": n_geom, "seconds": el, "geometries_per_s": n_geom / el,
            "per_geometry_ms": el / max(len(objs), 1) * 1e3, "hessian_dim": objs[0].n_kappa + pqc.theta_shape,
            "including_ingest": {"ingest_seconds": ing_seq, "seconds": el + ing_seq,
                                 "geometries_per_s": n_geom / (el + ing_seq),
                                 "note": "+ the bitwise symmetry test of every geometry's integrals (one pass each, "
                                         "a host readback each: what OO_pqc does on first use of a new molecule); "
                                         "this rank's share, host->device copies excluded"},
            "scaling": (f"{mode} ({n_geom} geometries in the job, {len(objs)} on this GPU, geometry g on "
                        f"rank g mod n_gpus)"),
            "lockstep": {"seconds": el_b, "geometries_per_s": n_geom / el_b,
                         "per_geometry_ms": el_b / max(len(objs), 1) * 1e3,
                         "step_ms_all_reps": [t * 1e3 for t in times],
                         "including_ingest": {"ingest_ms": ing_b * 1e3, "seconds": el_b + ing_b,
                                              "geometries_per_s": n_geom / (el_b + ing_b),
                                              "note": "+ OO_pqc_batch.reverify_integrals(): symmetry tests + packed copy "
                                                      "of the shard's geometries in ONE pass over the stack, flags read "
                                                      "back; host->device copies excluded"},
                         "max_abs_energy_difference_vs_sequential": agree,
                         "energy_gradient_hessian_call_us": egh_us,
                         "direction_launch_us": dir_us,
                         "line_search_trial_us": trial_us,
                         "note": "the shard's geometries stepped together on the batched device path "
                                 "(OO_pqc_batch.damped_newton_step): one oovqe_oo_hessian_batch call, one "
                                 "oovqe_newton_direction launch, batched line-search trials, one host sync "
                                 "per trial; median of 5 steps"},
            "newton_direction_us": newton_direction_timing(g_b, h_b),
            "strong_projection": (strong_projection(pqc, objs, theta0, freeze)
                                  if (mode == "strong" and world == 1) else None),
            "mean_energy_after_step": float(full.mean().item())}


def strong_projection(pqc, objs, theta0, freeze, sizes=(64, 32, 16, 8)):
    """The strong-scaling curve of the 64-geometry Berry-phase step as ONE GPU can measure it: the path has no
    data-path collective, so the work of a rank at N GPUs is exactly a lockstep step over 64 / N geometries.
    Per size: median time until the step's outputs the loop consumes (new thetas, orbitals, energies) are
    complete (`step_ms`: the lowest Hessian eigenvalues, a diagnostic nothing downstream reads, still running
    on the side stream), the same with the eigenvalues joined (`step_with_lowest_eig_ms`), and K back-to-back
    steps with one join at the end (`pipelined_step_ms`: how a loop of steps runs)."""
    import auto_oo_amd as aoo
    out = {}
    bopt = aoo.BatchedNewtonStep(verbose=0)
    for G in sizes:
        if G > len(objs):
            continue
        sub = objs[:G]
        batch = aoo.OO_pqc_batch(pqc, [oo._mol_ref for oo in sub], NCAS, NELECAS,
                                 oao_mo_coeffs=[oo.oao_mo_coeff for oo in sub], freeze_active=freeze)
        thetas0 = theta0.reshape(1, -1).repeat(G, 1).contiguous()
        c_saved = batch.oao_mo_coeff.clone()

        def restore():
            batch.oao_mo_coeff.copy_(c_saved)
            batch.refresh_mo_coeff()

        # warm-up: workspaces, code objects -- and the clocks (a GPU that has just idled through the set-up of the stack
        # runs latency-bound launches slower for its first ~10 ms)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.1:
            batch.damped_newton_step(thetas0, bopt, defer_lowest=True)[2].result()
            restore()
        torch.cuda.synchronize()
        crit, joined = [], []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th, en, pend = batch.damped_newton_step(thetas0, bopt, defer_lowest=True)
            done = torch.cuda.Event()
            done.record()
            done.synchronize()
            t1 = time.perf_counter()
            pend.result()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            crit.append(t1 - t0)
            joined.append(t2 - t0)
            restore()
        k_steps = 6
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th, pending = thetas0, []
        for _ in range(k_steps):
            th, en, pend = batch.damped_newton_step(th, bopt, defer_lowest=True)
            pending.append(pend)
        for pend in pending:
            pend.result()
        torch.cuda.synchronize()
        pipe = (time.perf_counter() - t0) / k_steps
        restore()
        out[str(G)] = {"step_ms": sorted(crit)[len(crit) // 2] * 1e3,
                       "step_with_lowest_eig_ms": sorted(joined)[len(joined) // 2] * 1e3,
                       "pipelined_step_ms": pipe * 1e3}
        del batch
    base = out.get(str(sizes[0]))
    if base:
        for G in sizes[1:]:
            if str(G) in out:
                o = out[str(G)]
                o["implied_gpus"] = sizes[0] // G
                o["implied_speedup"] = base["step_ms"] / o["step_ms"]
                o["implied_speedup_with_lowest_eig"] = base["step_with_lowest_eig_ms"] / o["step_with_lowest_eig_ms"]
                o["implied_speedup_pipelined"] = base["pipelined_step_ms"] / o["pipelined_step_ms"]
    out["note"] = ("lockstep damped Newton step (gradient + full Hessian of all geometries in one call, Cholesky "
                   "directions, batched line search, orbital rotation, final energies) over 64 / N geometries on "
                   "this one GPU = the per-rank work at N GPUs (no collective inside a step); implied_speedup = "
                   "t(64) / t(64 / N)")
    return out


def newton_direction_timing(grads, hess):
    """The direction kernel alone (newton.hip) on the shard's real Hessians: one problem per launch
    and all of them in one launch."""
    from auto_oo_amd import ops

    def timed(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    return {"n": int(grads.shape[1]), "single": timed(lambda: ops.newton_direction(hess[0], grads[0]), 20),
            "batch": int(grads.shape[0]),
            "batched_launch": timed(lambda: ops.newton_direction(hess, grads), 10),
            "single_direction_only": timed(lambda: ops.newton_direction(hess[0], grads[0], defer_lowest=True), 20),
            "batched_direction_only": timed(lambda: ops.newton_direction(hess, grads, defer_lowest=True), 10),
            "note": "direction_only: the Cholesky fast path with the lowest eigenvalues left running on the side "
                    "stream (back-to-back calls: their band route bounds the rate)"}


def kupccd_extra():
    """BASELINE.json configs[4]: kUpCCD CAS(8e,8o), k = 1 and 2 layers (56 / 112 thetas,
    ansatze/kUpCCD.py:16-33), 16-qubit register simulated in its 4900-determinant (N_alpha, N_beta)
    sector: state, state + RDMs + reverse-mode theta-gradient per second, and one full OO evaluation
    (E + full gradient) on a synthetic N=43 geometry; the RDM stage against the fp64 matrix peak (its Gram
    is 2 a^4 Dc flop per state, its HBM traffic a few KB) and the adjoint stage against HBM (W written once,
    read once)."""
    import auto_oo_amd as aoo
    from auto_oo_amd.synthetic import synthetic_problem
    ncas, nelecas = 8, 8
    rng = np.random.default_rng(6)
    c1 = torch.tensor(rng.standard_normal((ncas, ncas)), device="cuda")
    c2 = torch.tensor(rng.standard_normal((ncas,) * 4), device="cuda")

    def timed(fn, warm=3, reps=10):
        # keep the GPU busy for >= 0.2 s first: after host-side set-up the clocks sit at idle and the
        # first milliseconds of latency-bound launches run 2-4x slower
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.2:
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    P = synthetic_problem(NAO, 20265)
    mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], NELEC)
    layers = []
    for k in (1, 2):
        pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="kupccd", k=k)
        eng = pqc._sector
        n_theta = int(pqc.theta_shape)
        rec = {"k": k, "n_theta": n_theta, "n_gates": pqc._n_gates, "sector_dim": eng.Dc, "batches": []}
        for batch in (1, 256):
            th = torch.tensor(rng.uniform(0, 2 * np.pi, (batch, n_theta)), device="cuda")
            t_state = timed(lambda: eng.state(th))

            def full():
                psi_c = eng.state(th)
                eng.rdms(psi_c)
                return eng.adjoint(th, psi_c, c1, c2)
            t_full = timed(full, warm=2, reps=5)
            psi_c = eng.state(th)
            t_rdm = timed(lambda: eng.rdms(psi_c), warm=2, reps=5)
            t_adj = timed(lambda: eng.adjoint(th, psi_c, c1, c2), warm=2, reps=5)
            a2 = ncas * ncas
            gram_flops = 2.0 * a2 * a2 * eng.Dc * batch              # W = Ms^T V: a^2 x a^2 coefficients per determinant
            # Gamma = Gram of the a^2 vectors E_pq psi, symmetric: the 16 x 16 tiles on and above the diagonal
            nt = (a2 + 15) // 16
            gram_issued = 2.0 * (nt * (nt + 1) // 2) * 256 * eng.Dc * batch
            w_bytes = 2.0 * a2 * eng.Dc * 8 * batch                  # W = Ms^T V written once, read once
            dense_bytes = batch * pqc._n_gates * 2.0 * (1 << pqc.n_qubits) * 16.0   # SURVEY.md section 8(d)'s gate-apply figure
            rec["batches"].append({
                "batch": batch, "state_us": t_state * 1e6,
                "gate_apply": {"us_per_gate_per_state": t_state * 1e6 / pqc._n_gates / batch,
                               "gates": pqc._n_gates, "us_per_circuit_per_state": t_state * 1e6 / batch,
                               "dense_complex128_bytes_of_the_reference_simulator": dense_bytes,
                               "dense_equivalent_GBs": dense_bytes / t_state / 1e9,
                               "hbm_bytes_actually_moved": batch * (n_theta + eng.Dc) * 8.0,
                               "note": "SURVEY.md section 8(d) prices a gate at 2 x D x 16 B (a dense complex128 register read and "
                                       "written per FermionicDoubleExcitation: 2 MiB per gate and state at 16 qubits); here the "
                                       "state is the REAL 4 900-determinant (N_alpha, N_beta) sector vector, LDS-resident for the "
                                       "whole circuit (one workgroup per state, every gate a list of determinant pairs): HBM sees "
                                       "theta in and psi out, so the dense-equivalent GB/s is a statement about the work avoided, "
                                       "not about memory traffic; the kernel is latency-bound (one workgroup per state)"},
                "state_rdm_grad_us": t_full * 1e6,
                "state_rdm_grad_evals_per_s": batch / t_full,
                "rdm_stage": {"us": t_rdm * 1e6, "bound": "mfma", "flops": gram_issued,
                              "achieved": gram_issued / t_rdm / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": gram_issued / t_rdm / 1e12 / FP64_PEAK_TFLOPS,
                              "hbm_bytes_algorithmic": batch * (eng.Dc + a2 + a2 * a2) * 8.0,
                              "note": "psi read, gamma / Gamma written; the a^2 vectors E_pq psi live in LDS only; "
                                      "flops = the matrix-core work of the symmetric Gram (tiles mt <= nt)"},
                "adjoint_stage": {"us": t_adj * 1e6, "bound": "mfma", "flops": gram_flops,
                                  "achieved": gram_flops / t_adj / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": gram_flops / t_adj / 1e12 / FP64_PEAK_TFLOPS,
                                  "hbm_bytes_algorithmic": batch * 2.0 * eng.Dc * 8,
                                  "round3_w_bytes_no_longer_moved": w_bytes,
                                  "note": "round 4: lambda = (Hop + Hop^T) psi in the string-driven form -- G_a Psi + Psi "
                                          "G_b^T from two small dense matrices per call, the mixed alpha-beta term formed, "
                                          "multiplied by the a^2 x a^2 coefficients on the matrix cores (flops = 2 a^4 Dc) and "
                                          "gathered inside LDS; W = Ms^T (E psi), 2.5 MB per state written and gathered "
                                          "again in round 3, no longer exists.  The time also holds the reverse sweep "
                                          "through the gates (latency-bound, one workgroup per state)"}})
        oo = aoo.OO_pqc(pqc, mol, ncas, nelecas, oao_mo_coeff=P["oao_mo_coeff"])
        th1 = torch.tensor(rng.uniform(0, 2 * np.pi, n_theta), device="cuda")
        rec["oo_eval_us"] = timed(lambda: oo.energy_and_gradient(th1), warm=3, reps=20) * 1e6
        rec["n_kappa"] = oo.n_kappa
        if k == 1:
            # round 4: second derivatives inside the sector -- the full (n_theta + n_kappa)^2 Hessian of
            # OO_pqc.full_hessian (oo_pqc.py:136-148) and one damped Newton step of full_optimization's body
            rec["full_hessian_us"] = timed(lambda: oo.full_hessian(th1), warm=2, reps=5) * 1e6
            th_s = torch.tensor(rng.normal(0, 0.3, n_theta), device="cuda")
            kap0 = torch.zeros(oo.n_kappa, dtype=torch.float64, device="cuda")
            e_before = oo.energy_from_parameters(th_s).item()
            opt = aoo.NewtonStep(verbose=0)

            def newton_step():
                g = oo.full_gradient(th_s)
                H = oo.full_hessian(th_s)
                return opt.damped_newton_step(oo.energy_from_parameters, (th_s, kap0), g, H)
            rec["newton_step_us"] = timed(newton_step, warm=1, reps=3) * 1e6
            new, low = newton_step()
            rec["newton_step_energy_drop"] = e_before - oo.energy_from_parameters(new[0], new[1]).item()
            rec["hessian_dim"] = n_theta + oo.n_kappa
        layers.append(rec)
    # round 5: configs[4]'s circuit on configs[3]'s loop -- the sector engine under the geometry batch: kUpCCD CAS(8e,8o),
    # k = 1, over a stack of synthetic N = 43 geometries (OO_pqc_batch: states, RDMs, reverse sweeps and the operator
    # of all geometries in one launch sequence each, per-geometry CAS coefficients inside the sector kernels)
    Gb = 16
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="kupccd", k=1)
    probs = [synthetic_problem(NAO, 20265 + 1000 * g) for g in range(Gb)]
    mols = [aoo.Moldata(P_["int1e_ao"], P_["int2e_ao"], P_["overlap"], P_["nuc"], NELEC) for P_ in probs]
    batch = aoo.OO_pqc_batch(pqc, mols, ncas, nelecas, oao_mo_coeffs=[P_["oao_mo_coeff"] for P_ in probs])
    n_theta = int(pqc.theta_shape)
    ths = torch.tensor(rng.normal(0, 0.3, (Gb, n_theta)), device="cuda")
    single = aoo.OO_pqc(pqc, mols[0], ncas, nelecas, oao_mo_coeff=probs[0]["oao_mo_coeff"])
    t_eval = timed(lambda: batch.energy_and_gradient(ths), warm=2, reps=5)
    t_one = timed(lambda: single.energy_and_gradient(ths[0]), warm=2, reps=5)
    t_hess = timed(lambda: batch.energy_gradient_hessian(ths), warm=1, reps=3)
    c_saved = batch.oao_mo_coeff.clone()
    e0 = batch.energy(ths)

    def step():
        batch.oao_mo_coeff.copy_(c_saved)
        batch.refresh_mo_coeff()
        return batch.damped_newton_step(ths)
    t_step = timed(step, warm=1, reps=3)
    _, e1, _ = step()
    batch.oao_mo_coeff.copy_(c_saved)              # (the step adopted new orbitals: back to the start for the comparison)
    batch.refresh_mo_coeff()
    eg_b = batch.energy_and_gradient(ths)
    e_s, g_s = single.energy_and_gradient(ths[0])
    geometry_batch = {"geometries": Gb, "n_theta": n_theta, "n_kappa": batch.n_kappa, "N": NAO,
                      "energy_gradient_us": t_eval * 1e6, "evaluations_per_s": Gb / t_eval,
                      "single_geometry_energy_gradient_us": t_one * 1e6,
                      "energy_gradient_hessian_us": t_hess * 1e6, "lockstep_newton_step_us": t_step * 1e6,
                      "max_abs_difference_vs_single_geometry": float(max(abs(eg_b[0, 0] - e_s), (eg_b[0, 1:] - g_s).abs().max())),
                      "every_energy_lowered_by_the_step": bool((e1 < e0).all().item()),
                      "note": "OO_pqc_batch on a sector circuit: one state / RDM / reverse-sweep launch sequence and one "
                              "oovqe_cas_eval_batch call for all geometries; the Hessian adds the derivative RDMs of all "
                              "geometries by polarisation, one oovqe_orbital_hessian_batch call and the operator on psi and "
                              "its tangents of all geometries (oovqe_sector_lambda_pg)"}
    return {"layers": layers, "geometry_batch": geometry_batch}


def transform_microbench(N):
    """BASELINE.json configs[2]: full 4-index transform + expm at N=200 on synthetic data."""
    from auto_oo_amd import ops
    dev = "cuda"
    g = torch.rand((N, N, N, N), dtype=torch.float64, device=dev) - 0.5
    C = torch.rand((N, N), dtype=torch.float64, device=dev) - 0.5
    o = torch.empty_like(g)
    w = torch.empty_like(g)
    for _ in range(2):
        ops.general_4index_transform(g, C, C, C, C, out=o, work=w)
    torch.cuda.synchronize()
    reps = 5
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(reps)]
    for a, b in evs:
        a.record()
        ops.general_4index_transform(g, C, C, C, C, out=o, work=w)
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e-3 for a, b in evs)
    med = ts[len(ts) // 2]
    tf = 8.0 * N ** 5 / med / 1e12
    K = (C - C.T) * (0.05 / 0.29)     # entries ~ N(0, 0.05^2)-sized rotation generator
    for _ in range(3):
        ops.expm(K, -1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ops.expm(K, -1.0)
    torch.cuda.synchronize()
    ex_us = (time.perf_counter() - t0) / 10 * 1e6
    del o, w
    torch.cuda.empty_cache()
    oo_eval = oo_evaluation_large(N, g, C)
    del g
    torch.cuda.empty_cache()
    return dict(N=N, ms=med * 1e3, tflops=tf, peak_tflops=FP64_PEAK_TFLOPS,
                frac=tf / FP64_PEAK_TFLOPS, flops=8.0 * N ** 5, expm_us=ex_us,
                note="four chained fp64-MFMA mode contractions; algorithmic 8 N^5 flop",
                oo_evaluation=oo_eval)


def oo_evaluation_large(N, g, C):
    """configs[2] as ONE OO evaluation: N = 200, CAS(6e,6o), n_occ = 20 (M = 26) -- beyond the packed
    batched kernels (N <= 48, M <= 16), so the library takes the streaming T2 path: persistent
    half_stream_kernel over the N^4 tensor, K1 for q->x and p->n, Fock stage.  Timed on a general
    tensor (every slab read) and on a p<->q / r<->s symmetric one (slabs p <= q only)."""
    import auto_oo_amd as aoo
    from auto_oo_amd import ops
    ncas, nelecas, n_occ = 6, 6, 20
    M = n_occ + ncas
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="ucc")
    theta = torch.tensor(np.random.default_rng(8).uniform(0, 2 * np.pi, pqc.theta_shape), device="cuda")
    g1, g2 = pqc.get_rdms(theta)
    g1, g2 = g1[None].contiguous(), g2[None].contiguous()
    rows, cols = aoo.excitations.tril_tables(N, aoo.non_redundant_indices(
        np.arange(n_occ), n_occ + np.arange(ncas), np.arange(M, N), False))
    kr, kc = torch.as_tensor(rows).to("cuda"), torch.as_tensor(cols).to("cuda")
    Q, _ = torch.linalg.qr(C)
    Q = Q.contiguous()
    h = (C + C.T).contiguous()
    work = torch.empty(aoo._lib.load().oovqe_cas_eval_work_size(N, n_occ, ncas, 1), dtype=torch.float64,
                       device="cuda")

    def timed(flags, packed=None):
        for _ in range(2):
            ops.cas_eval(g, h, Q, g1, g2, 31.0, n_occ, ncas, kr, kc, work=work, eri_flags=flags, g_packed=packed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ops.cas_eval(g, h, Q, g1, g2, 31.0, n_occ, ncas, kr, kc, work=work, eri_flags=flags, g_packed=packed)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 5
    t_gen = timed(0)
    # make the tensor exactly p<->q and r<->s symmetric in place (values stay O(1): timing only)
    g.add_(g.transpose(0, 1).clone()).mul_(0.5)
    g.add_(g.transpose(2, 3).clone()).mul_(0.5)
    flags = ops.eri_flags(g)
    t_sym_unpacked = timed(flags)
    # the resident tile-packed copy an OO_energy / OO_pqc object keeps of such integrals (ops.eri_pack, one-off):
    # slabs p <= q, tile triangle of each slab, in the order and with the weights stage 1 consumes it
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    packed = ops.eri_pack(g)
    torch.cuda.synchronize()
    t_pack = time.perf_counter() - t0
    t_sym = timed(flags, packed)
    packed_bytes = 8.0 * packed.numel()
    stage1_kernel = aoo._lib.load().oovqe_last_stage1_kernel().decode()
    del packed
    full = 8.0 * N ** 4
    # both flags, N > 48: slabs p <= q, and per slab the row tiles up to the diagonal one of every
    # 16-column tile (half_stream_kernel<.., RS>)
    nt16 = (N + 15) // 16
    tri_rows = sum(min(N, 16 * (t + 1)) * min(16, N - 16 * t) for t in range(nt16))
    sym_bytes = 8.0 * (N * (N + 1) / 2.0) * tri_rows
    return {"N": N, "ncas": ncas, "nelecas": nelecas, "n_occ": n_occ, "M": M, "n_kappa": int(kr.numel()),
            "general_tensor": {"us": t_gen * 1e6, "stage1_bytes": full, "effective_GBs": full / t_gen / 1e9},
            "symmetric_tensor": {"us": t_sym * 1e6, "eri_flags": int(flags),
                                 "stage1_kernel": stage1_kernel,
                                 "stage1_bytes": packed_bytes,
                                 "effective_GBs": packed_bytes / t_sym / 1e9,
                                 "packed_copy_GB": packed_bytes / 1e9, "pack_once_ms": t_pack * 1e3,
                                 "without_packed_copy": {"us": t_sym_unpacked * 1e6, "stage1_bytes": sym_bytes,
                                                         "effective_GBs": sym_bytes / t_sym_unpacked / 1e9}},
            "note": "energy + orbital gradient for one RDM set; effective_GBs = bytes of g_ao the stage-1 sweep "
                    "must read (all of it; with both symmetry flags the slabs p <= q and in each slab the "
                    "row tiles up to the diagonal one -- from the resident tile-packed copy, as OO_pqc evaluates; "
                    "'without_packed_copy': the same tiles picked out of g_ao) / whole-evaluation time (the later "
                    "stages are inside the time)"}


def launch_command(n_ranks, port, argv):
    """The driver's own multi-GPU command line (one rank per GPU, rendezvous on 127.0.0.1)."""
    import socket
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
            f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1", "--master-port", str(port),
            os.path.abspath(__file__)] + list(argv)


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks as a child
    torch.distributed.run job and return its exit code.  Nothing in this process has touched the
    GPU yet (no exec of a GPU-initialised process: the launcher is a plain child process)."""
    import subprocess
    cmd = launch_command(args.gpus, args.master_port, sys.argv[1:])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # a multi-rank job has RCCL's streams next to the library's side streams: give each of them a hardware
        # queue of its own (HIP's default of four would make two of them run one after the other).  Read by the
        # HIP runtime when it initialises, i.e. below; one rank alone stays on the default.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("OOVQE_BENCH_ECHO_RANK"):
        print(f"bench.py: rank {rank} of {world}", file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        if os.environ.get("OOVQE_BENCH_ECHO_RANK"):
            time.sleep(3.0)     # (the launcher test counts the ranks' lines: the first failing rank ends its siblings)
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if args.backend == "gloo":
        local_rank = 0                      # rehearsal: every rank uses the only GPU
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    from auto_oo_amd import ops
    from auto_oo_amd.parallel import shard_geometries, gather_results

    # weak scaling: every GPU owns args.geoms geometries; geometry g of the job lives on rank g mod world
    n_geom_total = args.geoms * world
    my_geoms = shard_geometries(n_geom_total, rank, world)
    pqc, batch, single, thetas = build_geometries(my_geoms)
    G = batch.G
    if args.general_eri:
        batch.eri_flags = 0
    n_out = 1 + batch.n_theta + batch.n_kappa
    results = torch.zeros((G, n_out), dtype=torch.float64, device="cuda")

    # (multi-rank jobs always keep the in-order calls: their RCCL streams and GPU_MAX_HW_QUEUES = 8 change how HIP maps
    # streams to hardware queues, and the deferred form has only been measured on one-rank boxes)
    deferred = args.deferred and world == 1
    step_marks = None          # (debugging aid: OOVQE_BENCH_STEP_TIMES=1 records an event behind every step of the timed region)

    def run(n_calls, defer=None):
        """n_calls steps: each one batched call over ALL G geometries of this rank's shard.  Each
        call returns its own [G, 1 + n_theta + n_kappa] result tensor; the rows of the last call
        are what the final exchange gathers.  The steps are independent evaluations; with --deferred each call is
        DEFERRED (OO_pqc_batch.energy_and_gradient(defer=True): enqueued on the library's two side streams in
        turn, each with its own workspace) and all of them are joined to the current stream before the exchange --
        the latency-bound tail of one call (q -> x / p -> n, Fock panels, assembly) then runs under the N^4 sweep of
        the next call instead of in front of it.  Per call the same launches, the same bits."""
        defer = deferred if defer is None else defer
        last, pend = None, []
        for _ in range(n_calls):
            if defer:
                pend.append(batch.energy_and_gradient(thetas, defer=True))
            else:
                last = batch.energy_and_gradient(thetas)
            if step_marks is not None:
                ev_ = torch.cuda.Event(enable_timing=True)
                ev_.record()
                step_marks.append(ev_)
        # every call joined: the side streams are in-order, so the last call of each of the two streams stands for all
        # calls before it -- all n_calls result tensors are complete behind these two waits
        for p_ in pend[-2:]:
            p_.wait()
        if pend:
            last = pend[-1].result()
        if last is not None:
            results.copy_(last)
        return n_calls

    # set-up (not part of W or K): every code path of the timed region runs once, so that lazily
    # loaded code objects (ours and torch's index_put/copy kernels used by the final exchange: ~30 ms
    # the first time a fresh box reads them from disk) are resident, and the GPU clocks have ramped.
    # (the library's pool of HIP events is made on the first profile_begin: 16 384 hipEventCreate calls, ~20 ms of host
    # time during which the GPU idles and its clocks fall -- made here, not in front of the timed region)
    ops.profile_begin()
    ops.profile_end()
    # The exchange runs once BEFORE the priming: its first call reads torch's index_put kernel from disk (tens of
    # milliseconds with an idle GPU).  Rounds 1-4 made that call between the priming and the warm-up steps, the clocks
    # fell during it and needed ~25 steps (12 ms) to come back: with the driver's --warmup 5 the 20 timed steps ran
    # through that ramp (540 -> 475 us per step: OOVQE_BENCH_STEP_TIMES=1 prints the intervals; --warmup 30 hid it).
    run(2)
    gather_results(results, my_geoms, n_geom_total, dist)
    torch.cuda.synchronize()
    t_prime = time.perf_counter()
    while time.perf_counter() - t_prime < args.prime_seconds:
        run(8)
        torch.cuda.synchronize()
    run(args.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ops.profile_begin()   # HIP events around every half-transform launch, on its stream
    if os.environ.get("OOVQE_BENCH_STEP_TIMES"):
        step_marks = [torch.cuda.Event(enable_timing=True)]
        step_marks[0].record()
    t0 = time.perf_counter()
    n_calls = run(args.steps)
    t_submit = time.perf_counter() - t0                              # host time to enqueue everything
    gathered = gather_results(results, my_geoms, n_geom_total, dist)   # the one exchange step
    t_gather = time.perf_counter() - t0
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64,
                            device="cpu" if args.backend == "gloo" else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    kern_total_ms, kern_count, _ = ops.profile_end()
    if step_marks is not None:
        print("step intervals (us):", [round(a_.elapsed_time(b_) * 1e3, 1) for a_, b_ in zip(step_marks, step_marks[1:])],
              file=sys.stderr, flush=True)
        step_marks = None

    # extras (untimed for `value`; SHORT BURSTS of 40 calls after other work, not the sustained regime of the headline:
    # the chip runs this workload against its power limit and a burst finds higher clocks): the same steps as in-order
    # calls on the current stream and as deferred calls over the two side streams, and deferred calls of 224 of the
    # 256 geometries (the sweep then leaves 32 CUs to the tails of the previous call)
    pipelining = None
    if world == 1 and not args.general_eri:
        def timed_calls(n, count, defer):
            torch.cuda.synchronize()
            t = time.perf_counter()
            pend = [batch.evaluate_deferred(thetas, count=count) if defer else batch.evaluate(thetas, count=count)
                    for _ in range(n)]
            if defer:
                for p_ in pend:
                    p_.result()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n
        pipelining = {"headline_calls": "deferred" if deferred else "in order"}
        for label, count, defer in (("in_order_256", G, False), ("deferred_256", G, True),
                                    ("deferred_224", G - 32, True)):
            if count < 1:
                continue
            timed_calls(6, count, defer)
            t_c = min(timed_calls(40, count, defer) for _ in range(2))
            pipelining[label] = {"geometries_per_call": count, "us_per_call": t_c * 1e6,
                                 "evaluations_per_s": count / t_c}
        pipelining["note"] = ("independent calls in bursts of 40; deferred = OO_pqc_batch.evaluate_deferred (two side streams / "
                              "workspace slots in turn, N^4 sweeps of the two streams ordered one after the other, all calls "
                              "joined at the end); bit-identical to the in-order calls "
                              "(tests/test_full_size_gpu.py::test_deferred_evaluations_equal_the_in_order_calls_bit_for_bit).  "
                              "In the SUSTAINED regime of the headline (0.5 s of priming in front of the timed steps) both forms "
                              "give the same evaluations/s to within the run-to-run spread (bench.py --deferred): the sweep "
                              "slows by what the overlapped tails take (its HIP-event duration 370 -> 410 us) -- the chip is "
                              "power-limited on this workload (DESIGN.md section 5), so hiding latency does not buy throughput")
    # ingest (untimed for `value`; SURVEY.md section 8(d) fixes g_ao): what it costs to make a geometry's integrals
    # usable once they are in HBM -- the bitwise symmetry tests and the packed resident copy, ONE pass over the stack
    # (oovqe_eri_ingest) -- and the host -> device copy of one geometry's tensors from pinned memory, separately
    ingest = None
    if world == 1 and not args.general_eri:
        torch.cuda.synchronize()
        t_i = []
        for _ in range(3):
            t = time.perf_counter()
            batch.reverify_integrals()
            torch.cuda.synchronize()
            t_i.append(time.perf_counter() - t)
        t_ing = min(t_i)
        tensor_bytes = 8.0 * NAO ** 4
        packed_bytes = 8.0 * batch._eri_packed.shape[1] if batch._eri_packed is not None else 0.0
        host = torch.empty((NAO,) * 4, dtype=torch.float64).pin_memory()
        host.copy_(batch.int2e_ao[0])
        dst = torch.empty_like(batch.int2e_ao[0])
        dst.copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(8):
            dst.copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        t_h2d = (time.perf_counter() - t) / 8
        del host, dst
        ingest = {"setup_us_per_geometry": t_ing / G * 1e6, "geometries": G, "ms_per_stack": t_ing * 1e3,
                  "bytes_per_geometry": tensor_bytes + packed_bytes,
                  "effective_GBs": (tensor_bytes + packed_bytes) * G / t_ing / 1e9,
                  "frac_of_hbm_peak": (tensor_bytes + packed_bytes) * G / t_ing / 1e9 / HBM_PEAK_GBS,
                  "h2d_us_per_geometry_pinned": t_h2d * 1e6, "h2d_GBs": tensor_bytes / t_h2d / 1e9,
                  "evaluation_calls_equivalent": None,
                  "note": "OO_pqc_batch.reverify_integrals() = oovqe_eri_ingest over the stack (every slab of g_ao read "
                          "once: bitwise p<->q / r<->s tests, packed copy written; rounds 1-4: a check pass + a pack "
                          "pass, 3.24 ms per 256 geometries) + mo_coeff refresh, one host readback of the flags included; "
                          "the host->device copy of a geometry's 27 MB (pinned) is NOT in it and NOT in any figure of "
                          "this line (SURVEY.md section 8(d): inputs resident in HBM)"}
    # per-launch breakdown of one evaluation call, from a separate untimed pass (bracketing every
    # launch costs dispatch gaps, so it is kept out of the timed region)
    ops.profile_begin(detail=True)
    run(16)
    torch.cuda.synchronize()
    kern_by = ops.profile_end()[2]
    kern_s = kern_total_ms * 1e-3 / kern_count if kern_count else float("nan")
    M = batch._n_occ + NCAS
    # which stage-1 kernel the library picks (cas.hip: fused_plan): the persistent T3 kernel once a
    # launch covers >= 48 slabs per CU, the one-slab-per-wave T2 kernel below that
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    t3_path = G * NAO ** 2 >= 48 * n_cu
    # p<->q symmetric integrals (verified bit for bit when the batch was built, as PySCF's int2e
    # is): the library reads only the N(N+1)/2 slabs p <= q -- the algorithmic bytes of the sweep
    # are then 8 N^2 * N(N+1)/2 (half of SURVEY.md section 8(d)'s 8 N^4) + the packed J written
    pq_sym = bool(batch.eri_flags & ops.ERI_PQ_SYMMETRIC)
    rs_sym = bool(batch.eri_flags & ops.ERI_RS_SYMMETRIC)
    # the stage-1 kernel the library actually dispatched for the timed calls (name + template arguments)
    from auto_oo_amd import _lib as _aoo_lib
    dispatched = _aoo_lib.load().oovqe_last_stage1_kernel().decode()
    tri = NAO * (NAO + 1) // 2
    if pq_sym and t3_path:
        bytes_per_eval = 8.0 * NAO ** 2 * tri + 8.0 * tri * M ** 2
        if rs_sym:
            # each slab is symmetric too: the packed copy holds its upper triangle only, and only
            # the columns y <= z of its result are written
            slab_elems = sum(NAO - (r & ~1) for r in range(NAO))   # upper triangle, even row starts
            bytes_per_eval = 8.0 * slab_elems * tri + 8.0 * tri * (M * (M + 1) // 2)
        kernel_name = (dispatched
                       + " (J[p<=q,y,z] = sum_rs C[r,y] g[p,q,r,s] C[s,z] over the "
                       "upper triangle of slabs"
                       + (", streaming the packed copy (upper triangle of each slab)"
                          if rs_sym else "")
                       + "; integrals verified p<->q" + (" and r<->s" if rs_sym else "") + " symmetric)")
    elif pq_sym:
        bytes_per_eval = 8.0 * NAO ** 2 * tri + 8.0 * NAO ** 2 * M ** 2
        kernel_name = (dispatched + ", slabs p <= q mirrored into T2[p,q] and T2[q,p] "
                       "(integrals verified p<->q symmetric)")
    elif t3_path:
        # g_ao read once + T3 (8 N M^3) written
        bytes_per_eval = 8.0 * NAO ** 4 + 8.0 * NAO * M ** 3
        kernel_name = (dispatched + " (T3[p,x,y,z] = sum_q C[q,x] "
                       "sum_rs C[r,y] g[p,q,r,s] C[s,z])")
    else:
        bytes_per_eval = 8.0 * NAO ** 4 + 8.0 * NAO ** 2 * M ** 2
        kernel_name = dispatched + " (T2[p,q,y,z] = sum_rs C[r,y] g[p,q,r,s] C[s,z])"
    # evaluations per launch from the calls made (the event pool brackets at most 8192 launches of
    # a long run: the average duration is then over those, the bytes are still per launch)
    evals_per_launch = float(G)
    if kern_count < min(args.steps, 8192):
        raise SystemExit(f"roofline: only {kern_count} of {args.steps} stage-1 launches were bracketed")
    alg_bytes = bytes_per_eval * evals_per_launch                   # per launch (batched)
    achieved = alg_bytes / kern_s / 1e9
    # HBM traffic of that kernel from the rocprofv3 PMC passes (FETCH_SIZE x2 [gfx950 correction] +
    # WRITE_SIZE, separate passes; profiles/pmc_half_transform.json), scaled to this launch size
    traffic, traffic_note = None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_half_transform.json")
    if os.path.exists(pmc_path) and t3_path:        # (the PMC passes were taken on the batched kernel)
        try:
            import hashlib
            with open(pmc_path) as fh:
                pmc = json.load(fh)
            with open(os.path.join(ROOT, "auto_oo_amd", "csrc", "cas.hip"), "rb") as fh:
                sha = hashlib.sha256(fh.read()).hexdigest()
            # the counters belong to ONE build of the stage-1 source and to ONE kernel: anything else gets no figure
            if pmc.get("cas_hip_sha256") != sha:
                traffic_note = "profiles/pmc_half_transform.json was measured on another version of cas.hip"
            elif not pmc.get("kernel", "").startswith(dispatched):
                traffic_note = f"profiles/pmc_half_transform.json is about {pmc.get('kernel')!r}, not {dispatched!r}"
            elif pmc.get("pq_symmetric", False) == pq_sym:
                traffic = pmc["hbm_bytes_per_launch"] / pmc.get("geometries_per_launch", 64) * evals_per_launch
        except Exception as exc:
            traffic, traffic_note = None, f"pmc file unreadable: {exc}"

    out = {
        "metric": METRIC,
        "value": world * args.steps * G / elapsed,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"configs[1] shape: N={NAO} AOs, n_occ=6, CAS(4e,3o), UCCD n_theta=4, "
                         f"n_kappa={batch.n_kappa}; one step = one batched call evaluating energy + full "
                         f"gradient of all {G} geometries of the rank ({n_geom_total} synthetic "
                         f"geometries in the job, geometry g on rank g mod n_gpus); value = evaluations/s; "
                         + ("the steps are independent calls, deferred over the library's two side streams and all "
                            "joined before the exchange" if deferred else "in-order calls on one stream")),
            "evals_per_step": G,
            "geometries_per_rank": G,
            "batched_calls": n_calls,
            "calls": "deferred (OO_pqc_batch.energy_and_gradient(defer=True))" if deferred else "in order",
            "host_submit_us_per_call": t_submit / max(n_calls, 1) * 1e6,
            "timed_region_us": {"enqueue_and_join": t_submit * 1e6, "exchange_enqueued_at": t_gather * 1e6,
                                "elapsed": elapsed * 1e6},
            "parallelism": f"geometry-sharded x{world}, one all_gather at the end",
            "pipelining": pipelining,
            "ingest": ingest,
            # what torch.distributed itself reports (a SCALE line proves RCCL saw N ranks)
            "dist_backend": dist.get_backend() if dist is not None else None,
            "dist_world_size": dist.get_world_size() if dist is not None else 1,
            "devices_visible": torch.cuda.device_count(),
            "device_of_rank0": torch.cuda.current_device(),
        },
        "roofline": {
            "kernel": kernel_name,
            "eri_pq_symmetric": pq_sym,
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_note": traffic_note,
            "kernel_dispatched": dispatched,
            "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_bytes_per_eval": bytes_per_eval,
            # SURVEY.md section 8(d)'s figure reads the full tensor (every integral 4x): for reference
            "full_tensor_bytes_per_eval": 8.0 * NAO ** 4,
            "full_tensor_equivalent_GBs": 8.0 * NAO ** 4 * evals_per_launch / kern_s / 1e9,
            "evals_per_launch": evals_per_launch,
            "avg_launch_us": kern_s * 1e6,
            "all_launches_avg_us": {k: (v[0] / v[1] * 1e3 if v[1] else None) for k, v in kern_by.items()},
            "launches_timed": kern_count,
        },
    }
    if ingest is not None:
        ingest["evaluation_calls_equivalent"] = ingest["ms_per_stack"] / out["ms_per_step"]
    if pq_sym and rs_sym and t3_path and NAO == 43 and M == 9:
        # The same launch against the fp64 matrix pipe (informational; DESIGN.md "What bounds stage 1": the
        # kernel is bound by MFMA issue at a power-limited clock, not by HBM).  34 v_mfma_f64_16x16x4 per
        # slab (11 + 4 + 8 first products, 3 + 4 + 4 second), 2 048 flop each, padding included.
        mfma_flops = 34 * 2048.0 * tri * evals_per_launch
        out["roofline"]["mfma_view"] = {
            "mfma_instructions_per_slab": 34,
            "achieved_TFLOPs_incl_tile_padding": mfma_flops / kern_s / 1e12,
            "peak_TFLOPs": 78.6,
            "frac_of_peak_at_2.4GHz": mfma_flops / kern_s / 1e12 / 78.6,
            # 34 x 64 pipe cycles per slab, 946 slabs per geometry, 1 024 SIMDs: the launch cannot be shorter
            "pipe_floor_us_at_2.4GHz": 34 * 64.0 * tri * evals_per_launch / 1024 / 2.4e3,
            "pipe_floor_us_at_1.75GHz": 34 * 64.0 * tri * evals_per_launch / 1024 / 1.75e3,
            "frac_of_pipe_floor_at_1.75GHz": (34 * 64.0 * tri * evals_per_launch / 1024 / 1.75e3) / (kern_s * 1e6),
            "useful_flop_share_of_issued": (2.0 * 967 * 9 + 2.0 * 43 * 81) / (34 * 2048.0),
            "note": "the core clock under this kernel is ~1.75 GHz (power-limited with the HBM stream: tools/tri_spread.hip, "
                    "profiles/r02_t_stage1_cycles.txt); against the matrix pipe at that clock the launch stands at "
                    "frac_of_pipe_floor -- the HBM fraction understates how close the kernel is to ITS bound",
        }
    if n_geom_total <= 64:
        # small jobs (tests): the gathered energies bit for bit, geometry by geometry
        out["gathered_energies_hex"] = [float(e).hex() for e in gathered[:, 0].tolist()]
    if rank == 0 and world == 1:
        # latency of ONE un-batched evaluation through the drop-in API (not the headline)
        th0 = thetas[0].contiguous()
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.2:          # clocks up after the host-side work above
            for _ in range(50):
                single.energy_and_gradient(th0)
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(500):
            single.energy_and_gradient(th0)
        torch.cuda.synchronize()
        out["single_eval_us"] = (time.perf_counter() - t1) / 500 * 1e6
    if not args.no_berry:
        # configs[3]: the 64-geometry Berry-phase-loop workload.  "strong": --berry-geoms geometries in
        # the whole job, split over the GPUs (the north star's >= 6x at 8 GPUs is about this one);
        # "weak": --berry-geoms geometries per GPU (the same run when n_gpus = 1).
        nb = max(args.berry_geoms, world)
        strong_geoms = shard_geometries(nb, rank, world)
        strong = berry_loop_extra(strong_geoms, nb, dist, world, args.backend, "strong")
        if world > 1:
            weak_geoms = shard_geometries(nb * world, rank, world)
            weak = berry_loop_extra(weak_geoms, nb * world, dist, world, args.backend, "weak")
        else:
            weak = dict(strong, scaling=f"weak ({nb} geometries per GPU; identical to the strong run at 1 GPU)")
        # the same workload in the regime the reference's loop runs in (one step per loop point from the
        # previous point's solution: positive definite Hessians), strong figure only
        tracking = berry_loop_extra(strong_geoms, nb, dist, world, args.backend, "strong", regime="tracking")
        if rank == 0:
            out["berry_loop"] = {"strong": strong, "weak": weak, "tracking_strong": tracking}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            cb, e_ref, g_ref = cpu_baseline()
            out["cpu_baseline"] = cb
            # parity of what was just benchmarked (geometry 0) against the oracle
            e_gpu = float(gathered[0, 0].item())
            out["parity"] = {"dE_vs_oracle": abs(e_gpu - e_ref),
                             "max_dgrad_vs_oracle":
                                 float((gathered[0, 1:].cpu() - g_ref).abs().max())}
        if world == 1 and not args.no_transform:
            out["transform"] = transform_microbench(args.transform_n)
        if world == 1 and not args.no_kupccd:
            out["kupccd_cas88"] = kupccd_extra()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
