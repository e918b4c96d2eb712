"""configs[2] as ONE OO evaluation (N = 200, CAS(6e,6o), n_occ = 20) eight times, symmetric tensor (argv[1] = 0: general;
argv[2] = packed: from the tile-packed copy)
-- for rocprofv3 --kernel-trace (tools/trace_one_call.py, marker cas_final lists the launches of one call)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import auto_oo_amd as aoo
from auto_oo_amd import ops
sym = (sys.argv[1] != "0") if len(sys.argv) > 1 else True
N, ncas, nelecas, n_occ = 200, 6, 6, 20
M = n_occ + ncas
gen = torch.Generator(device="cuda").manual_seed(3)
g = torch.rand((N, N, N, N), dtype=torch.float64, device="cuda", generator=gen).mul_(2).sub_(1)
C = torch.rand((N, N), dtype=torch.float64, device="cuda", generator=gen) - 0.5
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
work = torch.empty(aoo._lib.load().oovqe_cas_eval_work_size(N, n_occ, ncas, 1), dtype=torch.float64, device="cuda")
flags = 0
if sym:
    g.add_(g.transpose(0, 1).clone()).mul_(0.5)
    g.add_(g.transpose(2, 3).clone()).mul_(0.5)
    flags = ops.eri_flags(g)
packed = None
if sym and len(sys.argv) > 2 and sys.argv[2] == "packed":
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    packed = ops.eri_pack(g)
    torch.cuda.synchronize()
    print(f"eri_pack: {(time.perf_counter() - t0) * 1e3:.2f} ms, {packed.numel() * 8 / 1e9:.2f} GB")
for variant in ([0, 1, 2, 4, 0] if packed is not None and len(sys.argv) > 3 else [0]):
    with aoo._lib.debug_options(tiles_variant=variant):
        for _ in range(3):
            ops.cas_eval(g, h, Q, g1, g2, 31.0, n_occ, ncas, kr, kc, work=work, eri_flags=flags, g_packed=packed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            ops.cas_eval(g, h, Q, g1, g2, 31.0, n_occ, ncas, kr, kc, work=work, eri_flags=flags, g_packed=packed)
        torch.cuda.synchronize()
    print(f"flags {flags}{' packed' if packed is not None else ''} variant {variant}: "
          f"{(time.perf_counter() - t0) / 8 * 1e6:.1f} us per evaluation", flush=True)
