"""Random shapes through oovqe_cas_eval: symmetry flags 3 and 1 against 0 on 8-fold symmetric integrals
generated on the device (tools only)."""
import sys, numpy as np, torch
import auto_oo_amd as aoo
from auto_oo_amd import ops
DEV = "cuda"
n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = 0.0
done = []
for trial in range(n_trials):
    N = int(rng.integers(5, 140))
    ncas = int(rng.integers(1, 7))
    n_occ = int(rng.integers(0, min(N - ncas, 48 - ncas) + 1))
    if n_occ + ncas >= N:
        continue
    nrdm = int(rng.integers(1, 4))
    M = n_occ + ncas
    gen = torch.Generator(device=DEV).manual_seed(trial)
    B = torch.randn((8, N, N), generator=gen, dtype=torch.float64, device=DEV)
    B = 0.5 * (B + B.transpose(1, 2))
    g = (torch.einsum("Lpq,Lrs->pqrs", B, B) / 8.0).contiguous()
    g = 0.5 * (g + g.transpose(0, 1)); g = (0.5 * (g + g.transpose(2, 3))).contiguous()
    h = torch.randn((N, N), generator=gen, dtype=torch.float64, device=DEV); h = 0.5 * (h + h.T)
    Q, _ = torch.linalg.qr(torch.randn((N, N), generator=gen, dtype=torch.float64, device=DEV))
    g1 = torch.randn((nrdm, ncas, ncas), generator=gen, dtype=torch.float64, device=DEV)
    g2 = torch.randn((nrdm, ncas, ncas, ncas, ncas), generator=gen, dtype=torch.float64, device=DEV)
    rows, cols = aoo.excitations.tril_tables(N, aoo.non_redundant_indices(
        np.arange(n_occ), n_occ + np.arange(ncas), np.arange(M, N), False))
    if len(rows) == 0:
        continue
    kr, kc = torch.as_tensor(rows).to(DEV), torch.as_tensor(cols).to(DEV)
    if ops.eri_flags(g) != 3:
        print("flags", ops.eri_flags(g)); sys.exit(1)
    try:
        outs = [ops.cas_eval(g, h, Q.contiguous(), g1, g2, 1.5, n_occ, ncas, kr, kc, want_matrices=True,
                             want_integrals=True, eri_flags=f) for f in (0, 1, 3)]
    except RuntimeError as e:
        print("skip", N, n_occ, ncas, str(e)[:80]); continue
    done.append((N, n_occ, ncas, nrdm))
    for key in ("c0", "c1", "c2", "E", "gvec", "dE", "fock", "gmat", "Gm", "hmo"):
        a = outs[0][key]
        if a is None or a.numel() == 0:
            continue
        scale = max(1.0, float(a.abs().max()))
        for f, o in zip((1, 3), outs[1:]):
            err = float((a - o[key]).abs().max()) / scale
            worst = max(worst, err)
            if err > 5e-12:
                print("FAIL", trial, N, n_occ, ncas, nrdm, key, f, err, flush=True); sys.exit(1)
print("ok", len(done), "of", n_trials, "trials ran, worst relative error", worst, "largest N", max(d[0] for d in done), "N > 48:", sum(1 for d in done if d[0] > 48))
