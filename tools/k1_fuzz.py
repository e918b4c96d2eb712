"""Random-shape fuzz of oovqe_mode_contract against einsum (tools only; the suite runs 40 of these)."""
import sys, numpy as np, torch
from auto_oo_amd import ops, _lib
DEV = "cuda"
n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = 0.0
for trial in range(n_trials):
    last = rng.random() < 0.3
    K = int(rng.integers(1, 260))
    J = int(rng.integers(1, 260))
    if last:
        A, B = int(rng.integers(1, 9000)), 1
    else:
        A = int(rng.integers(1, 7))
        B = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 250, 1000, 2999, 4096, 8190, 40002, 70000, 131072]))
    if A * K * B > 3e8 or A * J * B > 3e8:
        continue
    opts = {}
    r = rng.random()
    if r < 0.15: opts = dict(k1_no_pair=1)
    elif r < 0.25: opts = dict(k1_force_wide=1)
    gen = torch.Generator(device=DEV).manual_seed(trial)
    off = int(rng.integers(0, 2))          # odd offsets: buffers only 8-byte aligned
    Tbuf = torch.randn(A * K * B + 1, generator=gen, dtype=torch.float64, device=DEV)
    T = Tbuf[off:off + A * K * B].reshape(A, K, B)
    C = torch.randn((K, J), generator=gen, dtype=torch.float64, device=DEV)
    obuf = torch.empty(A * J * B + 1, dtype=torch.float64, device=DEV)
    out = obuf[off:off + A * J * B]
    with _lib.debug_options(**opts):
        ops.mode_contract(T, C, A, K, J, B, last=last, out=out)
    ref = torch.einsum("kj,akb->ajb", C, T.contiguous())
    err = float((out.reshape(A, J, B) - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    worst = max(worst, err)
    if err > 1e-11:
        print("FAIL", trial, last, A, K, J, B, opts, off, err, flush=True)
        sys.exit(1)
print("ok", n_trials, "trials, worst relative error", worst)
