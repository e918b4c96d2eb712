"""Which rows of a mode contraction differ from einsum (tools only)."""
import numpy as np, torch
from auto_oo_amd import ops
DEV = "cuda"
for last, A, K, J, B in [(0, 1, 13, 13, 2197), (0, 13, 13, 13, 169), (0, 3, 43, 9, 81), (0, 2, 100, 250, 40), (0, 1, 4, 1, 1),
                         (0, 7, 41, 16, 16), (1, 50, 13, 13, 1), (1, 1000, 100, 250, 1), (0, 1, 200, 200, 4000)]:
    rng = np.random.default_rng(1)
    if last:
        T = torch.tensor(rng.standard_normal((A, K))); C = torch.tensor(rng.standard_normal((K, J)))
        ref = T @ C
        out = ops.mode_contract(T.to(DEV), C.to(DEV), A, K, J, 1, last=True).cpu().reshape(A, J)
        bad = ((out - ref).abs() > 1e-10).nonzero()
        print("LAST", A, K, J, "bad", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:10], "cols", sorted(set(bad[:, 1].tolist()))[:20])
    else:
        T = torch.tensor(rng.standard_normal((A, K, B))); C = torch.tensor(rng.standard_normal((K, J)))
        ref = torch.einsum("kj,akb->ajb", C, T)
        out = ops.mode_contract(T.to(DEV), C.to(DEV), A, K, J, B, last=False).cpu().reshape(A, J, B)
        bad = ((out - ref).abs() > 1e-10).nonzero()
        print("INNER", A, K, J, B, "bad", len(bad), "a", sorted(set(bad[:, 0].tolist()))[:10], "j", sorted(set(bad[:, 1].tolist()))[:20],
              "b", sorted(set(bad[:, 2].tolist()))[:20])
for A, K, J, B in [(1, 200, 200, 4000), (3, 96, 96, 96 * 96), (96, 96, 96, 96), (2, 100, 250, 40), (5, 77, 90, 1000), (2, 130, 208, 2046), (7, 60, 81, 34)]:
    rng = np.random.default_rng(2)
    T = torch.tensor(rng.standard_normal((A, K, B))); C = torch.tensor(rng.standard_normal((K, J)))
    ref = torch.einsum("kj,akb->ajb", C, T)
    out = ops.mode_contract(T.to(DEV), C.to(DEV), A, K, J, B, last=False).cpu().reshape(A, J, B)
    bad = ((out - ref).abs() > 1e-10).nonzero()
    print("INNER(pair?)", A, K, J, B, "bad", len(bad), "a", sorted(set(bad[:, 0].tolist()))[:10], "j", sorted(set(bad[:, 1].tolist()))[:20],
          "b", sorted(set(bad[:, 2].tolist()))[:20])
