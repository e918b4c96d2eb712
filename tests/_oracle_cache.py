"""Oracle results that take minutes of host CPU (autograd Hessians through three N^5 transforms at N = 43) as
committed fixtures: ``oracle_values(key, compute)`` returns the arrays stored in
tests/golden/oracle_cache/<key>.npz when the file exists and calls ``compute()`` -- the oracle itself -- otherwise
(and writes the file when OOVQE_WRITE_ORACLE_CACHE is set: tests/golden/make_oracle_cache.py is the committed
generator; tests/test_oracle_goldens.py re-derives the cheap entries of every file on the CPU, so a stale fixture
cannot pass).  Symmetric matrices are stored as their upper triangle."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CACHE = os.path.join(HERE, "golden", "oracle_cache")


def pack_sym(M):
    M = np.asarray(M)
    return M[np.triu_indices(M.shape[0])]


def unpack_sym(v):
    v = np.asarray(v)
    n = int(round((np.sqrt(8 * v.size + 1) - 1) / 2))
    M = np.zeros((n, n))
    M[np.triu_indices(n)] = v
    return M + np.triu(M, 1).T


def oracle_values(key, compute):
    path = os.path.join(CACHE, key + ".npz")
    if os.path.exists(path) and not os.environ.get("OOVQE_WRITE_ORACLE_CACHE"):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    out = {k: np.asarray(v) for k, v in compute().items()}
    if os.environ.get("OOVQE_WRITE_ORACLE_CACHE"):
        os.makedirs(CACHE, exist_ok=True)
        np.savez(path, **out)
    return out
