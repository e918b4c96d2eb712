"""Generator of tests/golden/oracle_cache/*.npz: runs the CPU oracle (oracle/cpu_ref.py) for the comparisons whose
oracle side takes minutes of host time (full Hessians by autograd through three N^5 transforms at N = 43), exactly
the functions the GPU tests call (tests/test_newton_gpu.py: oracle_config3_n43, oracle_small_hessians,
oracle_hessian_n43_rng9).  No GPU needed.
    OOVQE_WRITE_ORACLE_CACHE=1 python tests/golden/make_oracle_cache.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["OOVQE_WRITE_ORACLE_CACHE"] = "1"

from tests import test_newton_gpu as T      # noqa: E402  (imports without a GPU; nothing is launched)

for name, fn in (("small 13/3", lambda: T.oracle_small_hessians(13, 3, False)),
                 ("small 13/2 frozen", lambda: T.oracle_small_hessians(13, 2, True)),
                 ("small 20/5", lambda: T.oracle_small_hessians(20, 5, False)),
                 ("hessian n43 rng9", T.oracle_hessian_n43_rng9),
                 ("config3 n43", T.oracle_config3_n43)):
    t0 = time.perf_counter()
    out = fn()
    print(f"{name}: {time.perf_counter() - t0:.1f} s, {sorted(out)}", flush=True)
