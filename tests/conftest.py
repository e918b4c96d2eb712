import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle runs on torch's intra-op threads.  A one-GPU box shows all 256 CPUs of its host but grants a
    # share of 16: with torch's default (one thread per visible CPU) the oracle's autograd Hessians crawl -- the
    # N = 43 Hessian of test_config3_unit_of_work_at_cc_pvdz_shape took 285 s there and 9 s on 8 threads here.
    import torch
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(share, os.cpu_count() or 1, 16)))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def lib_options():
    """Set the library's test / measurement switches for one test (include/oovqe.h:
    oovqe_debug_set_option) and restore them afterwards: ``lib_options(cas_unfused=1)``."""
    from auto_oo_amd import _lib
    active = []

    def set_options(**options):
        ctx = _lib.debug_options(**options)
        ctx.__enter__()
        active.append(ctx)
    yield set_options
    for ctx in reversed(active):
        ctx.__exit__(None, None, None)
