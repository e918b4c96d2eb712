"""auto_oo_amd: MI355X-native orbital-optimized VQE engine behind auto_oo's cost-function API.

Export list mirrors the reference's src/auto_oo/__init__.py:3-27 for the hot path
(OO_energy / OO_pqc / Parameterized_circuit and helpers); ``Moldata`` is the array-backed stand-in
for ``Moldata_pyscf``.  Importing the package needs the built HIP library
(auto_oo_amd/lib/liboovqe_hip.so): there is no CPU fallback.
"""
from . import _lib

_lib.load()   # fail loudly at import time when the HIP extension is missing

from .pqc import Parameterized_circuit                      # noqa: E402
from .moldata import Moldata, ao_to_oao, get_formal_geo                     # noqa: E402
from .gaussian import Moldata_sto3g                         # noqa: E402
from .oo_pqc import OO_pqc                                  # noqa: E402
from .batch import OO_pqc_batch                             # noqa: E402
from .oo_energy import (                                    # noqa: E402
    OO_energy,
    mo_ao_to_mo_oao,
    int1e_transform,
    int2e_transform,
    general_4index_transform,
    uniform_4index_transform,
    vector_to_skew_symmetric,
    skew_symmetric_to_vector,
    non_redundant_indices,
)
from .newton_raphson import NewtonStep, BatchedNewtonStep   # noqa: E402
from .berry import ActiveSpaceRotation, bogoliubov_atob_cas, state_overlap   # noqa: E402
from .excitations import generalized_pair_doubles           # noqa: E402
from .active_space import active_space_integrals, molecular_hamiltonian_coefficients   # noqa: E402

__all__ = [
    "Parameterized_circuit", "Moldata", "Moldata_sto3g", "ao_to_oao", "get_formal_geo", "OO_pqc", "OO_pqc_batch", "OO_energy", "mo_ao_to_mo_oao",
    "int1e_transform", "int2e_transform", "general_4index_transform", "uniform_4index_transform",
    "vector_to_skew_symmetric", "skew_symmetric_to_vector", "non_redundant_indices", "NewtonStep", "BatchedNewtonStep", "ActiveSpaceRotation", "bogoliubov_atob_cas", "state_overlap",
    "generalized_pair_doubles", "active_space_integrals", "molecular_hamiltonian_coefficients",
]
