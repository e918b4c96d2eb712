"""ctypes binding of liboovqe_hip.so (the C ABI declared in include/oovqe.h).

There is NO CPU fallback: if the shared library is missing, or a compute entry point is called
without a HIP device, this module raises.  PyTorch is used only as the owner of device memory and
streams; every pointer handed to the library is ``tensor.data_ptr()`` of a contiguous CUDA(HIP)
tensor.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OOVQE_LIB_PATH") or os.path.join(_HERE, "lib", "liboovqe_hip.so")

c_double_p = ctypes.c_void_p
c_int32_p = ctypes.c_void_p
c_stream = ctypes.c_void_p


class OovqeError(RuntimeError):
    pass


class GateT(ctypes.Structure):
    """Mirror of oovqe_gate_t (include/oovqe.h)."""
    _fields_ = [("mask_hi", ctypes.c_uint32), ("mask_lo", ctypes.c_uint32),
                ("mask_par", ctypes.c_uint32), ("theta_idx", ctypes.c_int32),
                ("sign", ctypes.c_int32), ("nfix", ctypes.c_int32),
                ("pos", ctypes.c_int32 * 4)]


GATE_NBYTES = ctypes.sizeof(GateT)


class NewtonStepT(ctypes.Structure):
    """Mirror of oovqe_newton_step_t (include/oovqe.h): the argument block of oovqe_oo_newton_step_batch."""
    _POINTERS = ("theta", "gates", "g_ao", "h_ao", "nuc", "g_packed", "oao_coeff", "oao_mo_coeff", "mo_coeff",
                 "kap_row", "kap_col", "pairs", "work_hessian", "work_eval", "work_pd", "work_rest",
                 "work_rest_side", "work_rotate", "out", "hessian", "grad", "energy", "flat", "dp", "lowest",
                 "shift", "info", "t", "state", "flags", "points_a", "points_b", "trial_oao", "trial_mo",
                 "trial_out")
    _DOUBLES = ("lambda_min", "mu", "rho", "alpha", "beta")
    _INTS = ("n_theta", "n_gates", "n_qubits", "N", "n_occ", "ncas", "n_kappa", "n_pairs", "batch", "aug",
             "speculate", "side_wg")
    _fields_ = ([(k, ctypes.c_void_p) for k in _POINTERS] + [(k, ctypes.c_double) for k in _DOUBLES] +
                [(k, ctypes.c_int32) for k in _INTS] + [("init_index", ctypes.c_uint32),
                                                        ("eri_flags", ctypes.c_uint32)])

# name -> (restype, argtypes)   -- must list every symbol of include/oovqe.h
SIGNATURES = {
    "oovqe_version": (ctypes.c_int, []),
    "oovqe_last_error": (ctypes.c_char_p, []),
    "oovqe_device_count": (ctypes.c_int, []),
    "oovqe_last_stage1_kernel": (ctypes.c_char_p, []),
    "oovqe_profile_begin": (ctypes.c_int, []),
    "oovqe_profile_begin_detail": (ctypes.c_int, []),
    "oovqe_profile_end": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(ctypes.c_int)]),
    "oovqe_profile_end_labels": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double),
                                                ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    "oovqe_general_4index_transform": (ctypes.c_int, [c_double_p] * 5 + [ctypes.c_int, c_double_p,
                                                                      c_double_p, c_stream]),
    "oovqe_matmul_nn": (ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, c_double_p, c_stream]),
    "oovqe_matmul_nn_batch": (ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, c_double_p, c_stream]),
    "oovqe_matmul_tn": (ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, c_double_p, c_stream]),
    "oovqe_mode_contract": (ctypes.c_int, [c_double_p, c_double_p, c_double_p, ctypes.c_int64,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                           ctypes.c_int, c_stream]),
    "oovqe_expm_skew": (ctypes.c_int, [c_double_p, c_int32_p, c_int32_p, ctypes.c_int, ctypes.c_int,
                                       c_double_p, c_double_p, c_double_p, c_stream]),
    "oovqe_expm": (ctypes.c_int, [c_double_p, ctypes.c_double, ctypes.c_int, c_double_p, c_double_p,
                                  c_stream]),
    "oovqe_circuit_state": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_uint32, ctypes.c_int, c_double_p,
                                           c_double_p, c_stream]),
    "oovqe_rdms": (ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  c_double_p, c_double_p, c_double_p, c_stream]),
    "oovqe_rdms_tangent": (ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, c_double_p, c_double_p,
                                          c_double_p, c_stream]),
    "oovqe_circuit_rdms": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int,
                                          ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                          c_double_p, c_double_p, c_stream]),
    "oovqe_cas_half_transform": (ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int, ctypes.c_int,
                                                c_double_p, c_stream]),
    "oovqe_cas_finish_transform": (ctypes.c_int, [c_double_p, c_double_p, c_double_p, ctypes.c_int,
                                                  ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                                  c_stream]),
    "oovqe_cas_energy_gradient": (ctypes.c_int, [c_double_p] * 4 + [ctypes.c_int, ctypes.c_double,
                                                                   ctypes.c_int, ctypes.c_int,
                                                                   ctypes.c_int, c_int32_p,
                                                                   c_int32_p, ctypes.c_int]
                                  + [c_double_p] * 8 + [c_stream]),
    "oovqe_cas_energy_gradient_ws": (ctypes.c_int, [c_double_p] * 4 + [ctypes.c_int, ctypes.c_double,
                                                                      ctypes.c_int, ctypes.c_int,
                                                                      ctypes.c_int, c_int32_p,
                                                                      c_int32_p, ctypes.c_int]
                                     + [c_double_p] * 9 + [c_stream]),
    "oovqe_cas_energy_gradient_work_size": (ctypes.c_int64, [ctypes.c_int] * 4),
    "oovqe_cas_eval": (ctypes.c_int, [c_double_p] * 5 + [ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                                        ctypes.c_int, ctypes.c_int, c_int32_p,
                                                        c_int32_p, ctypes.c_int]
                       + [c_double_p] * 11 + [ctypes.c_uint, c_stream]),
    "oovqe_cas_eval_packed": (ctypes.c_int, [c_double_p] * 5 + [ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                                               ctypes.c_int, ctypes.c_int, c_int32_p,
                                                               c_int32_p, ctypes.c_int]
                              + [c_double_p] * 11 + [ctypes.c_uint, c_double_p, c_stream]),
    "oovqe_eri_symmetry_flags": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_int,
                                                ctypes.POINTER(ctypes.c_uint), c_stream]),
    "oovqe_cas_eval_work_size": (ctypes.c_int64, [ctypes.c_int] * 4),
    "oovqe_fock_core_active": (ctypes.c_int, [c_double_p, c_double_p, c_double_p, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, c_double_p, c_double_p,
                                              c_stream]),
    "oovqe_orbital_hessian": (ctypes.c_int, [c_double_p] * 6 + [ctypes.c_int, ctypes.c_int,
                                                               ctypes.c_int, c_int32_p, c_int32_p,
                                                               ctypes.c_int, c_double_p, c_double_p,
                                                               c_double_p, c_stream]),
    "oovqe_orbital_hessian_work_size": (ctypes.c_int64, [ctypes.c_int] * 3),
    "oovqe_circuit_second_tangents": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_void_p,
                                                     ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                                                     c_int32_p, ctypes.c_int, c_double_p, c_double_p,
                                                     c_stream]),
    "oovqe_circuit_hessian_assemble": (ctypes.c_int, [c_double_p, c_double_p, c_double_p, c_double_p,
                                                      ctypes.c_int, c_int32_p, ctypes.c_int,
                                                      ctypes.c_int, c_double_p, c_stream]),
    "oovqe_circuit_hessian": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_uint32, c_double_p,
                                             c_double_p, c_int32_p, ctypes.c_int, c_double_p, c_double_p,
                                             c_stream]),
    "oovqe_circuit_hessian_work_size": (ctypes.c_int64, [ctypes.c_int] * 4),
    "oovqe_sector_state": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_uint32, c_int32_p, c_int32_p,
                                          c_int32_p, c_int32_p, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, c_double_p, c_double_p, c_stream]),
    "oovqe_sector_state_deriv": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_uint32, c_int32_p, c_int32_p,
                                                c_int32_p, c_int32_p, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, c_int32_p, ctypes.c_int, c_double_p, c_stream]),
    "oovqe_sector_rdms": (ctypes.c_int, [c_double_p, ctypes.c_int, c_int32_p, c_int32_p, c_int32_p,
                                         c_int32_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         c_double_p, c_double_p, c_double_p, c_stream]),
    "oovqe_sector_adjoint": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                            ctypes.c_int, c_int32_p, c_int32_p, c_int32_p, c_int32_p,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p,
                                            c_double_p, c_double_p, c_double_p, c_double_p, c_stream]),
    "oovqe_sector_work_size": (ctypes.c_int64, [ctypes.c_int] * 4),
    "oovqe_sector_lambda": (ctypes.c_int, [c_double_p, ctypes.c_int, c_int32_p, c_int32_p, c_int32_p, c_int32_p,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p,
                                           ctypes.c_void_p, c_double_p, c_double_p, c_stream]),
    "oovqe_sector_geometry_coefficients_ok": (ctypes.c_int, [ctypes.c_int] * 3),
    "oovqe_sector_adjoint_pg": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                               ctypes.c_int, c_int32_p, c_int32_p, c_int32_p, c_int32_p,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p,
                                               c_double_p, c_double_p, ctypes.c_int64, c_int32_p, ctypes.c_int,
                                               ctypes.c_void_p, c_double_p, c_double_p, c_stream]),
    "oovqe_sector_lambda_pg": (ctypes.c_int, [c_double_p, ctypes.c_int, c_int32_p, c_int32_p, c_int32_p, c_int32_p,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p,
                                              c_double_p, ctypes.c_int64, ctypes.c_void_p, c_double_p, c_double_p,
                                              c_stream]),
    "oovqe_sector_tables_size": (ctypes.c_int64, [ctypes.c_int] * 3),
    "oovqe_sector_rdms_tb": (ctypes.c_int, [c_double_p, ctypes.c_int, c_int32_p, c_int32_p, c_int32_p,
                                            c_int32_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                            c_double_p, c_double_p, c_double_p, c_stream]),
    "oovqe_sector_pairs_size": (ctypes.c_int64, [ctypes.c_int] * 3),
    "oovqe_sector_pairs": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_int32_p, c_int32_p,
                                          c_int32_p, c_int32_p, ctypes.c_int, ctypes.c_int, c_int32_p, c_stream]),
    "oovqe_sector_state_pl": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_uint32, c_int32_p, c_int32_p,
                                             c_int32_p, c_int32_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, c_int32_p, ctypes.c_int, c_double_p, c_double_p,
                                             c_stream]),
    "oovqe_sector_state_deriv_pl": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_uint32, c_int32_p, c_int32_p,
                                                   c_int32_p, c_int32_p, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, c_int32_p, ctypes.c_int, c_int32_p, ctypes.c_int,
                                                   c_double_p, c_stream]),
    "oovqe_sector_adjoint_pl": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                               ctypes.c_int, c_int32_p, c_int32_p, c_int32_p, c_int32_p,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p,
                                               c_double_p, c_double_p, c_int32_p, ctypes.c_int, ctypes.c_void_p,
                                               c_double_p, c_double_p, c_stream]),
    "oovqe_oo_eval": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_uint32, c_double_p, c_double_p, c_double_p,
                                     ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     c_int32_p, c_int32_p, ctypes.c_int, ctypes.c_int, c_double_p,
                                     c_double_p, ctypes.c_uint, c_stream]),
    "oovqe_oo_eval_work_size": (ctypes.c_int64, [ctypes.c_int] * 7),
    "oovqe_oo_eval_out_size": (ctypes.c_int64, [ctypes.c_int] * 4),
    "oovqe_oo_eval_batch": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_uint32, c_double_p, c_double_p,
                                           c_double_p, c_double_p, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, c_int32_p, c_int32_p, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, c_double_p, c_double_p,
                                           ctypes.c_uint, c_double_p, c_stream]),
    "oovqe_eri_packed_size": (ctypes.c_int64, [ctypes.c_int]),
    "oovqe_eri_pack": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_int, c_double_p, c_stream]),
    "oovqe_cas_eval_batch": (ctypes.c_int, [c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, ctypes.c_int,
                                            c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_int32_p, c_int32_p,
                                            ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                            ctypes.c_uint, c_double_p, c_stream]),
    "oovqe_eri_ingest": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_int, c_double_p,
                                        ctypes.POINTER(ctypes.c_uint), c_stream]),
    "oovqe_circuit_rdms_is_small": (ctypes.c_int, [ctypes.c_int] * 4),
    "oovqe_spin_rdms": (ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int, ctypes.c_int, c_double_p,
                                       c_double_p, c_stream]),
    "oovqe_debug_set_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "oovqe_debug_get_option": (ctypes.c_int, [ctypes.c_char_p]),
    "oovqe_newton_direction": (ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                              ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                              c_double_p, c_stream]),
    "oovqe_newton_direction_work_size": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int]),
    "oovqe_newton_direction_pd": (ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_double, c_double_p, c_double_p, c_double_p,
                                                 c_double_p, c_stream]),
    "oovqe_newton_direction_pd_work_size": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int]),
    "oovqe_newton_direction_pd_max_n": (ctypes.c_int, []),
    "oovqe_newton_direction_has_pd": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "oovqe_newton_direction_rest": (ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                                   ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_int,
                                                   c_double_p, c_double_p, c_double_p, c_double_p, c_stream]),
    "oovqe_newton_direction_rest_work_size": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int]),
    "oovqe_newton_direction_max_n": (ctypes.c_int, []),
    "oovqe_orbital_hessian_batch": (ctypes.c_int, [c_double_p] * 6 + [ctypes.c_int, ctypes.c_int,
                                                                     ctypes.c_int, c_int32_p, c_int32_p,
                                                                     ctypes.c_int, ctypes.c_int, c_double_p,
                                                                     c_double_p, ctypes.c_uint, c_stream]),
    "oovqe_circuit_hessian_batch": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_int, ctypes.c_uint32, c_double_p,
                                                   c_double_p, c_int32_p, ctypes.c_int, ctypes.c_int,
                                                   c_double_p, c_double_p, c_stream]),
    "oovqe_oo_hessian_batch": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_uint32, c_double_p, c_double_p,
                                              c_double_p, c_double_p, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, c_int32_p, c_int32_p, ctypes.c_int, c_int32_p,
                                              ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                              ctypes.c_uint, c_double_p, c_stream]),
    "oovqe_oo_hessian_work_size": (ctypes.c_int64, [ctypes.c_int] * 7),
    "oovqe_linesearch_points": (ctypes.c_int, [c_double_p, c_double_p, c_double_p, c_double_p, ctypes.c_double,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p,
                                               c_double_p, c_stream]),
    "oovqe_linesearch_update": (ctypes.c_int, [c_double_p, ctypes.c_int64, c_double_p, c_double_p, c_double_p,
                                               ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               c_double_p, c_double_p, c_double_p, c_double_p, c_stream]),
    "oovqe_rotate_orbitals_batch": (ctypes.c_int, [c_double_p, c_int32_p, c_int32_p, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_int, c_double_p, c_double_p,
                                                   c_double_p, c_double_p, c_stream]),
    "oovqe_oo_newton_step_batch": (ctypes.c_int, [ctypes.POINTER(NewtonStepT), c_stream, c_stream]),
    "oovqe_newton_step_size": (ctypes.c_int, []),
}

_lib = None


def load():
    """Load liboovqe_hip.so and attach the prototypes.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OovqeError(
            f"{LIB_PATH} not found: build it with auto_oo_amd/csrc/build.sh "
            "(or __graft_entry__.build()).  auto_oo_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.oovqe_newton_step_size() != ctypes.sizeof(NewtonStepT):
        raise OovqeError(f"oovqe_newton_step_t is {lib.oovqe_newton_step_size()} bytes in {LIB_PATH}, "
                         f"{ctypes.sizeof(NewtonStepT)} in _lib.NewtonStepT: rebuild the library")
    _lib = lib
    return lib


class debug_options:
    """Context manager for the library's test / measurement switches (include/oovqe.h:
    oovqe_debug_set_option), e.g. ``with debug_options(cas_unfused=1): ...``.  Restores the previous
    values on exit.  For tests/ and tools/; product code never sets them."""

    def __init__(self, **options):
        self.options = options
        self.saved = {}

    def __enter__(self):
        lib = load()
        for name, value in self.options.items():
            key = name.encode()
            old = lib.oovqe_debug_get_option(key)
            if old < 0:
                raise OovqeError(f"unknown debug option {name!r}")
            self.saved[name] = old
            check(lib.oovqe_debug_set_option(key, int(value)), "oovqe_debug_set_option")
        return self

    def __exit__(self, *exc):
        lib = load()
        for name, value in self.saved.items():
            lib.oovqe_debug_set_option(name.encode(), value)
        return False


def check(rc, what):
    if rc != 0:
        msg = load().oovqe_last_error().decode("utf-8", "replace")
        raise OovqeError(f"{what} failed (rc={rc}): {msg}")


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_device():
    """Fail loudly when no HIP device is usable (product code never falls back to CPU)."""
    if not torch.cuda.is_available():
        raise OovqeError("auto_oo_amd needs a HIP device (MI355X); none is visible and there is "
                         "no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def dptr(t, dtype=torch.float64):
    """Device pointer of a contiguous CUDA tensor of the expected dtype."""
    if t is None:
        return None
    if not t.is_cuda:
        raise OovqeError("expected a CUDA(HIP) tensor")
    if t.dtype != dtype:
        raise OovqeError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise OovqeError("expected a contiguous tensor")
    return ctypes.c_void_p(t.data_ptr())
