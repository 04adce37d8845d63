"""ctypes binding of libhalo2_mi355x.so (include/halo2_mi355x.h).  No torch types cross the ABI."""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# eight commitments in flight need more than the HIP runtime's default of 4 hardware queues (capi.hip sets the same
# default when the library is loaded; here it is set as early as the package import, before torch touches the GPU)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
LIB_PATH = os.environ.get("HALO2_MI355X_LIB") or os.path.join(CSRC, "libhalo2_mi355x.so")   # override: A/B builds
HOSTCHECK_PATH = os.path.join(CSRC, "libhm_hostcheck.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "halo2_mi355x.h")

_u64p = ctypes.POINTER(ctypes.c_uint64)
_vp = ctypes.c_void_p
NO_CHAIN = ctypes.c_size_t(-1).value      # HM_NO_CHAIN of the header


class Halo2Mi355xError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libhalo2_mi355x error {code}: {message}")
        self.code = code


class MsmStats(ctypes.Structure):
    _fields_ = [("digits_ms", ctypes.c_double), ("sort_ms", ctypes.c_double), ("accumulate_ms", ctypes.c_double),
                ("reduce_ms", ctypes.c_double), ("total_ms", ctypes.c_double), ("accumulate_kernel_ms", ctypes.c_double),
                ("pairs", ctypes.c_uint64),
                ("tasks", ctypes.c_uint64), ("window_bits", ctypes.c_uint32), ("windows", ctypes.c_uint32)]


class Stats(ctypes.Structure):
    """hm_stats: per-call counters since start / hm_reset_stats (include/halo2_mi355x.h)."""
    _fields_ = [("msm_calls", ctypes.c_uint64), ("msm_points", ctypes.c_uint64), ("ntt_calls", ctypes.c_uint64),
                ("ntt_elements", ctypes.c_uint64), ("msm_calls_by_log2", ctypes.c_uint64 * 32),
                ("ntt_calls_by_log2", ctypes.c_uint64 * 32), ("msm_h2d_us", ctypes.c_double), ("msm_device_us", ctypes.c_double),
                ("msm_host_us", ctypes.c_double), ("ntt_h2d_us", ctypes.c_double), ("ntt_device_us", ctypes.c_double),
                ("ntt_d2h_us", ctypes.c_double), ("h2d_bytes", ctypes.c_uint64), ("d2h_bytes", ctypes.c_uint64),
                ("vector_calls", ctypes.c_uint64 * 8), ("vector_elements", ctypes.c_uint64 * 8),
                ("coset_table_bytes", ctypes.c_uint64), ("coset_tables", ctypes.c_uint64),
                ("ntt_table_bytes", ctypes.c_uint64), ("ntt_tables", ctypes.c_uint64),
                ("host_copies_direct", ctypes.c_uint64), ("host_copies_staged", ctypes.c_uint64), ("host_ranges_registered", ctypes.c_uint64)]
    KINDS = ("eval_polynomial", "graph_evaluate", "kate_division", "grand_product", "batch_invert", "linear_combination", "lookup_permute")


class BasesInfo(ctypes.Structure):
    """hm_bases_info (include/halo2_mi355x.h)."""
    _fields_ = [("n", ctypes.c_uint64), ("device_bytes", ctypes.c_uint64), ("parked_bytes", ctypes.c_uint64),
                ("default_tables_dropped", ctypes.c_uint64), ("table_windows", ctypes.c_uint32), ("table_window_bits", ctypes.c_uint32),
                ("devices", ctypes.c_uint32), ("sliced", ctypes.c_uint32)]


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into csrc/libhalo2_mi355x.so (in-tree) with hipcc."""
    args = ["make", "-C", CSRC, "-j4"]
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True)
    subprocess.run(args, check=True)
    return LIB_PATH


_SIGNATURES = {
    "hm_device_count": (ctypes.c_int, []),
    "hm_set_device": (ctypes.c_int, [ctypes.c_int]),
    "hm_shutdown": (ctypes.c_int, []),
    "hm_last_error": (ctypes.c_char_p, []),
    "hm_version": (ctypes.c_char_p, []),
    "hm_msm_bn256_g1": (ctypes.c_int, [_u64p, _u64p, ctypes.c_size_t, _u64p, ctypes.POINTER(ctypes.c_int)]),
    "hm_msm_bn256_g1_jacobian": (ctypes.c_int, [_u64p, _u64p, ctypes.c_size_t, _u64p]),
    "hm_register_bases": (ctypes.c_int, [_u64p, ctypes.c_size_t, _u64p]),
    "hm_release_bases": (ctypes.c_int, [ctypes.c_uint64]),
    "hm_msm_bn256_g1_h": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_size_t, _u64p, ctypes.c_size_t, _u64p,
                                          ctypes.POINTER(ctypes.c_int)]),
    "hm_register_bases_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _u64p]),
    "hm_register_bases_precomp": (ctypes.c_int, [_u64p, ctypes.c_size_t, _u64p]),
    "hm_register_bases_precomp_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _u64p]),
    "hm_register_bases_plain": (ctypes.c_int, [_u64p, ctypes.c_size_t, _u64p]),
    "hm_register_bases_plain_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _u64p]),
    "hm_get_bases_info": (ctypes.c_int, [ctypes.c_uint64, ctypes.POINTER(BasesInfo)]),
    "hm_msm_bn256_g1_dev": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_size_t, _vp, ctypes.c_size_t, _vp, _u64p]),
    "hm_msm_submit_dev": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_size_t, _vp, ctypes.c_size_t, _vp, _u64p]),
    "hm_msm_wait": (ctypes.c_int, [ctypes.c_uint64, _u64p]),
    "hm_msm_batch_bn256_g1_dev": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t,
                                                 ctypes.c_size_t, _vp, _u64p]),
    "hm_msm_batch_bn256_g1_h": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t,
                                              ctypes.c_size_t, _u64p]),
    "hm_g1_sum": (ctypes.c_int, [_u64p, ctypes.c_size_t, _u64p]),
    "hm_coeff_to_extended_bn256_fr_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, _u64p, ctypes.c_uint32,
                                                         ctypes.c_uint32, _u64p, ctypes.c_void_p]),
    "hm_coeff_to_coset_bn256_fr_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _u64p, ctypes.c_uint32, _u64p, ctypes.c_int, _vp]),
    "hm_coset_to_coeff_bn256_fr_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, _u64p, ctypes.c_uint32, _u64p, _u64p, _vp]),
    "hm_coeff_to_cosets_bn256_fr_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _u64p, ctypes.c_uint32, _u64p, ctypes.c_size_t, ctypes.c_int, _vp]),
    "hm_cosets_to_coeff_bn256_fr_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, _u64p, ctypes.c_uint32, _u64p, _u64p, _vp]),
    "hm_msm_set_window": (ctypes.c_int, [ctypes.c_int]),
    "hm_msm_set_phase_timing": (ctypes.c_int, [ctypes.c_int]),
    "hm_set_host_base_cache": (ctypes.c_int, [ctypes.c_int]),
    "hm_set_fixed_base_threshold": (ctypes.c_int, [ctypes.c_uint32]),
    "hm_set_msm_devices": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    "hm_ntt_bn256_fr": (ctypes.c_int, [_u64p, _u64p, ctypes.c_uint32]),
    "hm_set_host_copies": (ctypes.c_int, [ctypes.c_int]),
    "hm_host_register": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    "hm_host_unregister": (ctypes.c_int, [ctypes.c_void_p]),
    "hm_device_malloc": (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    "hm_device_free": (ctypes.c_int, [ctypes.c_void_p]),
    "hm_copy_to_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "hm_copy_to_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "hm_copy_many_to_device": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_size_t]),
    "hm_copy_many_to_host": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_size_t]),
    "hm_device_synchronize": (ctypes.c_int, []),
    "hm_coeff_to_extended_bn256_fr": (ctypes.c_int, [_u64p, _u64p, _u64p, ctypes.c_uint32, ctypes.c_uint32, _u64p]),
    "hm_extended_to_coeff_bn256_fr": (ctypes.c_int, [_u64p, ctypes.c_size_t, _u64p, ctypes.c_uint32, _u64p, _u64p]),
    "hm_ntt_bn256_fr_dev": (ctypes.c_int, [_vp, _u64p, ctypes.c_uint32, _vp]),
    "hm_ntt_batch_bn256_fr_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, _u64p, ctypes.c_uint32, _u64p, _u64p, _vp]),
    "hm_ifft_bn256_fr_dev": (ctypes.c_int, [_vp, _u64p, ctypes.c_uint32, _u64p, _vp]),
    "hm_coset_ntt_bn256_fr_dev": (ctypes.c_int, [_vp, _u64p, ctypes.c_uint32, _u64p, _vp]),
    "hm_graph_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t, _u64p, ctypes.c_size_t, ctypes.c_size_t,
                                       ctypes.POINTER(ctypes.c_int32), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint32, _u64p]),
    "hm_graph_evaluate_dev": (ctypes.c_int, [ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, _u64p, ctypes.c_size_t,
                                             ctypes.c_uint32, _vp, _vp]),
    "hm_graph_evaluate_flags_dev": (ctypes.c_int, [ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, _u64p, ctypes.c_size_t,
                                                   ctypes.c_uint32, _vp, ctypes.c_uint32, _vp]),
    "hm_graph_evaluate_segments_dev": (ctypes.c_int, [ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, _u64p, ctypes.c_size_t,
                                                      ctypes.c_uint32, ctypes.c_uint32, _vp, ctypes.c_uint32, _vp]),
    "hm_quotient_by_cosets_bn256_fr_dev": (ctypes.c_int, [ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, _u64p, ctypes.c_size_t,
                                                          ctypes.c_uint32, _u64p, _u64p, ctypes.c_size_t, ctypes.c_size_t, _vp, _vp]),
    "hm_quotient_partials_bn256_fr_dev": (ctypes.c_int, [ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, _u64p,
                                                         ctypes.c_size_t, ctypes.c_uint32, _u64p, _u64p, ctypes.c_size_t, _vp, _vp]),
    "hm_quotient_combine_bn256_fr_dev": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), _u64p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_size_t, _vp, _vp]),
    "hm_graph_destroy": (ctypes.c_int, [ctypes.c_uint64]),
    "hm_fr_powers_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, _u64p, _vp]),
    "hm_fr_mul_periodic_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, _u64p, ctypes.c_uint32, _vp]),
    "hm_lookup_permute_bn256_fr_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, _vp, _vp]),
    "hm_lookup_permute_batch_bn256_fr_dev": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t,
                                                          ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                                          ctypes.POINTER(ctypes.c_int), _vp]),
    "hm_kate_division_bn256_fr_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, _u64p, _vp, _vp]),
    "hm_fr_grand_product_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, _u64p, _vp, _vp]),
    "hm_kate_division_batch_bn256_fr_dev": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_size_t, _u64p, ctypes.POINTER(_vp), ctypes.c_size_t, _vp]),
    "hm_fr_grand_product_batch_dev": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_size_t, _u64p, ctypes.c_size_t, ctypes.POINTER(_vp),
                                                     ctypes.c_size_t, _vp]),
    "hm_fr_batch_invert_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, _vp]),
    "hm_fr_linear_combination_dev": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), _u64p, ctypes.c_size_t, ctypes.c_size_t, _vp, _vp]),
    "hm_fr_random_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.c_uint64, _vp]),
    "hm_fr_affine_sequence_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, _u64p, _u64p, _vp]),
    "hm_fr_dot_bn256_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _u64p, _vp]),
    "hm_fr_scale_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, _u64p, _vp]),
    "hm_fr_distribute_powers_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, _u64p, _vp]),
    "hm_g1_fixed_base_mul_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, _u64p, _vp, _vp]),
    "hm_extended_to_coeff_bn256_fr_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, _u64p, ctypes.c_uint32, _u64p, _u64p, _vp]),
    "hm_eval_polynomial_bn256_fr_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32), _u64p, ctypes.c_size_t,
                                                       _u64p, _vp]),
    "hm_get_msm_stats": (ctypes.c_int, [ctypes.POINTER(MsmStats)]),
    "hm_get_stats": (ctypes.c_int, [ctypes.POINTER(Stats)]),
    "hm_reset_stats": (ctypes.c_int, []),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load the HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the MSM/NTT path)")
        try:
            # torch bundles its own libamdhip64; if ours pulled /opt/rocm's copy in first, torch would
            # later load a second HIP runtime and see no device.  Load torch's first when it exists.
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


FI_LIB_PATH = os.path.join(CSRC, "libhalo2_mi355x_fi.so")
_fi = None


def load_fi() -> ctypes.CDLL:
    """The TEST build of the same sources with the fault points of the C-ABI barrier compiled in (csrc/Makefile:
    -DHM_FAULT_INJECTION) and hm_test_arm_fault(point, after) exported.  Used by tests/test_capi_faults.py only; a second,
    independent copy of the library in the process (its own contexts, its own handles)."""
    global _fi
    if _fi is None:
        load()                                   # torch's HIP runtime first, as for the product library
        lib = ctypes.CDLL(FI_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        lib.hm_test_arm_fault.restype = ctypes.c_int
        lib.hm_test_arm_fault.argtypes = [ctypes.c_char_p, ctypes.c_long]
        _fi = lib
    return _fi


def check(rc: int) -> None:
    if rc != 0:
        raise Halo2Mi355xError(rc, load().hm_last_error().decode())
