"""ctypes binding of include/gbp_ba.h (libgbp_hip.so).  Loading fails loudly: there is no CPU path."""
from __future__ import annotations

import ctypes as ct
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GBP_HIP_LIB', os.path.join(_HERE, 'libgbp_hip.so'))   # override: A/B builds of the same ABI

_dp = ct.POINTER(ct.c_double)
_ip = ct.POINTER(ct.c_int32)
_bp = ct.POINTER(ct.c_uint8)

LOSS = {None: 0, 'None': 0, 'none': 0, 'huber': 1, 'constant': 2}
ABI_VERSION = 3
FLAG_NO_FUSED = 1
FLAG_DEVICE_INPUT = 2
FLAG_FORCE_FUSED = 4
PLAN_INFO_FIELDS = 11
CAM_PARTIAL_DOUBLES = 27
COMM_ID_BYTES = 128
XCH_ALWAYS = 1
PEER_HANDLE_BYTES = 64
PEER_SAME_PROCESS = 1
PEER_RENDEZVOUS = 2
PEER_MAX_RANKS = 16
EXCHANGE_FN = ct.CFUNCTYPE(ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_uint64, ct.c_void_p)


def torch_rccl_path():
    """PyTorch's bundled librccl.so when torch is installed (the build that matches the HIP runtime torch maps), else None
    (the library then takes whatever librccl the process holds, or the system one)."""
    import importlib.util
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin or os.environ.get('GBP_SYSTEM_HIP'):
        return None
    path = os.path.join(os.path.dirname(spec.origin), 'lib', 'librccl.so')
    return path.encode() if os.path.exists(path) else None


class Desc(ct.Structure):
    _fields_ = [('n_cams', ct.c_int32), ('n_lmks', ct.c_int32), ('n_factors', ct.c_int32), ('device', ct.c_int32),
                ('K', ct.c_double * 4),
                ('cam_means', _dp), ('lmk_means', _dp), ('meas', _dp), ('cam_idx', _ip), ('lmk_idx', _ip),
                ('gauss_noise_std', ct.c_double),
                ('loss', ct.c_int32), ('num_undamped_iters', ct.c_int32), ('min_linear_iters', ct.c_int32),
                ('flags', ct.c_int32),
                ('nstds', ct.c_double), ('beta', ct.c_double), ('eta_damping', ct.c_double)]


class GbpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libgbp_hip error {code}: {msg}")
        self.code = code


# every symbol include/gbp_ba.h declares: (restype, argtypes)
SIGNATURES = {
    'gbp_abi_version': (ct.c_int, []),
    'gbp_last_error': (ct.c_char_p, []),
    'gbp_ba_create': (ct.c_int, [ct.POINTER(ct.c_void_p), ct.POINTER(Desc)]),
    'gbp_ba_destroy': (None, [ct.c_void_p]),
    'gbp_ba_set_stream': (ct.c_int, [ct.c_void_p, ct.c_void_p]),
    'gbp_ba_sync': (ct.c_int, [ct.c_void_p]),
    'gbp_ba_generate_priors': (ct.c_int, [ct.c_void_p, ct.c_double]),
    'gbp_ba_factor_lambda_max': (ct.c_int, [ct.c_void_p, _dp, _dp]),
    'gbp_ba_set_prior_scalars': (ct.c_int, [ct.c_void_p, _dp, _dp]),
    'gbp_ba_set_priors': (ct.c_int, [ct.c_void_p, _dp, _dp, _dp, _dp]),
    'gbp_ba_weaken_priors': (ct.c_int, [ct.c_void_p, ct.c_double]),
    'gbp_ba_update_beliefs': (ct.c_int, [ct.c_void_p]),
    'gbp_ba_iterate': (ct.c_int, [ct.c_void_p, ct.c_int32, ct.c_int32, ct.c_int32]),
    'gbp_ba_robustify': (ct.c_int, [ct.c_void_p]),
    'gbp_ba_relinearise': (ct.c_int, [ct.c_void_p]),
    'gbp_ba_compute_messages': (ct.c_int, [ct.c_void_p, ct.c_int32]),
    'gbp_ba_compute_factors': (ct.c_int, [ct.c_void_p]),
    'gbp_ba_are': (ct.c_int, [ct.c_void_p, _dp]),
    'gbp_ba_energy': (ct.c_int, [ct.c_void_p, _dp]),
    'gbp_ba_residual_sums': (ct.c_int, [ct.c_void_p, _dp]),
    'gbp_ba_get_beliefs': (ct.c_int, [ct.c_void_p, _dp, _dp, _dp, _dp]),
    'gbp_ba_get_means': (ct.c_int, [ct.c_void_p, _dp, _dp]),
    'gbp_ba_get_covariances': (ct.c_int, [ct.c_void_p, _dp, _dp]),
    'gbp_ba_get_priors': (ct.c_int, [ct.c_void_p, _dp, _dp, _dp, _dp]),
    'gbp_ba_get_messages': (ct.c_int, [ct.c_void_p, ct.c_int32, ct.c_int32, _dp, _dp, _dp, _dp]),
    'gbp_ba_get_factors': (ct.c_int, [ct.c_void_p, ct.c_int32, ct.c_int32, _dp, _dp, _dp, _ip, _ip, _dp]),
    'gbp_ba_get_relin_state': (ct.c_int, [ct.c_void_p, _ip, _dp, _dp, _bp]),
    'gbp_ba_get_relin_state_range': (ct.c_int, [ct.c_void_p, ct.c_int32, ct.c_int32, _ip, _dp, _dp, _bp]),
    'gbp_ba_count_relinearising': (ct.c_int, [ct.c_void_p, ct.POINTER(ct.c_int64)]),
    'gbp_ba_get_relin_counts': (ct.c_int, [ct.c_void_p, _ip, ct.c_int32]),
    'gbp_ba_eval_fn': (ct.c_int, [_dp, ct.c_int32, _dp, _dp, _dp, _dp, ct.c_int32]),
    'gbp_ba_get_kernel_times': (ct.c_int, [ct.c_void_p, _dp, ct.c_int32, ct.POINTER(ct.c_int32)]),
    'gbp_ba_get_sweep_clocks': (ct.c_int, [ct.c_void_p, _dp, ct.c_int32, ct.POINTER(ct.c_int32)]),
    'gbp_ba_comm_info': (ct.c_int, [ct.c_void_p, ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32)]),
    'gbp_ba_set_iters_since_relin': (ct.c_int, [ct.c_void_p, _ip]),
    'gbp_ba_fill_iters_since_relin': (ct.c_int, [ct.c_void_p, ct.c_int32]),
    'gbp_ba_shard_begin': (ct.c_int, [ct.c_void_p, ct.c_int32, ct.c_int32, ct.c_int32, ct.c_void_p]),
    'gbp_ba_shard_end': (ct.c_int, [ct.c_void_p, ct.c_void_p, ct.c_int32]),
    'gbp_ba_comm_unique_id': (ct.c_int, [ct.c_void_p, ct.c_char_p]),
    'gbp_ba_comm_init_rccl': (ct.c_int, [ct.c_void_p, ct.c_void_p, ct.c_int32, ct.c_int32, ct.c_int32, ct.c_char_p]),
    'gbp_ba_set_exchange': (ct.c_int, [ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int32, ct.c_int32, ct.c_int32]),
    'gbp_ba_comm_destroy': (ct.c_int, [ct.c_void_p]),
    'gbp_ba_peer_export': (ct.c_int, [ct.c_void_p, ct.c_int32, ct.c_void_p, ct.c_int32]),
    'gbp_ba_peer_connect': (ct.c_int, [ct.c_void_p, ct.c_int32, ct.c_int32, ct.c_void_p, ct.c_int32]),
    'gbp_ba_peer_selftest': (ct.c_int, [ct.c_void_p, ct.c_int32]),
    'gbp_ba_iterate_sharded': (ct.c_int, [ct.c_void_p, ct.c_int32, ct.c_int32, ct.c_int32]),
    'gbp_ba_update_beliefs_sharded': (ct.c_int, [ct.c_void_p]),
    'gbp_ba_set_kernel_timing': (ct.c_int, [ct.c_void_p, ct.c_int32]),
    'gbp_ba_get_kernel_timing': (ct.c_int, [ct.c_void_p, _dp, ct.POINTER(ct.c_int32), ct.POINTER(ct.c_char_p)]),
    'gbp_ba_means_snapshot': (ct.c_int, [ct.c_void_p]),
    'gbp_ba_means_fetch': (ct.c_int, [ct.c_void_p, _dp, _dp, ct.c_int32]),
    'gbp_bal_header': (ct.c_int, [ct.c_char_p, ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32)]),
    'gbp_bal_read': (ct.c_int, [ct.c_char_p, ct.c_int32, ct.c_int32, ct.c_int32, _dp, _dp, _dp, _dp, _ip, _ip]),
    'gbp_ba_state_size': (ct.c_int, [ct.c_void_p, ct.POINTER(ct.c_uint64)]),
    'gbp_ba_save_state': (ct.c_int, [ct.c_void_p, ct.c_void_p, ct.c_uint64]),
    'gbp_ba_load_state': (ct.c_int, [ct.c_void_p, ct.c_void_p, ct.c_uint64]),
    'gbp_ba_snapshot_state': (ct.c_int, [ct.c_void_p]),
    'gbp_ba_restore_snapshot': (ct.c_int, [ct.c_void_p]),
    'gbp_lin_create': (ct.c_int, [ct.POINTER(ct.c_void_p), ct.c_void_p]),
    'gbp_lin_destroy': (None, [ct.c_void_p]),
    'gbp_lin_sync': (ct.c_int, [ct.c_void_p]),
    'gbp_lin_update_beliefs': (ct.c_int, [ct.c_void_p]),
    'gbp_lin_iterate': (ct.c_int, [ct.c_void_p, ct.c_int32]),
    'gbp_lin_energy': (ct.c_int, [ct.c_void_p, _dp]),
    'gbp_lin_get_beliefs': (ct.c_int, [ct.c_void_p, _dp, _dp]),
    'gbp_lin_get_means': (ct.c_int, [ct.c_void_p, _dp]),
    'gbp_lin_get_messages': (ct.c_int, [ct.c_void_p, _dp, _dp, _dp, _dp]),
    'gbp_ba_fused_max_cams': (ct.c_int, []),
    'gbp_ba_plan_info': (ct.c_int, [ct.c_void_p, _ip, ct.c_int32]),
    'gbp_ba_phase_profile': (ct.c_int, [ct.c_void_p, ct.c_void_p, ct.c_int32, ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32)]),
    'gbp_ba_check_layout': (ct.c_int, [ct.c_void_p, ct.POINTER(ct.c_int32)]),
    'gbp_ba_info': (ct.c_int, [ct.c_void_p, ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32)]),
}



class LinDesc(ct.Structure):
    """gbp_lin_desc_t (include/gbp_lin.h)."""
    _fields_ = [('n_vars', ct.c_int32), ('dofs', ct.c_int32), ('n_factors', ct.c_int32), ('device', ct.c_int32),
                ('var_a', _ip), ('var_b', _ip), ('factor_eta', _dp), ('factor_lam', _dp), ('factor_const', _dp),
                ('prior_eta', _dp), ('prior_lam', _dp), ('eta_damping', ct.c_double)]


_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7; libgbp_hip.so links the
    system one, which has the SAME soname, so whichever is mapped first serves both -- and torch cannot initialise on a
    newer system runtime ("No HIP GPUs are available") when this library happened to be loaded before `import torch`
    (the sharded driver needs both in one process).  If torch is installed, map ITS runtime first; the kernels here only
    need the plain HIP API and run on either.  GBP_SYSTEM_HIP=1 keeps the system runtime."""
    import importlib.util
    import sys
    if 'torch' in sys.modules or os.environ.get('GBP_SYSTEM_HIP'):
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if os.path.exists(path):
        try:
            ct.CDLL(path, mode=ct.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen libgbp_hip.so and bind every declared symbol (AttributeError if one is missing)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -m gbp_amd.build` "
                              f"(hipcc --offload-arch=gfx950).  gbp_amd has no CPU fallback.")
        _share_torch_hip_runtime()
        lib = ct.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.gbp_abi_version() != ABI_VERSION:
            raise ImportError(f"libgbp_hip.so ABI {lib.gbp_abi_version()} != {ABI_VERSION}")
        _lib = lib
    return _lib


def available():
    """True when libgbp_hip.so exists and loads (host-only helpers such as the BAL reader work without a GPU)."""
    try:
        load()
        return True
    except (ImportError, OSError, AttributeError):
        return False


def check(rc):
    if rc != 0:
        raise GbpError(rc, load().gbp_last_error().decode('utf-8', 'replace'))


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def iptr(a):
    return None if a is None else a.ctypes.data_as(_ip)


def bptr(a):
    return None if a is None else a.ctypes.data_as(_bp)


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def i32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.int32)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a
