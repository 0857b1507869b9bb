"""ctypes binding of include/ndt_hip.h (the C-ABI shared library libndt_hip.so).

The library is built in-tree by ``gtsam_ndt_amd.build.build_all()`` (hipcc, gfx950) and
travels to the GPU box with the repo snapshot.  Loading fails loudly when it is missing:
there is no CPU or PyTorch fallback for the matcher.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NDT_HIP_LIB: an instrumented build of the same library (tools/ only)
LIB_PATH = os.environ.get("NDT_HIP_LIB") or os.path.join(_HERE, "lib", "libndt_hip.so")

NDT_OK = 0
NDT_NOT_CONVERGED = 1
NDT_DEGENERATE_HESSIAN = 2
NDT_TOO_FEW_HITS = 3
NDT_TOO_FEW_CELLS = 4
NDT_ERR_INVALID_ARG = -1
NDT_ERR_NO_TARGET = -2
NDT_ERR_HIP = -3
NDT_ERR_NO_DEVICE = -4
NDT_ERR_CAPACITY = -5
NDT_ERR_ALLOC = -6
NDT_ERR_RCCL = -7

# ndt2d_set_tuning / ndt2d_batch_set_tuning knobs
TUNING = {"launch_graphs": 1, "wide_threshold": 2, "short_scan_kernel": 3, "chunk_launches": 4, "binned_build": 5,
          "batch_small_variant": 6, "split_from": 8, "single_sync_build": 9,
          "batch_global_workgroups": 10}

HESSIAN_GAUSS_NEWTON = 0
HESSIAN_NEWTON = 1


class Params2D(C.Structure):
    _fields_ = [
        ("cell_size", C.c_double),
        ("min_points", C.c_int32),
        ("hessian_mode", C.c_int32),
        ("eig_ratio", C.c_double),
        ("d1", C.c_double),
        ("d2", C.c_double),
        ("max_iterations", C.c_int32),
        ("fixed_iterations", C.c_int32),
        ("eps_trans", C.c_double),
        ("eps_rot", C.c_double),
        ("step_max_trans", C.c_double),
        ("step_max_rot", C.c_double),
        ("min_hits", C.c_int32),
        ("overlap_grids", C.c_int32),
        ("line_search", C.c_int32),
        ("reserved", C.c_int32),
        ("step_scale", C.c_double),
    ]


class Result2D(C.Structure):
    _fields_ = [
        ("pose", C.c_double * 3),
        ("H", C.c_double * 9),
        ("g", C.c_double * 3),
        ("score", C.c_double),
        ("iterations", C.c_int32),
        ("n_hit", C.c_int32),
        ("status", C.c_int32),
        ("reserved", C.c_int32),
    ]


class Eval2D(C.Structure):
    _fields_ = [
        ("H", C.c_double * 9),
        ("g", C.c_double * 3),
        ("score", C.c_double),
        ("n_hit", C.c_int32),
        ("reserved", C.c_int32),
    ]


class GridInfo2D(C.Structure):
    _fields_ = [
        ("ox", C.c_float), ("oy", C.c_float), ("inv_cell", C.c_float), ("cell", C.c_float),
        ("width", C.c_int32), ("height", C.c_int32), ("n_valid", C.c_int32), ("n_points", C.c_int32),
    ]


class MapHeader(C.Structure):
    """ndt_map_header of include/ndt_hip.h (104 bytes): what ndt2d_save_map / ndt3d_save_map write first."""
    _fields_ = [("magic", C.c_uint32), ("version", C.c_uint32), ("dims", C.c_int32), ("ngrid", C.c_int32),
                ("width", C.c_int32), ("height", C.c_int32), ("depth", C.c_int32), ("cell_bytes", C.c_uint32),
                ("cell_size", C.c_double), ("n_cells", C.c_uint64), ("n_points", C.c_uint64),
                ("origin", (C.c_float * 3) * 4)]


MAP_MAGIC = 0x4d54444e


class Result3D(C.Structure):
    _fields_ = [("pose", C.c_double * 6), ("H", C.c_double * 36), ("g", C.c_double * 6), ("score", C.c_double),
                ("iterations", C.c_int32), ("n_hit", C.c_int32), ("status", C.c_int32), ("reserved", C.c_int32)]


class Eval3D(C.Structure):
    _fields_ = [("H", C.c_double * 36), ("g", C.c_double * 6), ("score", C.c_double), ("n_hit", C.c_int32),
                ("reserved", C.c_int32)]


class GridInfo3D(C.Structure):
    _fields_ = [("ox", C.c_float), ("oy", C.c_float), ("oz", C.c_float), ("inv_cell", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32), ("depth", C.c_int32), ("n_valid", C.c_int32)]


_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); the CPU test checks every one of these is exported and that
# the list matches the declarations in include/ndt_hip.h.
SIGNATURES = {
    "ndt_abi_version": (C.c_int32, []),
    "ndt_status_string": (C.c_char_p, [C.c_int32]),
    "ndt_last_error": (C.c_char_p, []),
    "ndt_device_count": (C.c_int32, []),
    "ndt_set_host_wait": (C.c_int32, [C.c_int32]),
    "ndt2d_default_params": (None, [C.POINTER(Params2D)]),
    "ndt2d_create": (C.c_int32, [C.POINTER(Params2D), C.c_int32, C.POINTER(_vp)]),
    "ndt2d_destroy": (C.c_int32, [_vp]),
    "ndt2d_set_target": (C.c_int32, [_vp, _vp, _vp, C.c_size_t]),
    "ndt2d_set_target_dev": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "ndt2d_reserve_target": (C.c_int32, [_vp, C.c_double, C.c_double, C.c_double, C.c_double]),
    "ndt2d_add_target_points": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ndt2d_add_target_points_dev": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, _vp, C.POINTER(C.c_size_t), _vp]),
    "ndt2d_get_grid_info": (C.c_int32, [_vp, C.POINTER(GridInfo2D)]),
    "ndt2d_get_grid": (C.c_int32, [_vp, _vp, _vp, _vp]),
    "ndt2d_map_size": (C.c_size_t, [_vp]),
    "ndt2d_save_map": (C.c_int32, [_vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ndt2d_load_map": (C.c_int32, [_vp, _vp, C.c_size_t]),
    "ndt2d_evaluate": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, _dp, C.POINTER(Eval2D)]),
    "ndt2d_evaluate_dev": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, _dp, C.POINTER(Eval2D)]),
    "ndt2d_align": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, _dp, C.POINTER(Result2D)]),
    "ndt2d_align_dev": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, _dp, C.POINTER(Result2D)]),
    "ndt2d_align_dev_async": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, _dp]),
    "ndt2d_align_finish": (C.c_int32, [_vp, C.POINTER(Result2D)]),
    "ndt2d_align_trace": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, _dp, _vp, C.c_int32, C.POINTER(C.c_int32), _vp]),
    "ndt2d_align_multi_scan_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, C.c_int32, _vp]),
    "ndt2d_align_multi_start_dev": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, _vp, C.c_int32, _vp]),
    "ndt2d_stream": (_vp, [_vp]),
    "ndt2d_set_tuning": (C.c_int32, [_vp, C.c_int32, C.c_int64]),
    "ndt2d_batch_set_tuning": (C.c_int32, [_vp, C.c_int32, C.c_int64]),
    "ndt2d_wait_stream": (C.c_int32, [_vp, _vp]),
    "ndt2d_batch_wait_stream": (C.c_int32, [_vp, _vp]),
    "ndt3d_wait_stream": (C.c_int32, [_vp, _vp]),
    "ndt3d_set_tuning": (C.c_int32, [_vp, C.c_int32, C.c_int64]),
    "ndt2d_calibrated_covariance": (C.c_int32, [_dp, C.c_int32, _dp]),
    "ndt_magnusson_constants": (C.c_int32, [C.c_double, C.c_double, C.c_int32, _dp, _dp]),
    "ndt2d_polar_to_points_dev": (C.c_int32, [_vp, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double, _vp, _vp, _vp]),
    "ndt2d_batch_create": (C.c_int32, [C.POINTER(Params2D), C.c_int32, C.POINTER(_vp)]),
    "ndt2d_batch_destroy": (C.c_int32, [_vp]),
    "ndt2d_batch_align": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "ndt2d_batch_align_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp]),
    "ndt2d_batch_stream": (_vp, [_vp]),
    "ndt2d_batch_last_large_count": (C.c_int64, [_vp]),
    "ndt2d_default_pyramid": (C.c_int32, [C.POINTER(Params2D), C.POINTER(Params2D)]),
    "ndt2d_batch_create_pyramid": (C.c_int32, [C.POINTER(Params2D), C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "ndt2d_multi_create_pyramid": (C.c_int32, [C.POINTER(Params2D), C.c_int32, _vp, C.c_int32, C.POINTER(_vp)]),
    "ndt2d_multi_create": (C.c_int32, [C.POINTER(Params2D), _vp, C.c_int32, C.POINTER(_vp)]),
    "ndt2d_multi_destroy": (C.c_int32, [_vp]),
    "ndt2d_multi_device_count": (C.c_int32, [_vp]),
    "ndt2d_multi_align": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "ndt2d_multi_align_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ndt2d_multi_plan": (C.c_int32, [C.c_int32, _vp, _vp, C.c_size_t, C.c_int32, _vp]),
    "ndt2d_multi_plan_hinted": (C.c_int32, [C.c_int32, _vp, _vp, C.c_size_t, C.c_int32, _vp, _vp]),
    "ndt3d_default_params": (None, [C.POINTER(Params2D)]),
    "ndt3d_create": (C.c_int32, [C.POINTER(Params2D), C.c_int32, C.POINTER(_vp)]),
    "ndt3d_destroy": (C.c_int32, [_vp]),
    "ndt3d_set_target": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_size_t]),
    "ndt3d_add_target_points": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ndt3d_set_target_dev": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "ndt3d_reserve_target": (C.c_int32, [_vp, _dp, _dp]),
    "ndt3d_range_image_to_points_dev": (C.c_int32, [_vp, C.c_int32, C.c_int32, _dp, C.c_double, C.c_double, C.c_double,
                                                    C.c_double, _vp, _vp, _vp, _vp]),
    "ndt3d_add_target_points_dev": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_size_t, _dp, C.POINTER(C.c_size_t), _vp]),
    "ndt3d_get_grid_info": (C.c_int32, [_vp, C.POINTER(GridInfo3D)]),
    "ndt3d_get_grid": (C.c_int32, [_vp, _vp, _vp, _vp]),
    "ndt3d_map_size": (C.c_size_t, [_vp]),
    "ndt3d_save_map": (C.c_int32, [_vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ndt3d_load_map": (C.c_int32, [_vp, _vp, C.c_size_t]),
    "ndt3d_evaluate": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_size_t, _dp, C.POINTER(Eval3D)]),
    "ndt3d_evaluate_dev": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_size_t, _dp, C.POINTER(Eval3D)]),
    "ndt3d_align": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_size_t, _dp, C.POINTER(Result3D)]),
    "ndt3d_align_dev_async": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_size_t, _dp]),
    "ndt3d_align_finish": (C.c_int32, [_vp, C.POINTER(Result3D)]),
    "ndt3d_stream": (_vp, [_vp]),
    "ndt3d_align_multi_scan_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _dp, C.c_int32, _vp]),
    "ndt3d_align_multi_start_dev": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_size_t, _dp, C.c_int32, _vp]),
    "ndt3d_align_trace": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_size_t, _dp, _vp, C.c_int32, C.POINTER(C.c_int32), _vp]),
    "ndt3d_align_dev": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_size_t, _dp, C.POINTER(Result3D)]),
    "ndt3d_batch_create": (C.c_int32, [C.POINTER(Params2D), C.c_int32, C.POINTER(_vp)]),
    "ndt3d_batch_create_pyramid": (C.c_int32, [C.POINTER(Params2D), C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "ndt3d_batch_destroy": (C.c_int32, [_vp]),
    "ndt3d_batch_align": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "ndt3d_batch_align_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp]),
    "ndt3d_batch_stream": (_vp, [_vp]),
    "ndt3d_batch_wait_stream": (C.c_int32, [_vp, _vp]),
    "ndt3d_batch_set_tuning": (C.c_int32, [_vp, C.c_int32, C.c_int64]),
    "ndt3d_multi_create": (C.c_int32, [C.POINTER(Params2D), _vp, C.c_int32, C.POINTER(_vp)]),
    "ndt3d_multi_create_pyramid": (C.c_int32, [C.POINTER(Params2D), C.c_int32, _vp, C.c_int32, C.POINTER(_vp)]),
    "ndt3d_multi_destroy": (C.c_int32, [_vp]),
    "ndt3d_multi_device_count": (C.c_int32, [_vp]),
    "ndt3d_multi_align": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "ndt3d_multi_align_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_size_t), _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load libndt_hip.so (once) and attach prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The NDT matcher has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.  If this library
    # pulled in the system copy first, a later `import torch` would find "No HIP GPUs", so
    # when torch is installed let it load its runtime before we resolve ours against it.
    import importlib.util
    if importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class NdtError(RuntimeError):
    def __init__(self, code: int, where: str):
        lib = load()
        msg = lib.ndt_status_string(code).decode()
        detail = lib.ndt_last_error().decode()
        super().__init__(f"{where}: {msg} ({code})" + (f": {detail}" if detail else ""))
        self.code = code


def check(code: int, where: str) -> int:
    if code < 0:
        raise NdtError(code, where)
    return code
