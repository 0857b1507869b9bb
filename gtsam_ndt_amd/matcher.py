"""Host-side mirror of the scan-matcher interface, over the C-ABI (include/ndt_hip.h).

``NdtMatcher2D`` is the Python twin of the C++ adapter ``ndt::NdtMatcherHip``
(include/ndt_matcher_hip.hpp): set a target scan/submap, align a source scan from an
initial SE(2) guess, get back pose + information matrix - the quantities a
scan-matcher -> GTSAM BetweenFactor<Pose2> bridge needs (BASELINE.json north_star; the
reference's own interface is not observable, /root/reference/README.md:1).

numpy arrays go through the host-pointer entry points; torch CUDA tensors (or any object
with ``data_ptr()``) go through the ``_dev`` entry points without a copy.  PyTorch is only
plumbing for device memory here.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L


@dataclass
class AlignResult:
    pose: tuple            # (tx, ty, theta)
    H: np.ndarray          # 3x3 Hessian of -score (information up to score scaling)
    g: np.ndarray
    score: float
    iterations: int
    n_hit: int
    status: int

    @property
    def converged(self) -> bool:
        return self.status == L.NDT_OK

    def covariance(self, hessian_mode: int = L.HESSIAN_GAUSS_NEWTON) -> np.ndarray:
        """The calibrated pose covariance a BetweenFactor noise model is built from
        (ndt2d_calibrated_covariance: S H^-1 S, factors from the Monte-Carlo calibration;
        hessian_mode = the form of H, i.e. the matcher's hessian_mode).  H^-1 alone underestimates
        the scatter of the estimate 3 to 4 times in standard deviation."""
        H = np.ascontiguousarray(self.H, dtype=np.float64)
        cov = np.zeros(9, dtype=np.float64)
        st = L.load().ndt2d_calibrated_covariance(H.ctypes.data_as(C.POINTER(C.c_double)), int(hessian_mode),
                                                  cov.ctypes.data_as(C.POINTER(C.c_double)))
        if st != L.NDT_OK:
            raise np.linalg.LinAlgError("Hessian is not positive definite")
        return cov.reshape(3, 3)


def default_params(**overrides) -> L.Params2D:
    p = L.Params2D()
    L.load().ndt2d_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(f"ndt2d_params has no field {k!r}")
        setattr(p, k, v)
    return p


def _is_dev(a) -> bool:
    return hasattr(a, "data_ptr")


def _host_f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _dev_ptr(t, n: int):
    import torch
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n):
        raise ValueError("device arrays must be contiguous float32 CUDA tensors of equal length")
    return C.c_void_p(t.data_ptr())


def _to_result(r: L.Result2D) -> AlignResult:
    return AlignResult(tuple(r.pose), np.array(r.H, dtype=np.float64).reshape(3, 3),
                       np.array(r.g, dtype=np.float64), float(r.score), int(r.iterations),
                       int(r.n_hit), int(r.status))


class NdtMatcher2D:
    """One handle = one device stream + one cached target grid."""

    def __init__(self, device: int = 0, params: L.Params2D | None = None, tuning: dict | None = None, **overrides):
        """tuning: execution-strategy knobs by name (L.TUNING), e.g. {"short_scan_kernel": 0}."""
        self._lib = L.load()
        self.params = params if params is not None else default_params(**overrides)
        if params is not None:
            for k, v in overrides.items():
                setattr(self.params, k, v)
        h = C.c_void_p()
        L.check(self._lib.ndt2d_create(C.byref(self.params), int(device), C.byref(h)), "ndt2d_create")
        self._h = h
        self._keep = None
        for k, v in (tuning or {}).items():
            self.set_tuning(k, v)

    def set_tuning(self, knob: str, value: int):
        L.check(self._lib.ndt2d_set_tuning(self._h, L.TUNING[knob], int(value)), "ndt2d_set_tuning")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ndt2d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- (i) target grid
    def set_target(self, x, y):
        if _is_dev(x):
            import torch
            n = x.numel()
            st = self._lib.ndt2d_set_target_dev(self._h, _dev_ptr(x, n), _dev_ptr(y, n), n,
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream))
        else:
            x, y = _host_f32(x), _host_f32(y)
            if x.shape != y.shape or x.ndim != 1:
                raise ValueError("x and y must be 1-D arrays of equal length")
            st = self._lib.ndt2d_set_target(self._h, x.ctypes.data, y.ctypes.data, x.size)
        L.check(st, "ndt2d_set_target")
        return self.grid_info()

    def reserve_target(self, xmin: float, ymin: float, xmax: float, ymax: float):
        """Empty grid over a chosen extent; fill it with add_target_points()."""
        L.check(self._lib.ndt2d_reserve_target(self._h, float(xmin), float(ymin), float(xmax), float(ymax)),
                "ndt2d_reserve_target")
        return self.grid_info()

    def add_target_points(self, x, y, pose=None) -> int:
        """Merge more points into the cached grid; returns how many fell outside its extent.
        Device tensors may carry a pose (tx, ty, theta) that moves them into the map frame first."""
        out = C.c_size_t(0)
        if _is_dev(x):
            import torch
            n = x.numel()
            p = (C.c_double * 3)(*[float(v) for v in pose]) if pose is not None else None
            L.check(self._lib.ndt2d_add_target_points_dev(self._h, _dev_ptr(x, n), _dev_ptr(y, n), n, p, C.byref(out),
                                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                    "ndt2d_add_target_points_dev")
            return int(out.value)
        if pose is not None:
            raise ValueError("pose is applied on the device: pass device tensors")
        x, y = _host_f32(x), _host_f32(y)
        L.check(self._lib.ndt2d_add_target_points(self._h, x.ctypes.data, y.ctypes.data, x.size, C.byref(out)),
                "ndt2d_add_target_points")
        return int(out.value)

    def grid_info(self) -> L.GridInfo2D:
        info = L.GridInfo2D()
        L.check(self._lib.ndt2d_get_grid_info(self._h, C.byref(info)), "ndt2d_get_grid_info")
        return info

    def grid(self):
        """(count int32 [H*W], mean float32 [H*W,2], icov float32 [H*W,3]) of the cached grid."""
        info = self.grid_info()
        nc = info.width * info.height
        count = np.zeros(nc, dtype=np.int32)
        mean = np.zeros((nc, 2), dtype=np.float32)
        icov = np.zeros((nc, 3), dtype=np.float32)
        L.check(self._lib.ndt2d_get_grid(self._h, count.ctypes.data, mean.ctypes.data, icov.ctypes.data),
                "ndt2d_get_grid")
        return count, mean, icov

    # ---- submap persistence
    def save_map(self) -> np.ndarray:
        """The cached grid as a flat uint8 buffer (ndt_map_header + the exact per-cell sums; ndt2d_save_map)."""
        buf = np.empty(int(self._lib.ndt2d_map_size(self._h)) or 1, dtype=np.uint8)
        L.check(self._lib.ndt2d_save_map(self._h, buf.ctypes.data, buf.size, None), "ndt2d_save_map")
        return buf

    def load_map(self, buf):
        """Make a saved map this handle's target: the sums are re-finalised with THIS handle's min_points /
        eig_ratio (ndt2d_load_map); cell_size and overlap_grids must match."""
        buf = np.ascontiguousarray(np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf.view(np.uint8))
        L.check(self._lib.ndt2d_load_map(self._h, buf.ctypes.data, buf.size), "ndt2d_load_map")

    # ---- (ii)+(iii) one evaluation
    def evaluate(self, sx, sy, pose):
        p = (C.c_double * 3)(*[float(v) for v in pose])
        out = L.Eval2D()
        if _is_dev(sx):
            n = sx.numel()
            self.wait_stream()
            L.check(self._lib.ndt2d_evaluate_dev(self._h, _dev_ptr(sx, n), _dev_ptr(sy, n), n, p, C.byref(out)), "ndt2d_evaluate_dev")
        else:
            sx, sy = _host_f32(sx), _host_f32(sy)
            L.check(self._lib.ndt2d_evaluate(self._h, sx.ctypes.data, sy.ctypes.data, sx.size, p, C.byref(out)),
                    "ndt2d_evaluate")
        return (np.array(out.H, dtype=np.float64).reshape(3, 3), np.array(out.g, dtype=np.float64),
                float(out.score), int(out.n_hit))

    def wait_stream(self, stream=None):
        """Order the handle's stream behind `stream` (default: torch's current stream): device arrays
        written there are complete before the handle's next kernels read them (ndt2d_wait_stream)."""
        if stream is None:
            import torch
            stream = torch.cuda.current_stream().cuda_stream
        L.check(self._lib.ndt2d_wait_stream(self._h, C.c_void_p(int(stream))), "ndt2d_wait_stream")

    # ---- full alignment
    def align(self, sx, sy, init_pose=(0.0, 0.0, 0.0)) -> AlignResult:
        p = (C.c_double * 3)(*[float(v) for v in init_pose])
        r = L.Result2D()
        if _is_dev(sx):
            n = sx.numel()
            self.wait_stream()           # the tensors may still be in flight on torch's current stream
            st = self._lib.ndt2d_align_dev(self._h, _dev_ptr(sx, n), _dev_ptr(sy, n), n, p, C.byref(r))
        else:
            sx, sy = _host_f32(sx), _host_f32(sy)
            st = self._lib.ndt2d_align(self._h, sx.ctypes.data, sy.ctypes.data, sx.size, p, C.byref(r))
        L.check(st, "ndt2d_align")
        return _to_result(r)

    def align_trace(self, sx, sy, init_pose=(0.0, 0.0, 0.0), capacity: int = 256):
        """Per-iteration trace (ndt2d_align_trace; host arrays): a list of AlignResult, entry j = the state
        after j + 1 updates (H, g, score, n_hit of the evaluation behind that update)."""
        sx, sy = _host_f32(sx), _host_f32(sy)
        p = (C.c_double * 3)(*[float(v) for v in init_pose])
        rows = (L.Result2D * capacity)()
        n_rows = C.c_int32(0)
        L.check(self._lib.ndt2d_align_trace(self._h, sx.ctypes.data, sy.ctypes.data, sx.size, p, C.cast(rows, C.c_void_p),
                                            capacity, C.byref(n_rows), None), "ndt2d_align_trace")
        return [_to_result(rows[j]) for j in range(n_rows.value)]

    def align_multi_start(self, sx, sy, init_poses):
        """Up to 64 alignments of the same device scan from different initial poses in one launch chain
        (ndt2d_align_multi_start_dev).  Returns a list of AlignResult, one per start."""
        poses = np.ascontiguousarray(init_poses, dtype=np.float64).reshape(-1, 3)
        m = poses.shape[0]
        n = sx.numel()
        out = (L.Result2D * m)()
        self.wait_stream()
        L.check(self._lib.ndt2d_align_multi_start_dev(self._h, _dev_ptr(sx, n), _dev_ptr(sy, n), n, poses.ctypes.data, m,
                                                      C.cast(out, C.c_void_p)), "ndt2d_align_multi_start_dev")
        return [_to_result(r) for r in out]

    def align_multi_scan(self, scans, init_poses):
        """Up to 64 different device scans, each from its own initial pose, in one launch chain
        (ndt2d_align_multi_scan_dev).  scans: list of (sx, sy) CUDA tensors.  Returns a list of AlignResult."""
        m = len(scans)
        poses = np.ascontiguousarray(init_poses, dtype=np.float64).reshape(m, 3)
        px, py, nn = (C.c_void_p * m)(), (C.c_void_p * m)(), (C.c_size_t * m)()
        for k, (sx, sy) in enumerate(scans):
            nn[k] = sx.numel()
            px[k] = _dev_ptr(sx, nn[k]).value
            py[k] = _dev_ptr(sy, nn[k]).value
        out = (L.Result2D * m)()
        self.wait_stream()
        L.check(self._lib.ndt2d_align_multi_scan_dev(self._h, px, py, nn, poses.ctypes.data, m, C.cast(out, C.c_void_p)),
                "ndt2d_align_multi_scan_dev")
        return [_to_result(r) for r in out]

    def align_async(self, sx, sy, init_pose=(0.0, 0.0, 0.0), producer_complete: bool = False):
        """Enqueue the whole loop on the handle's stream (device tensors only).  producer_complete=True:
        the caller knows the tensors are complete (their stream was synchronised), no ordering needed."""
        p = (C.c_double * 3)(*[float(v) for v in init_pose])
        n = sx.numel()
        self._keep = (sx, sy)
        if not producer_complete:
            self.wait_stream()
        L.check(self._lib.ndt2d_align_dev_async(self._h, _dev_ptr(sx, n), _dev_ptr(sy, n), n, p),
                "ndt2d_align_dev_async")

    def finish(self) -> AlignResult:
        r = L.Result2D()
        L.check(self._lib.ndt2d_align_finish(self._h, C.byref(r)), "ndt2d_align_finish")
        self._keep = None
        return _to_result(r)

    @property
    def stream(self) -> int:
        return int(self._lib.ndt2d_stream(self._h) or 0)


RESULT_DOUBLES = C.sizeof(L.Result2D) // 8     # 16 doubles + 4 int32


def _results_from_bytes(buf: np.ndarray, n: int):
    arr = (L.Result2D * n).from_buffer_copy(buf.tobytes())
    return [_to_result(r) for r in arr]


def pyramid_params(fine: L.Params2D | None = None, **overrides):
    """The library's standard 3-level coarse-to-fine schedule (ndt2d_default_pyramid) as a
    ctypes array of ndt2d_params, for NdtBatch2D(levels=...) / NdtMulti2D(levels=...)."""
    lib = L.load()
    fine = fine if fine is not None else default_params(**overrides)
    levels = (L.Params2D * 3)()
    L.check(lib.ndt2d_default_pyramid(C.byref(fine), levels), "ndt2d_default_pyramid")
    return levels


def _as_levels(levels):
    if isinstance(levels, C.Array):
        return levels, len(levels)
    arr = (L.Params2D * len(levels))(*levels)
    return arr, len(levels)


def _concat_pairs(targets, sources, inits):
    n = len(targets)
    toff = np.zeros(n + 1, dtype=np.uint64)
    soff = np.zeros(n + 1, dtype=np.uint64)
    toff[1:] = np.cumsum([len(t[0]) for t in targets])
    soff[1:] = np.cumsum([len(s[0]) for s in sources])
    tx = np.concatenate([_host_f32(t[0]) for t in targets]); ty = np.concatenate([_host_f32(t[1]) for t in targets])
    sx = np.concatenate([_host_f32(s[0]) for s in sources]); sy = np.concatenate([_host_f32(s[1]) for s in sources])
    init = np.ascontiguousarray(inits, dtype=np.float64).reshape(n, 3)
    return n, tx, ty, toff, sx, sy, soff, init


def multi_plan(n_shards: int, toff, soff, iterations_hint: int = 0, pair_iterations=None) -> np.ndarray:
    """The pair split ndt2d_multi_align uses (ndt2d_multi_plan; needs no device).  pair_iterations: optional per-pair
    iteration hints (ndt2d_multi_plan_hinted) for converged-mode batches."""
    lib = L.load()
    toff = np.ascontiguousarray(toff, dtype=np.uint64)
    soff = np.ascontiguousarray(soff, dtype=np.uint64)
    begin = np.zeros(int(n_shards) + 1, dtype=np.uint64)
    if pair_iterations is None:
        L.check(lib.ndt2d_multi_plan(int(n_shards), toff.ctypes.data, soff.ctypes.data, len(toff) - 1,
                                     int(iterations_hint), begin.ctypes.data), "ndt2d_multi_plan")
    else:
        hints = np.ascontiguousarray(pair_iterations, dtype=np.int32)
        if hints.shape != (len(toff) - 1,):
            raise ValueError("pair_iterations needs one entry per pair")
        L.check(lib.ndt2d_multi_plan_hinted(int(n_shards), toff.ctypes.data, soff.ctypes.data, len(toff) - 1,
                                            int(iterations_hint), hints.ctypes.data, begin.ctypes.data), "ndt2d_multi_plan_hinted")
    return begin


class NdtMulti2D:
    """The loop-closure batch over several devices from one host process (one context and one
    host thread per device).  Mirrors ndt2d_multi_* of include/ndt_hip.h."""

    def __init__(self, devices=None, params: L.Params2D | None = None, levels=None, **overrides):
        """levels: coarse-to-fine list of Params2D (e.g. pyramid_params()); otherwise one level."""
        self._lib = L.load()
        self.params = params if params is not None else default_params(**overrides)
        if params is not None:
            for k, v in overrides.items():
                setattr(self.params, k, v)
        h = C.c_void_p()
        if devices is None:
            ids, n = None, 0
        else:
            ids = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            n = len(devices)
        if levels is not None:
            self._levels, nl = _as_levels(levels)
            self.params = self._levels[nl - 1]
            L.check(self._lib.ndt2d_multi_create_pyramid(self._levels, nl, ids, n, C.byref(h)), "ndt2d_multi_create_pyramid")
        else:
            L.check(self._lib.ndt2d_multi_create(C.byref(self.params), ids, n, C.byref(h)), "ndt2d_multi_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ndt2d_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def device_count(self) -> int:
        return int(self._lib.ndt2d_multi_device_count(self._h))

    def align(self, targets, sources, inits):
        n, tx, ty, toff, sx, sy, soff, init = _concat_pairs(targets, sources, inits)
        out = np.zeros(n * RESULT_DOUBLES, dtype=np.float64)
        L.check(self._lib.ndt2d_multi_align(self._h, tx.ctypes.data, ty.ctypes.data, toff.ctypes.data,
                                            sx.ctypes.data, sy.ctypes.data, soff.ctypes.data, init.ctypes.data,
                                            n, out.ctypes.data), "ndt2d_multi_align")
        return _results_from_bytes(out, n)


def _multi_align_dev(self, shards):
    """ndt2d_multi_align_dev: shards[d] = dict of torch tensors resident on device d (tx, ty, toff, sx,
    sy, soff, init - the layout of NdtBatch2D.align_dev; an empty shard is None).  Every shard is
    aligned on its device and the rows are exchanged with one RCCL all-gather; returns the list of
    AlignResult in global pair order (shard 0's pairs, shard 1's, ...)."""
    import torch
    nd = self.device_count
    if len(shards) != nd:
        raise ValueError("one shard per device context")
    arr = lambda: (C.c_void_p * nd)()
    ptr = {k: arr() for k in ("tx", "ty", "toff", "sx", "sy", "soff", "init")}
    n_pairs = (C.c_size_t * nd)()
    for d, sh in enumerate(shards):
        n_pairs[d] = 0 if sh is None else int(sh["toff"].numel()) - 1
        for k in ptr:
            if sh is not None:
                t = sh[k]
                want = torch.float32 if k in ("tx", "ty", "sx", "sy") else (torch.int64 if k in ("toff", "soff") else torch.float64)
                if not (t.is_cuda and t.dtype == want and t.is_contiguous()):
                    raise ValueError("shard tensors must be contiguous CUDA tensors of the documented dtypes")
                ptr[k][d] = t.data_ptr()
    for sh in shards:                       # the contexts' streams are not torch's: finish the producers
        if sh is not None:
            torch.cuda.synchronize(sh["tx"].device)
    total = sum(n_pairs)
    out = np.zeros(total * RESULT_DOUBLES, dtype=np.float64)
    stride = C.c_size_t(0)
    L.check(self._lib.ndt2d_multi_align_dev(self._h, ptr["tx"], ptr["ty"], ptr["toff"], ptr["sx"], ptr["sy"], ptr["soff"],
                                            ptr["init"], n_pairs, None, C.byref(stride), out.ctypes.data),
            "ndt2d_multi_align_dev")
    self.last_shard_stride = int(stride.value)
    return _results_from_bytes(out, total)


NdtMulti2D.align_dev = _multi_align_dev


class NdtBatch2D:
    """Loop-closure candidate batch: independent scan pairs aligned concurrently on one GPU
    (one persistent workgroup per CU, target grid resident in LDS).  Mirrors
    ndt2d_batch_* of include/ndt_hip.h."""

    def __init__(self, device: int = 0, params: L.Params2D | None = None, levels=None, small_variant: bool = True,
                 global_workgroups: int | None = None, **overrides):
        """levels: coarse-to-fine list of Params2D (e.g. pyramid_params()); otherwise one level.
        small_variant=False: every pair on the 1024-thread kernel (ndt2d_batch_set_tuning).
        global_workgroups: workgroups (and 3.7 MB table slabs) of the global-table variant, 1..256 (default 256)."""
        self._lib = L.load()
        self.params = params if params is not None else default_params(**overrides)
        if params is not None:
            for k, v in overrides.items():
                setattr(self.params, k, v)
        h = C.c_void_p()
        if levels is not None:
            self._levels, nl = _as_levels(levels)
            self.params = self._levels[nl - 1]
            L.check(self._lib.ndt2d_batch_create_pyramid(self._levels, nl, int(device), C.byref(h)),
                    "ndt2d_batch_create_pyramid")
        else:
            L.check(self._lib.ndt2d_batch_create(C.byref(self.params), int(device), C.byref(h)), "ndt2d_batch_create")
        self._h = h
        self._keep = None
        if not small_variant:
            L.check(self._lib.ndt2d_batch_set_tuning(self._h, L.TUNING["batch_small_variant"], 0), "ndt2d_batch_set_tuning")
        if global_workgroups is not None:
            self.set_tuning("batch_global_workgroups", global_workgroups)

    def set_tuning(self, knob: str, value: int):
        L.check(self._lib.ndt2d_batch_set_tuning(self._h, L.TUNING[knob], int(value)), "ndt2d_batch_set_tuning")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ndt2d_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def stream(self) -> int:
        return int(self._lib.ndt2d_batch_stream(self._h) or 0)

    @property
    def last_large_count(self) -> int:
        """Pairs of the last align() call that needed the 1024-thread kernel variant (diagnostic)."""
        return int(self._lib.ndt2d_batch_last_large_count(self._h))

    def align(self, targets, sources, inits):
        """targets / sources: lists of (x, y) numpy pairs; inits: [n][3].  Returns a list of
        AlignResult.  Pairs over the on-chip capacity are re-run through the general path."""
        n, tx, ty, toff, sx, sy, soff, init = _concat_pairs(targets, sources, inits)
        out = np.zeros(n * RESULT_DOUBLES, dtype=np.float64)
        L.check(self._lib.ndt2d_batch_align(self._h, tx.ctypes.data, ty.ctypes.data, toff.ctypes.data,
                                            sx.ctypes.data, sy.ctypes.data, soff.ctypes.data, init.ctypes.data,
                                            n, out.ctypes.data), "ndt2d_batch_align")
        return _results_from_bytes(out, n)

    def align_dev(self, tx, ty, toff, sx, sy, soff, init, out=None, stream=None):
        """Everything already on the device (torch CUDA tensors: float32 clouds, int64
        offsets [n+1], float64 init [n,3]).  Asynchronous; returns the float64 [n,18] result
        tensor (decode with ``decode``) - valid once the stream is synchronised.  stream=None runs on
        the context's own stream, ordered behind torch's current stream on the way in and ahead of it
        on the way out, so the call composes with torch code like any torch op."""
        import torch
        n = int(toff.numel()) - 1
        if out is None:
            out = torch.empty((n, RESULT_DOUBLES), dtype=torch.float64, device=tx.device)
        for t, dt in ((tx, torch.float32), (ty, torch.float32), (sx, torch.float32), (sy, torch.float32),
                      (toff, torch.int64), (soff, torch.int64), (init, torch.float64), (out, torch.float64)):
            if not (t.is_cuda and t.dtype == dt and t.is_contiguous()):
                raise ValueError("batch tensors must be contiguous CUDA tensors of the documented dtypes")
        self._keep = (tx, ty, toff, sx, sy, soff, init, out)
        if not stream:
            # the context's own stream: order it behind torch's current stream, which may still be
            # writing the clouds
            L.check(self._lib.ndt2d_batch_wait_stream(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                    "ndt2d_batch_wait_stream")
        L.check(self._lib.ndt2d_batch_align_dev(
            self._h, C.c_void_p(tx.data_ptr()), C.c_void_p(ty.data_ptr()), C.c_void_p(toff.data_ptr()),
            C.c_void_p(sx.data_ptr()), C.c_void_p(sy.data_ptr()), C.c_void_p(soff.data_ptr()),
            C.c_void_p(init.data_ptr()), n, C.c_void_p(out.data_ptr()),
            C.c_void_p(stream if stream is not None else 0)), "ndt2d_batch_align_dev")
        if not stream:
            # ... and torch's current stream behind the context's: `out` can be consumed there in order
            torch.cuda.current_stream().wait_stream(torch.cuda.ExternalStream(self.stream))
        return out

    @staticmethod
    def decode(out_tensor):
        """float64 [n,18] device/host tensor -> list of AlignResult (synchronises)."""
        a = out_tensor.detach().cpu().numpy()
        return _results_from_bytes(np.ascontiguousarray(a).reshape(-1), a.shape[0])


@dataclass
class AlignResult3D:
    pose: tuple            # (tx, ty, tz, roll, pitch, yaw), R = Rz(yaw) Ry(pitch) Rx(roll)
    H: np.ndarray          # 6x6 Gauss-Newton Hessian of -score
    g: np.ndarray
    score: float
    iterations: int
    n_hit: int
    status: int


def default_params3d(**overrides) -> L.Params2D:
    p = L.Params2D()
    L.load().ndt3d_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(f"ndt3d_params has no field {k!r}")
        setattr(p, k, v)
    return p


class NdtMatcher3D:
    """3D SE(3) variant (BASELINE config 5); mirrors ndt3d_* of include/ndt_hip.h."""

    def __init__(self, device: int = 0, tuning: dict | None = None, **overrides):
        """tuning: execution-strategy knobs by name (L.TUNING; 3D handles know "single_sync_build")."""
        self._lib = L.load()
        self.params = default_params3d(**overrides)
        h = C.c_void_p()
        L.check(self._lib.ndt3d_create(C.byref(self.params), int(device), C.byref(h)), "ndt3d_create")
        self._h = h
        for k, v in (tuning or {}).items():
            L.check(self._lib.ndt3d_set_tuning(self._h, L.TUNING[k], int(v)), "ndt3d_set_tuning")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ndt3d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_target(self, x, y, z):
        if _is_dev(x):
            import torch
            n = x.numel()
            L.check(self._lib.ndt3d_set_target_dev(self._h, _dev_ptr(x, n), _dev_ptr(y, n), _dev_ptr(z, n), n,
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                    "ndt3d_set_target_dev")
            return self.grid_info()
        x, y, z = _host_f32(x), _host_f32(y), _host_f32(z)
        L.check(self._lib.ndt3d_set_target(self._h, x.ctypes.data, y.ctypes.data, z.ctypes.data, x.size),
                "ndt3d_set_target")
        return self.grid_info()

    def reserve_target(self, lo, hi):
        """Empty voxel grid over the box lo .. hi (x, y, z); fill it with add_target_points()."""
        a = (C.c_double * 3)(*[float(v) for v in lo])
        b = (C.c_double * 3)(*[float(v) for v in hi])
        L.check(self._lib.ndt3d_reserve_target(self._h, a, b), "ndt3d_reserve_target")
        return self.grid_info()

    def add_target_points(self, x, y, z, pose=None) -> int:
        """Merge more points into the cached voxel grid; returns how many fell outside its extent.
        Device tensors may carry a pose (tx, ty, tz, roll, pitch, yaw) that moves them into the map frame first."""
        out = C.c_size_t(0)
        if _is_dev(x):
            import torch
            n = x.numel()
            p = (C.c_double * 6)(*[float(v) for v in pose]) if pose is not None else None
            L.check(self._lib.ndt3d_add_target_points_dev(self._h, _dev_ptr(x, n), _dev_ptr(y, n), _dev_ptr(z, n), n, p,
                                                          C.byref(out), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                    "ndt3d_add_target_points_dev")
            return int(out.value)
        if pose is not None:
            raise ValueError("pose is applied on the device: pass device tensors")
        x, y, z = _host_f32(x), _host_f32(y), _host_f32(z)
        L.check(self._lib.ndt3d_add_target_points(self._h, x.ctypes.data, y.ctypes.data, z.ctypes.data, x.size,
                                                  C.byref(out)), "ndt3d_add_target_points")
        return int(out.value)

    def grid_info(self) -> L.GridInfo3D:
        info = L.GridInfo3D()
        L.check(self._lib.ndt3d_get_grid_info(self._h, C.byref(info)), "ndt3d_get_grid_info")
        return info

    def grid(self):
        info = self.grid_info()
        nc = info.width * info.height * info.depth
        count = np.zeros(nc, dtype=np.int32)
        mean = np.zeros((nc, 3), dtype=np.float32)
        icov = np.zeros((nc, 6), dtype=np.float32)
        L.check(self._lib.ndt3d_get_grid(self._h, count.ctypes.data, mean.ctypes.data, icov.ctypes.data),
                "ndt3d_get_grid")
        return count, mean, icov

    def save_map(self) -> np.ndarray:
        """The cached voxel grid as a flat uint8 buffer (ndt3d_save_map)."""
        buf = np.empty(int(self._lib.ndt3d_map_size(self._h)) or 1, dtype=np.uint8)
        L.check(self._lib.ndt3d_save_map(self._h, buf.ctypes.data, buf.size, None), "ndt3d_save_map")
        return buf

    def load_map(self, buf):
        buf = np.ascontiguousarray(np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf.view(np.uint8))
        L.check(self._lib.ndt3d_load_map(self._h, buf.ctypes.data, buf.size), "ndt3d_load_map")

    def evaluate(self, sx, sy, sz, pose):
        p = (C.c_double * 6)(*[float(v) for v in pose])
        out = L.Eval3D()
        if _is_dev(sx):
            import torch
            n = sx.numel()
            L.check(self._lib.ndt3d_wait_stream(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "ndt3d_wait_stream")
            L.check(self._lib.ndt3d_evaluate_dev(self._h, _dev_ptr(sx, n), _dev_ptr(sy, n), _dev_ptr(sz, n), n, p, C.byref(out)),
                    "ndt3d_evaluate_dev")
        else:
            sx, sy, sz = _host_f32(sx), _host_f32(sy), _host_f32(sz)
            L.check(self._lib.ndt3d_evaluate(self._h, sx.ctypes.data, sy.ctypes.data, sz.ctypes.data, sx.size, p,
                                             C.byref(out)), "ndt3d_evaluate")
        return (np.array(out.H, dtype=np.float64).reshape(6, 6), np.array(out.g, dtype=np.float64),
                float(out.score), int(out.n_hit))

    def align(self, sx, sy, sz, init_pose=(0.0,) * 6) -> AlignResult3D:
        p = (C.c_double * 6)(*[float(v) for v in init_pose])
        r = L.Result3D()
        if _is_dev(sx):
            import torch
            n = sx.numel()
            L.check(self._lib.ndt3d_wait_stream(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                    "ndt3d_wait_stream")
            st = self._lib.ndt3d_align_dev(self._h, _dev_ptr(sx, n), _dev_ptr(sy, n), _dev_ptr(sz, n), n, p, C.byref(r))
        else:
            sx, sy, sz = _host_f32(sx), _host_f32(sy), _host_f32(sz)
            st = self._lib.ndt3d_align(self._h, sx.ctypes.data, sy.ctypes.data, sz.ctypes.data, sx.size, p, C.byref(r))
        L.check(st, "ndt3d_align")
        return self._result(r)

    @staticmethod
    def _result(r) -> AlignResult3D:
        return AlignResult3D(tuple(r.pose), np.array(r.H, dtype=np.float64).reshape(6, 6),
                             np.array(r.g, dtype=np.float64), float(r.score), int(r.iterations), int(r.n_hit),
                             int(r.status))

    def align_multi_scan(self, scans, init_poses):
        """Up to 64 different device scans [(x, y, z), ...] against the cached voxel grid, each from its own initial
        pose, in one launch chain (ndt3d_align_multi_scan_dev); scan k's result equals align(scan k, pose k) bit for bit."""
        import torch
        m = len(scans)
        poses = np.ascontiguousarray(init_poses, dtype=np.float64).reshape(m, 6)
        ns = (C.c_size_t * m)(*[int(s[0].numel()) for s in scans])
        ptr = [(C.c_void_p * m)(*[_dev_ptr(s[a], s[0].numel()).value for s in scans]) for a in range(3)]
        out = (L.Result3D * m)()
        L.check(self._lib.ndt3d_wait_stream(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "ndt3d_wait_stream")
        L.check(self._lib.ndt3d_align_multi_scan_dev(self._h, C.cast(ptr[0], C.c_void_p), C.cast(ptr[1], C.c_void_p),
                                                     C.cast(ptr[2], C.c_void_p), C.cast(ns, C.c_void_p),
                                                     poses.ctypes.data_as(L._dp), m, C.cast(out, C.c_void_p)),
                "ndt3d_align_multi_scan_dev")
        return [self._result(out[k]) for k in range(m)]

    def align_multi_start(self, sx, sy, sz, init_poses):
        """Up to 64 alignments of one device scan from different initial poses in one launch chain."""
        import torch
        poses = np.ascontiguousarray(init_poses, dtype=np.float64).reshape(-1, 6)
        m, n = poses.shape[0], sx.numel()
        out = (L.Result3D * m)()
        L.check(self._lib.ndt3d_wait_stream(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "ndt3d_wait_stream")
        L.check(self._lib.ndt3d_align_multi_start_dev(self._h, _dev_ptr(sx, n), _dev_ptr(sy, n), _dev_ptr(sz, n), n,
                                                      poses.ctypes.data_as(L._dp), m, C.cast(out, C.c_void_p)),
                "ndt3d_align_multi_start_dev")
        return [self._result(out[k]) for k in range(m)]

    def align_trace(self, sx, sy, sz, init_pose=(0.0,) * 6, capacity: int = 256):
        """Per-iteration trace (ndt3d_align_trace; host arrays): a list of AlignResult3D, entry j = the state
        after j + 1 updates (H, g, score, n_hit of the evaluation behind that update)."""
        sx, sy, sz = _host_f32(sx), _host_f32(sy), _host_f32(sz)
        p = (C.c_double * 6)(*[float(v) for v in init_pose])
        rows = (L.Result3D * capacity)()
        n_rows = C.c_int32(0)
        L.check(self._lib.ndt3d_align_trace(self._h, sx.ctypes.data, sy.ctypes.data, sz.ctypes.data, sx.size, p,
                                            C.cast(rows, C.c_void_p), capacity, C.byref(n_rows), None), "ndt3d_align_trace")
        return [self._result(rows[j]) for j in range(n_rows.value)]

    def align_async(self, sx, sy, sz, init_pose=(0.0,) * 6, producer_complete: bool = False):
        """Enqueue the loop on the handle's stream (device tensors only); finish() waits and fetches."""
        import torch
        p = (C.c_double * 6)(*[float(v) for v in init_pose])
        n = sx.numel()
        self._keep = (sx, sy, sz)
        if not producer_complete:
            L.check(self._lib.ndt3d_wait_stream(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "ndt3d_wait_stream")
        L.check(self._lib.ndt3d_align_dev_async(self._h, _dev_ptr(sx, n), _dev_ptr(sy, n), _dev_ptr(sz, n), n, p),
                "ndt3d_align_dev_async")

    def finish(self) -> AlignResult3D:
        r = L.Result3D()
        L.check(self._lib.ndt3d_align_finish(self._h, C.byref(r)), "ndt3d_align_finish")
        self._keep = None
        return self._result(r)

    @property
    def stream(self) -> int:
        return int(self._lib.ndt3d_stream(self._h) or 0)


RESULT3_DOUBLES = C.sizeof(L.Result3D) // 8     # 51: ndt3d_result as float64 words


class NdtBatch3D:
    """3D loop-closure candidate batch (ndt3d_batch_* of include/ndt_hip.h): independent 3D scan pairs
    aligned concurrently, one persistent workgroup per CU with the pair's voxel grid in LDS."""

    def __init__(self, device: int = 0, levels=None, global_workgroups: int | None = None, **overrides):
        """global_workgroups: workgroups (and 7.9 MB table slabs) of the global-table variant, 1..256 (default 256)."""
        self._lib = L.load()
        self.params = default_params3d(**overrides)
        h = C.c_void_p()
        if levels is not None:
            self._levels, nl = _as_levels(levels)
            self.params = self._levels[nl - 1]
            L.check(self._lib.ndt3d_batch_create_pyramid(self._levels, nl, int(device), C.byref(h)),
                    "ndt3d_batch_create_pyramid")
        else:
            L.check(self._lib.ndt3d_batch_create(C.byref(self.params), int(device), C.byref(h)), "ndt3d_batch_create")
        self._h = h
        self._keep = None
        if global_workgroups is not None:
            self.set_tuning("batch_global_workgroups", global_workgroups)

    def set_tuning(self, knob: str, value: int):
        L.check(self._lib.ndt3d_batch_set_tuning(self._h, L.TUNING[knob], int(value)), "ndt3d_batch_set_tuning")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ndt3d_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def stream(self) -> int:
        return int(self._lib.ndt3d_batch_stream(self._h) or 0)

    @staticmethod
    def _results(buf: np.ndarray, n: int):
        arr = (L.Result3D * n).from_buffer_copy(np.ascontiguousarray(buf).tobytes())
        return [NdtMatcher3D._result(r) for r in arr]

    def align(self, targets, sources, inits):
        """targets / sources: lists of (x, y, z) numpy triples; inits: [n][6].  Returns a list of
        AlignResult3D.  Pairs over the on-chip capacity are re-run through the single-pair path."""
        n = len(targets)
        toff = np.zeros(n + 1, dtype=np.uint64)
        soff = np.zeros(n + 1, dtype=np.uint64)
        toff[1:] = np.cumsum([len(t[0]) for t in targets])
        soff[1:] = np.cumsum([len(s[0]) for s in sources])
        t = [np.concatenate([_host_f32(c[a]) for c in targets]) for a in range(3)]
        s = [np.concatenate([_host_f32(c[a]) for c in sources]) for a in range(3)]
        init = np.ascontiguousarray(inits, dtype=np.float64).reshape(n, 6)
        out = np.zeros(n * RESULT3_DOUBLES, dtype=np.float64)
        L.check(self._lib.ndt3d_batch_align(self._h, t[0].ctypes.data, t[1].ctypes.data, t[2].ctypes.data, toff.ctypes.data,
                                            s[0].ctypes.data, s[1].ctypes.data, s[2].ctypes.data, soff.ctypes.data,
                                            init.ctypes.data, n, out.ctypes.data), "ndt3d_batch_align")
        return self._results(out, n)

    def align_dev(self, t, toff, s, soff, init, out=None, stream=None):
        """Everything already on the device: t / s = (x, y, z) float32 CUDA tensors (concatenated clouds),
        int64 offsets [n+1], float64 init [n,6].  Asynchronous; returns the float64 [n,51] result tensor
        (decode with ``decode``).  Stream semantics as NdtBatch2D.align_dev."""
        import torch
        n = int(toff.numel()) - 1
        if out is None:
            out = torch.empty((n, RESULT3_DOUBLES), dtype=torch.float64, device=t[0].device)
        for x, dt in ((t[0], torch.float32), (t[1], torch.float32), (t[2], torch.float32), (s[0], torch.float32),
                      (s[1], torch.float32), (s[2], torch.float32), (toff, torch.int64), (soff, torch.int64),
                      (init, torch.float64), (out, torch.float64)):
            if not (x.is_cuda and x.dtype == dt and x.is_contiguous()):
                raise ValueError("batch tensors must be contiguous CUDA tensors of the documented dtypes")
        self._keep = (t, toff, s, soff, init, out)
        if not stream:
            L.check(self._lib.ndt3d_batch_wait_stream(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                    "ndt3d_batch_wait_stream")
        vp = lambda x: C.c_void_p(x.data_ptr())
        L.check(self._lib.ndt3d_batch_align_dev(self._h, vp(t[0]), vp(t[1]), vp(t[2]), vp(toff), vp(s[0]), vp(s[1]), vp(s[2]),
                                                vp(soff), vp(init), n, vp(out),
                                                C.c_void_p(stream if stream is not None else 0)), "ndt3d_batch_align_dev")
        if not stream:
            torch.cuda.current_stream().wait_stream(torch.cuda.ExternalStream(self.stream))
        return out

    @classmethod
    def decode(cls, out_tensor):
        """float64 [n,51] device/host tensor -> list of AlignResult3D (synchronises)."""
        a = out_tensor.detach().cpu().numpy()
        return cls._results(a.reshape(-1), a.shape[0])


class NdtMulti3D:
    """The 3D loop-closure batch over several devices from one process (ndt3d_multi_* of include/ndt_hip.h): one
    NdtBatch3D-like context and one host thread per device; `align` takes host pairs, `align_dev` device-resident
    shards whose result rows are exchanged with one RCCL all-gather."""

    def __init__(self, devices=None, levels=None, **overrides):
        self._lib = L.load()
        self.params = default_params3d(**overrides)
        h = C.c_void_p()
        ids, n = (None, 0) if devices is None else ((C.c_int32 * len(devices))(*[int(d) for d in devices]), len(devices))
        if levels is not None:
            self._levels, nl = _as_levels(levels)
            self.params = self._levels[nl - 1]
            L.check(self._lib.ndt3d_multi_create_pyramid(self._levels, nl, ids, n, C.byref(h)), "ndt3d_multi_create_pyramid")
        else:
            L.check(self._lib.ndt3d_multi_create(C.byref(self.params), ids, n, C.byref(h)), "ndt3d_multi_create")
        self._h = h
        self.last_shard_stride = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ndt3d_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def device_count(self) -> int:
        return int(self._lib.ndt3d_multi_device_count(self._h))

    def align(self, targets, sources, inits):
        """targets / sources: lists of (x, y, z) numpy triples; inits [n][6]; results in pair order."""
        n = len(targets)
        toff = np.zeros(n + 1, dtype=np.uint64)
        soff = np.zeros(n + 1, dtype=np.uint64)
        toff[1:] = np.cumsum([len(t[0]) for t in targets])
        soff[1:] = np.cumsum([len(s[0]) for s in sources])
        t = [np.concatenate([_host_f32(c[a]) for c in targets]) for a in range(3)]
        s = [np.concatenate([_host_f32(c[a]) for c in sources]) for a in range(3)]
        init = np.ascontiguousarray(inits, dtype=np.float64).reshape(n, 6)
        out = np.zeros(n * RESULT3_DOUBLES, dtype=np.float64)
        L.check(self._lib.ndt3d_multi_align(self._h, t[0].ctypes.data, t[1].ctypes.data, t[2].ctypes.data, toff.ctypes.data,
                                            s[0].ctypes.data, s[1].ctypes.data, s[2].ctypes.data, soff.ctypes.data,
                                            init.ctypes.data, n, out.ctypes.data), "ndt3d_multi_align")
        return NdtBatch3D._results(out, n)

    def align_dev(self, shards):
        """shards[d] = dict(t=(x, y, z), toff, s=(x, y, z), soff, init) of torch tensors resident on device d (the layout
        of NdtBatch3D.align_dev; None for an empty shard).  Returns the results in global pair order."""
        import torch
        nd = self.device_count
        if len(shards) != nd:
            raise ValueError("one shard per device context")
        keys = ("tx", "ty", "tz", "toff", "sx", "sy", "sz", "soff", "init")
        ptr = {k: (C.c_void_p * nd)() for k in keys}
        n_pairs = (C.c_size_t * nd)()
        for d, sh in enumerate(shards):
            n_pairs[d] = 0 if sh is None else int(sh["toff"].numel()) - 1
            if sh is None:
                continue
            flat = {"tx": sh["t"][0], "ty": sh["t"][1], "tz": sh["t"][2], "toff": sh["toff"], "sx": sh["s"][0], "sy": sh["s"][1],
                    "sz": sh["s"][2], "soff": sh["soff"], "init": sh["init"]}
            for k, t in flat.items():
                want = torch.int64 if k in ("toff", "soff") else (torch.float64 if k == "init" else torch.float32)
                if not (t.is_cuda and t.dtype == want and t.is_contiguous()):
                    raise ValueError("shard tensors must be contiguous CUDA tensors of the documented dtypes")
                ptr[k][d] = t.data_ptr()
            torch.cuda.synchronize(flat["tx"].device)        # the contexts' streams are not torch's: finish the producers
        total = sum(n_pairs)
        out = np.zeros(total * RESULT3_DOUBLES, dtype=np.float64)
        stride = C.c_size_t(0)
        L.check(self._lib.ndt3d_multi_align_dev(self._h, *[ptr[k] for k in keys], n_pairs, None, C.byref(stride), out.ctypes.data),
                "ndt3d_multi_align_dev")
        self.last_shard_stride = int(stride.value)
        return NdtBatch3D._results(out, total)


def magnusson_constants(outlier_ratio: float, cell_size: float, dim: int = 2):
    """(d1, d2) of Magnusson's outlier-mixture score for ndt2d_params / ndt3d_params."""
    d1, d2 = C.c_double(), C.c_double()
    L.check(L.load().ndt_magnusson_constants(float(outlier_ratio), float(cell_size), int(dim), C.byref(d1), C.byref(d2)),
            "ndt_magnusson_constants")
    return d1.value, d2.value


def polar_to_points(ranges, angle_min: float, angle_inc: float, range_min: float = 0.0, range_max: float = 1e30):
    """Range/bearing scan (torch CUDA float32 tensor) -> (x, y) CUDA tensors, on the device."""
    import torch
    n = ranges.numel()
    x = torch.empty(n, dtype=torch.float32, device=ranges.device)
    y = torch.empty(n, dtype=torch.float32, device=ranges.device)
    L.check(L.load().ndt2d_polar_to_points_dev(_dev_ptr(ranges, n), n, float(angle_min), float(angle_inc),
                                               float(range_min), float(range_max), _dev_ptr(x, n), _dev_ptr(y, n),
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)),
            "ndt2d_polar_to_points_dev")
    return x, y


# Coarse-to-fine schedule relative to the finest cell: (cell multiplier, eig_ratio).  Coarse
# levels widen the Gaussians (larger cells AND a larger eigenvalue floor) so that alignments
# started outside the fine grid's ~0.2-cell basin still converge.
PYRAMID_LEVELS = ((4.0, 0.1), (2.0, 0.03))


def range_image_to_points(ranges, elevations, azimuth0: float, azimuth_inc: float, range_min: float = 0.0,
                          range_max: float = 1e30):
    """Range image [n_elev, n_azim] (torch CUDA float32 tensor) of a multi-beam lidar -> (x, y, z) CUDA tensors of
    n_elev * n_azim points, on the device (ndt3d_range_image_to_points_dev)."""
    import torch
    n_elev, n_azim = ranges.shape
    n = n_elev * n_azim
    out = [torch.empty(n, dtype=torch.float32, device=ranges.device) for _ in range(3)]
    el = (C.c_double * n_elev)(*[float(v) for v in elevations])
    L.check(L.load().ndt3d_range_image_to_points_dev(_dev_ptr(ranges, n), n_elev, n_azim, el, float(azimuth0), float(azimuth_inc),
                                                     float(range_min), float(range_max), _dev_ptr(out[0], n), _dev_ptr(out[1], n),
                                                     _dev_ptr(out[2], n), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
            "ndt3d_range_image_to_points_dev")
    return tuple(out)


class NdtPyramid2D:
    """Multi-resolution alignment: one NdtMatcher2D per level, each level started from the
    previous level's pose; the last level runs with the caller's parameters."""

    def __init__(self, device: int = 0, levels=PYRAMID_LEVELS, **overrides):
        fine = default_params(**overrides)
        self.levels = []
        for mult, er in levels:
            kw = dict(overrides)
            kw.update(cell_size=fine.cell_size * mult, eig_ratio=er, eps_trans=1e-3, eps_rot=1e-4,
                      max_iterations=30, fixed_iterations=0, step_max_trans=fine.step_max_trans * mult)
            self.levels.append(NdtMatcher2D(device, **kw))
        self.levels.append(NdtMatcher2D(device, **overrides))

    def close(self):
        for m in self.levels:
            m.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_target(self, x, y):
        return [m.set_target(x, y) for m in self.levels][-1]

    def align(self, sx, sy, init_pose=(0.0, 0.0, 0.0)) -> AlignResult:
        pose, total = tuple(init_pose), 0
        r = None
        for m in self.levels:
            r = m.align(sx, sy, pose)
            total += r.iterations
            if r.status not in (L.NDT_OK, L.NDT_NOT_CONVERGED):
                break
            pose = r.pose
        r.iterations = total
        return r


class NdtPyramid3D:
    """Coarse-to-fine 3D alignment: one NdtMatcher3D per level (cells 4c and 2c with a stronger
    eigenvalue clamp and loose stops, then the caller's parameters), each level started from the
    previous level's pose - the 3D counterpart of NdtPyramid2D."""

    def __init__(self, device: int = 0, levels=PYRAMID_LEVELS, **overrides):
        fine = default_params3d(**overrides)
        self.levels = []
        for mult, er in levels:
            kw = dict(overrides)
            kw.update(cell_size=fine.cell_size * mult, eig_ratio=er, eps_trans=1e-3, eps_rot=1e-4,
                      max_iterations=30, fixed_iterations=0, step_max_trans=fine.step_max_trans * mult)
            self.levels.append(NdtMatcher3D(device, **kw))
        self.levels.append(NdtMatcher3D(device, **overrides))

    def close(self):
        for m in self.levels:
            m.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_target(self, x, y, z):
        return [m.set_target(x, y, z) for m in self.levels][-1]

    def align(self, sx, sy, sz, init_pose=(0.0,) * 6):
        pose, total, r = tuple(init_pose), 0, None
        for m in self.levels:
            r = m.align(sx, sy, sz, pose)
            total += r.iterations
            if r.status not in (L.NDT_OK, L.NDT_NOT_CONVERGED):
                break
            pose = r.pose
        r.iterations = total
        return r
