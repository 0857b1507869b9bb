"""ctypes loader for oracle/ndt_oracle.c (the C restatement of oracle/ndt2d.py).
TEST INFRASTRUCTURE ONLY - see the header of ndt_oracle.c.  Parity unpinned."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .ndt2d import NdtParams

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libndt_oracle.so")


class CParams(C.Structure):
    _fields_ = [("cell_size", C.c_double), ("min_points", C.c_int32), ("hessian_mode", C.c_int32),
                ("eig_ratio", C.c_double), ("d1", C.c_double), ("d2", C.c_double),
                ("max_iterations", C.c_int32), ("fixed_iterations", C.c_int32),
                ("eps_trans", C.c_double), ("eps_rot", C.c_double),
                ("step_max_trans", C.c_double), ("step_max_rot", C.c_double),
                ("min_hits", C.c_int32), ("reserved", C.c_int32),
                ("line_search", C.c_int32), ("reserved2", C.c_int32), ("step_scale", C.c_double)]


class CResult(C.Structure):
    _fields_ = [("pose", C.c_double * 3), ("H", C.c_double * 9), ("g", C.c_double * 3),
                ("score", C.c_double), ("iterations", C.c_int32), ("n_hit", C.c_int32),
                ("status", C.c_int32), ("reserved", C.c_int32)]


class CResult3(C.Structure):
    _fields_ = [("pose", C.c_double * 6), ("H", C.c_double * 36), ("g", C.c_double * 6),
                ("score", C.c_double), ("iterations", C.c_int32), ("n_hit", C.c_int32),
                ("status", C.c_int32), ("reserved", C.c_int32)]


_libs = {}


def load(path: str | None = None):
    """The default build (oracle/_build/libndt_oracle.so) or another build of the same source
    (bench.py's cpu_baseline leg passes its -O3 -march=native build)."""
    path = path or LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run __graft_entry__.build()")
        lib = C.CDLL(path)
        lib.orc2d_build_grid.restype = C.c_void_p
        lib.orc2d_build_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(CParams)]
        lib.orc2d_free_grid.argtypes = [C.c_void_p]
        lib.orc2d_grid_info.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        lib.orc2d_grid_copy.argtypes = [C.c_void_p] * 5
        lib.orc2d_evaluate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                       C.POINTER(CParams), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
        lib.orc2d_align.restype = C.c_int32
        lib.orc2d_align.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.POINTER(CParams), C.c_int, C.POINTER(CResult)]
        lib.orc_max_threads.restype = C.c_int32
        lib.orc3d_build_grid.restype = C.c_void_p
        lib.orc3d_build_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(CParams)]
        lib.orc3d_free_grid.argtypes = [C.c_void_p]
        lib.orc3d_grid_info.argtypes = [C.c_void_p] * 5
        lib.orc3d_grid_copy.argtypes = [C.c_void_p] * 5
        lib.orc3d_evaluate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                       C.POINTER(CParams), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc3d_align.restype = C.c_int32
        lib.orc3d_align.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.POINTER(CParams), C.c_int, C.POINTER(CResult3)]
        _libs[path] = lib
    return _libs[path]


def cparams(p: NdtParams) -> CParams:
    return CParams(p.cell_size, p.min_points, p.hessian_mode, p.eig_ratio, p.d1, p.d2, p.max_iterations,
                   p.fixed_iterations, p.eps_trans, p.eps_rot, p.step_max_trans, p.step_max_rot,
                   p.min_hits, 0, p.line_search, 0, p.step_scale)


class CGrid:
    def __init__(self, tx, ty, prm: NdtParams, lib_path: str | None = None):
        self.lib = load(lib_path)
        self.prm = cparams(prm)
        self._tx = np.ascontiguousarray(tx, dtype=np.float32)
        self._ty = np.ascontiguousarray(ty, dtype=np.float32)
        self.ptr = self.lib.orc2d_build_grid(self._tx.ctypes.data, self._ty.ctypes.data, self._tx.size,
                                             C.byref(self.prm))
        ox, oy, ic = C.c_float(), C.c_float(), C.c_float()
        W, H, nv = C.c_int32(), C.c_int32(), C.c_int32()
        self.lib.orc2d_grid_info(self.ptr, C.addressof(ox), C.addressof(oy), C.addressof(ic),
                                 C.addressof(W), C.addressof(H), C.addressof(nv))
        self.ox, self.oy, self.inv_c = np.float32(ox.value), np.float32(oy.value), np.float32(ic.value)
        self.W, self.H, self.n_valid = W.value, H.value, nv.value

    def arrays(self):
        nc = self.W * self.H
        count = np.zeros(nc, np.int64)
        mean = np.zeros((nc, 2))
        icov = np.zeros((nc, 3))
        valid = np.zeros(nc, np.uint8)
        self.lib.orc2d_grid_copy(self.ptr, count.ctypes.data, mean.ctypes.data, icov.ctypes.data,
                                 valid.ctypes.data)
        return count, mean, icov, valid.astype(bool)

    def evaluate(self, sx, sy, pose, threads: int = 1):
        sx = np.ascontiguousarray(sx, np.float32); sy = np.ascontiguousarray(sy, np.float32)
        p = np.array(pose, dtype=np.float64)
        H = np.zeros(9); g = np.zeros(3); s = C.c_double(); nh = C.c_int32()
        self.lib.orc2d_evaluate(self.ptr, sx.ctypes.data, sy.ctypes.data, sx.size, p.ctypes.data,
                                C.byref(self.prm), threads, H.ctypes.data, g.ctypes.data,
                                C.addressof(s), C.addressof(nh))
        return H.reshape(3, 3), g, s.value, nh.value

    def align(self, sx, sy, init, threads: int = 1):
        sx = np.ascontiguousarray(sx, np.float32); sy = np.ascontiguousarray(sy, np.float32)
        p = np.array(init, dtype=np.float64)
        r = CResult()
        self.lib.orc2d_align(self.ptr, sx.ctypes.data, sy.ctypes.data, sx.size, p.ctypes.data,
                             C.byref(self.prm), threads, C.byref(r))
        return {"pose": tuple(r.pose), "H": np.array(r.H).reshape(3, 3), "g": np.array(r.g),
                "score": r.score, "n_hit": r.n_hit, "iterations": r.iterations, "status": r.status}

    def close(self):
        if self.ptr:
            self.lib.orc2d_free_grid(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cparams3(p) -> CParams:
    """oracle.ndt3d.Ndt3Params -> the shared parameter struct (ndt3d_params is ndt2d_params)."""
    return CParams(p.cell_size, p.min_points, p.hessian_mode, p.eig_ratio, p.d1, p.d2, p.max_iterations,
                   p.fixed_iterations, p.eps_trans, p.eps_rot, p.step_max_trans, p.step_max_rot,
                   p.min_hits, 0, p.line_search, 0, p.step_scale)


class CGrid3:
    """The C twin of oracle.ndt3d (orc3d_*): build_grid3 / evaluate3 / align3."""

    def __init__(self, tx, ty, tz, prm, lib_path: str | None = None):
        self.lib = load(lib_path)
        self.prm = cparams3(prm)
        self._t = [np.ascontiguousarray(a, dtype=np.float32) for a in (tx, ty, tz)]
        self.ptr = self.lib.orc3d_build_grid(*(a.ctypes.data for a in self._t), self._t[0].size, C.byref(self.prm))
        o = (C.c_float * 3)()
        dims = (C.c_int32 * 3)()
        ic, nv = C.c_float(), C.c_int32()
        self.lib.orc3d_grid_info(self.ptr, C.addressof(o), C.addressof(ic), C.addressof(dims), C.addressof(nv))
        self.o = np.array(list(o), dtype=np.float32)
        self.inv_c = np.float32(ic.value)
        self.dims = tuple(int(v) for v in dims)
        self.n_valid = nv.value

    def arrays(self):
        nc = self.dims[0] * self.dims[1] * self.dims[2]
        count = np.zeros(nc, np.int64)
        mean = np.zeros((nc, 3))
        icov = np.zeros((nc, 6))
        valid = np.zeros(nc, np.uint8)
        self.lib.orc3d_grid_copy(self.ptr, count.ctypes.data, mean.ctypes.data, icov.ctypes.data, valid.ctypes.data)
        return count, mean, icov, valid.astype(bool)

    @staticmethod
    def _f32(*arrs):
        return [np.ascontiguousarray(a, np.float32) for a in arrs]

    def evaluate(self, sx, sy, sz, pose, threads: int = 1):
        s = self._f32(sx, sy, sz)
        p = np.array(pose, dtype=np.float64)
        H = np.zeros(36); g = np.zeros(6); sc = C.c_double(); nh = C.c_int32()
        self.lib.orc3d_evaluate(self.ptr, *(a.ctypes.data for a in s), s[0].size, p.ctypes.data, C.byref(self.prm),
                                threads, H.ctypes.data, g.ctypes.data, C.addressof(sc), C.addressof(nh))
        return H.reshape(6, 6), g, sc.value, nh.value

    def align(self, sx, sy, sz, init, threads: int = 1, **overrides):
        """overrides: parameter fields changed for this call only (e.g. fixed_iterations=30)."""
        s = self._f32(sx, sy, sz)
        p = np.array(init, dtype=np.float64)
        prm = CParams.from_buffer_copy(self.prm)
        for k, v in overrides.items():
            setattr(prm, k, v)
        r = CResult3()
        self.lib.orc3d_align(self.ptr, *(a.ctypes.data for a in s), s[0].size, p.ctypes.data, C.byref(prm), threads, C.byref(r))
        return {"pose": tuple(r.pose), "H": np.array(r.H).reshape(6, 6), "g": np.array(r.g), "score": r.score,
                "n_hit": r.n_hit, "iterations": r.iterations, "status": r.status}

    def close(self):
        if self.ptr:
            self.lib.orc3d_free_grid(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
