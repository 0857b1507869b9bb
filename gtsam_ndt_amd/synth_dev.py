"""Device-side twin of ``synth.py`` (ctypes over include/ndt_synth.h, libndt_synth.so).

Workload generator for bench.py and the tests - not part of the matcher.  Produces the same
scans as ``synth.make_pair`` bit for bit, directly in HBM: the 4096 loop-closure candidates of
BASELINE config 4 (6.55 GB) take tens of milliseconds instead of minutes of numpy.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

from . import synth

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libndt_synth.so")
_vp = C.c_void_p

SIGNATURES = {
    "ndt_synth_room_scene": (C.c_int32, [C.c_uint64, C.c_double, C.c_double, C.c_double, _vp]),
    "ndt_synth_sample_dev": (C.c_int32, [_vp, C.c_int32, C.c_size_t, C.c_uint64, C.c_double, C.c_uint64, _vp, _vp, _vp,
                                         _vp, _vp]),
    "ndt_synth_config4_dev": (C.c_int32, [C.c_uint64, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, _vp, _vp, _vp, _vp,
                                          _vp, _vp, _vp, _vp, _vp]),
    "ndt_synth_lidar3d_dev": (C.c_int32, [_vp, _vp, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_uint64, _vp, C.c_int32,
                                          C.c_int32, C.c_double, C.c_int32, _vp, _vp, _vp, _vp]),
    "ndt_synth_last_error": (C.c_char_p, []),
}

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        import importlib.util
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401  (one HIP runtime per process: let torch load its copy first)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _check(code: int, where: str):
    if code != 0:
        raise RuntimeError(f"{where} failed ({code}): {load().ndt_synth_last_error().decode()}")


def room_scene(seed: int, L: float, x0: float = 0.0, y0: float = 0.0) -> synth.Scene2D:
    """synth.room_scene through the C++ twin (host only)."""
    seg = np.zeros((synth.N_SEG + 4 * synth.N_BOX + 4, 4), dtype=np.float64)
    _check(load().ndt_synth_room_scene(int(seed), float(L), float(x0), float(y0), seg.ctypes.data), "ndt_synth_room_scene")
    return synth.Scene2D(seg[:, 0].copy(), seg[:, 1].copy(), seg[:, 2].copy(), seg[:, 3].copy())


def sample_scene(sc: synth.Scene2D, n: int, seed: int, sigma: float = 0.01, first: int = 0, pose=None, device="cuda:0",
                 out=None):
    """synth.sample_scene (+ synth.to_source_frame when `pose` is given), float32, on the device.
    out = (x, y): contiguous float32 CUDA tensors (or slices of one) of n elements to fill."""
    import torch
    seg = np.ascontiguousarray(np.stack([sc.ax, sc.ay, sc.bx, sc.by], axis=1), dtype=np.float64)
    if out is not None:
        x, y = out
        if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n for t in (x, y)):
            raise ValueError("out must be two contiguous float32 CUDA tensors of n elements")
    else:
        x = torch.empty(n, dtype=torch.float32, device=device)
        y = torch.empty(n, dtype=torch.float32, device=device)
    p = cs = None
    if pose is not None:
        p = (C.c_double * 3)(*[float(v) for v in pose])
        cs = (C.c_double * 2)(math.cos(pose[2]), math.sin(pose[2]))
    with torch.cuda.device(x.device):
        _check(load().ndt_synth_sample_dev(seg.ctypes.data, seg.shape[0], n, int(seed), float(sigma), int(first),
                                           C.cast(p, _vp) if p is not None else None,
                                           C.cast(cs, _vp) if cs is not None else None, x.data_ptr(), y.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream), "ndt_synth_sample_dev")
    return x, y


def config4_batch(first_pair: int, n_pairs: int, n_tgt: int = 100_000, n_src: int = 100_000, sigma: float = synth.SIGMA,
                  device="cuda:0", indices=None):
    """Candidates first_pair .. first_pair + n_pairs - 1 of BASELINE config 4 - or, with `indices`, the candidates
    of that list of global pair indices in that order (a strided shard, dist.shard_range(..., strided=True)) -
    generated in HBM in the layout ndt2d_batch_align_dev takes (the dict dist.pack_pairs builds from host pairs, as
    torch tensors) plus ``pose`` [n_pairs, 3], the generating poses."""
    import torch
    dev = torch.device(device)
    if indices is not None:
        indices = [int(k) for k in indices]
        n_pairs = len(indices)
        if not indices or indices == list(range(indices[0], indices[0] + n_pairs)):      # contiguous after all: one call
            first_pair, indices = (indices[0] if indices else 0), None
    f32 = lambda n: torch.empty(n, dtype=torch.float32, device=dev)
    t = {"tx": f32(n_pairs * n_tgt), "ty": f32(n_pairs * n_tgt), "sx": f32(n_pairs * n_src), "sy": f32(n_pairs * n_src),
         "toff": torch.empty(n_pairs + 1, dtype=torch.int64, device=dev),
         "soff": torch.empty(n_pairs + 1, dtype=torch.int64, device=dev),
         "init": torch.empty((n_pairs, 3), dtype=torch.float64, device=dev),
         "pose": torch.empty((n_pairs, 3), dtype=torch.float64, device=dev)}
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        if indices is None:
            _check(load().ndt_synth_config4_dev(int(first_pair), n_pairs, n_tgt, n_src, float(sigma), t["tx"].data_ptr(),
                                                t["ty"].data_ptr(), t["sx"].data_ptr(), t["sy"].data_ptr(), t["toff"].data_ptr(),
                                                t["soff"].data_ptr(), t["init"].data_ptr(), t["pose"].data_ptr(),
                                                stream), "ndt_synth_config4_dev")
        else:
            # one generator call per candidate, each into its slot of the batch arrays; the per-call offset pairs
            # (0, n) land in scratch words and the batch's offset tables are written once at the end
            scratch = torch.empty(4, dtype=torch.int64, device=dev)
            for j, k in enumerate(indices):
                _check(load().ndt_synth_config4_dev(k, 1, n_tgt, n_src, float(sigma), t["tx"].data_ptr() + 4 * j * n_tgt,
                                                    t["ty"].data_ptr() + 4 * j * n_tgt, t["sx"].data_ptr() + 4 * j * n_src,
                                                    t["sy"].data_ptr() + 4 * j * n_src, scratch.data_ptr(), scratch.data_ptr() + 16,
                                                    t["init"].data_ptr() + 24 * j, t["pose"].data_ptr() + 24 * j,
                                                    stream), "ndt_synth_config4_dev")
            t["toff"].copy_(torch.arange(n_pairs + 1, dtype=torch.int64, device=dev) * n_tgt)
            t["soff"].copy_(torch.arange(n_pairs + 1, dtype=torch.int64, device=dev) * n_src)
    return t


def lidar_scan3d(seed: int, pose, n_elev: int = 64, n_azim: int = 2048, sigma: float = 0.02, L: float = 40.0,
                 height: float = 6.0, scene_seed: int = 5, device="cuda:0", out=None, firing_order: bool = False):
    """synth3d.lidar_scan on the device (ndt_synth_lidar3d_dev): (x, y, z) float32 CUDA tensors of n_elev * n_azim
    sensor-frame points.  Same scene, beams and noise as the numpy generator; equal to it up to float32 rounding
    (not bit for bit: the beam directions' cos / sin come from the device's libm).  firing_order=True: the same points
    in the order a spinning lidar's driver delivers them (all beams of one bearing, then the next bearing) instead of
    ring by ring."""
    import torch
    from . import synth3d
    lo, hi = synth3d.scene_boxes(scene_seed, L, height)
    lo = np.ascontiguousarray(lo, dtype=np.float64)
    hi = np.ascontiguousarray(hi, dtype=np.float64)
    n = n_elev * n_azim
    if out is None:
        out = tuple(torch.empty(n, dtype=torch.float32, device=device) for _ in range(3))
    if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n for t in out):
        raise ValueError("out must be three contiguous float32 CUDA tensors of n_elev * n_azim elements")
    p = (C.c_double * 6)(*[float(v) for v in pose])
    with torch.cuda.device(out[0].device):
        _check(load().ndt_synth_lidar3d_dev(lo.ctypes.data, hi.ctypes.data, lo.shape[0], float(L), float(height),
                                            float(synth3d.SENSOR_Z), int(seed), C.cast(p, _vp), int(n_elev), int(n_azim),
                                            float(sigma), int(bool(firing_order)), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                            torch.cuda.current_stream().cuda_stream), "ndt_synth_lidar3d_dev")
    return out
