"""Synthetic 64-beam lidar scans of a box room (BASELINE config 5).

A room (40 x 40 x 6 m, floor at z = 0) with 20 axis-aligned boxes on the floor is ray-cast
from a sensor pose: 64 elevations x 2048 azimuths = 131072 rays, nearest hit, Gaussian-like
range noise.  Target = scan from the map origin pose, source = scan from the pose T*, both in
their sensor frame, so aligning source to target recovers T*.  The reference holds no data or
generator (/root/reference/README.md:1); this is this repo's own workload.
"""
from __future__ import annotations

import numpy as np

from .synth import uniform01

T_STAR_3D = (0.30, -0.20, 0.05, 0.01, -0.01, 0.03)     # tx ty tz roll pitch yaw
SENSOR_Z = 1.5


def rotation(roll: float, pitch: float, yaw: float) -> np.ndarray:
    """R = Rz(yaw) Ry(pitch) Rx(roll)."""
    ca, sa = np.cos(roll), np.sin(roll)
    cb, sb = np.cos(pitch), np.sin(pitch)
    cg, sg = np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
    Ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
    Rz = np.array([[cg, -sg, 0], [sg, cg, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def scene_boxes(seed: int, L: float = 40.0, height: float = 6.0, n_box: int = 20):
    """(lo [n,3], hi [n,3]) of the clutter boxes; the room itself is [-L/2, L/2]^2 x [0, height]."""
    u = uniform01(seed, np.arange(6 * n_box, dtype=np.uint64)).reshape(n_box, 6)
    cx = (u[:, 0] - 0.5) * (L - 8.0)
    cy = (u[:, 1] - 0.5) * (L - 8.0)
    # keep a clearing around the sensor
    r = np.hypot(cx, cy)
    push = np.where(r < 4.0, 4.0 / np.maximum(r, 1e-6), 1.0)
    cx, cy = cx * push, cy * push
    hx = 0.5 + 2.0 * u[:, 2]
    hy = 0.5 + 2.0 * u[:, 3]
    hz = 0.5 + 3.5 * u[:, 4]
    lo = np.stack([cx - hx, cy - hy, np.zeros(n_box)], axis=1)
    hi = np.stack([cx + hx, cy + hy, hz], axis=1)
    return lo, hi


def lidar_scan(seed: int, pose, n_elev: int = 64, n_azim: int = 2048, sigma: float = 0.02,
               L: float = 40.0, height: float = 6.0, scene_seed: int = 5):
    """Ray-cast scan in the SENSOR frame from sensor pose `pose` (tx,ty,tz,roll,pitch,yaw) given
    in the map frame (the sensor sits SENSOR_Z above pose's origin).  float64 [n,3]."""
    lo, hi = scene_boxes(scene_seed, L, height)
    R = rotation(*pose[3:])
    o = np.array(pose[:3], dtype=np.float64) + np.array([0.0, 0.0, SENSOR_Z])
    el = np.deg2rad(np.linspace(-24.0, 20.0, n_elev))
    az = (np.arange(n_azim) + 0.5) * (2.0 * np.pi / n_azim)
    E, A = np.meshgrid(el, az, indexing="ij")
    d_s = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], axis=-1).reshape(-1, 3)
    d = d_s @ R.T                                       # ray directions in the map frame
    n = d.shape[0]
    inv = 1.0 / np.where(np.abs(d) < 1e-12, 1e-12, d)
    # room: we are inside, take the exit distance
    rlo = np.array([-L / 2, -L / 2, 0.0]); rhi = np.array([L / 2, L / 2, height])
    t_exit = np.minimum.reduce(np.maximum((rlo - o) * inv, (rhi - o) * inv), axis=1)
    t_hit = t_exit
    for b in range(lo.shape[0]):
        t1 = (lo[b] - o) * inv
        t2 = (hi[b] - o) * inv
        tn = np.max(np.minimum(t1, t2), axis=1)
        tf = np.min(np.maximum(t1, t2), axis=1)
        ok = (tn <= tf) & (tn > 0.0)
        t_hit = np.where(ok & (tn < t_hit), tn, t_hit)
    i = np.arange(n, dtype=np.uint64) * np.uint64(4)
    noise = (uniform01(seed, i) + uniform01(seed, i + np.uint64(1))
             + uniform01(seed, i + np.uint64(2)) + uniform01(seed, i + np.uint64(3)) - 2.0) * (1.7320508075688772 * sigma)
    return d_s * (t_hit + noise)[:, None]              # sensor-frame points


def make_pair3d(n_elev: int = 64, n_azim: int = 2048, pose=T_STAR_3D, sigma: float = 0.02):
    """Config 5.  The sensor offset SENSOR_Z is common to both scans, so the relative pose of
    the two sensor frames is exactly `pose`."""
    zero = (0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
    t = lidar_scan(101, zero, n_elev, n_azim, sigma)
    s = lidar_scan(102, pose, n_elev, n_azim, sigma)
    # express the relative pose between the two sensor frames: both are lifted by SENSOR_Z
    # along the MAP z axis, so sensor_B = T* o lift and sensor_A = lift; relative = lift^-1 T* lift
    R = rotation(*pose[3:])
    lift = np.array([0.0, 0.0, SENSOR_Z])
    rel_t = np.array(pose[:3]) + lift - lift            # R_A = I: the lifts cancel
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return {"tx": f(t[:, 0]), "ty": f(t[:, 1]), "tz": f(t[:, 2]), "sx": f(s[:, 0]), "sy": f(s[:, 1]),
            "sz": f(s[:, 2]), "pose": (float(rel_t[0]), float(rel_t[1]), float(rel_t[2]), *pose[3:]),
            "init": zero, "cell": 1.0, "config": 5}
