"""Seeded synthetic 2D/3D lidar scenes for tests and benchmarks.

Not part of the matcher: this is the workload generator SURVEY.md §8(d) asks for
("room + clutter" scenes, counter-based RNG so that any size can be generated in
place from a seed on the GPU box, never shipped).  The reference checkout holds no
data or generators (/root/reference/README.md:1 is its only line), so every choice
here is this repo's own.

Design rule: only exactly-rounded IEEE operations (+ - * / sqrt, comparisons) on
float64 are used after the integer RNG, so the numpy code below and the C++ twin in
``csrc/ndt_synth.cpp`` (built with -ffp-contract=off) produce bit-identical scans.
``tests/test_synth.py`` pins that.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = 0x9E3779B97F4A7C15
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)
_SQRT3 = 1.7320508075688772  # float64(sqrt(3)), shared literal with the C++ twin


def splitmix64(seed: int, counter) -> np.ndarray:
    """splitmix64 finaliser of (seed + (counter+1)*golden). Vectorised over counter."""
    c = np.asarray(counter, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (c + np.uint64(1)) * np.uint64(_GOLD) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, counter) -> np.ndarray:
    """float64 in [0,1): top 53 bits of splitmix64."""
    return (splitmix64(seed, counter) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _u(seed: int, k: int) -> float:
    return float(uniform01(seed, np.uint64(k)))


@dataclass
class Scene2D:
    """A set of line segments (ax,ay)-(bx,by)."""
    ax: np.ndarray
    ay: np.ndarray
    bx: np.ndarray
    by: np.ndarray

    @property
    def n(self) -> int:
        return int(self.ax.shape[0])

    def concat(self, other: "Scene2D") -> "Scene2D":
        return Scene2D(np.concatenate([self.ax, other.ax]), np.concatenate([self.ay, other.ay]),
                       np.concatenate([self.bx, other.bx]), np.concatenate([self.by, other.by]))


N_SEG = 24
N_BOX = 12


def room_scene(seed: int, L: float, x0: float = 0.0, y0: float = 0.0) -> Scene2D:
    """Outer LxL walls + 24 random segments + 12 boxes/diamonds, all inside the room.

    Scalar float64 arithmetic in a fixed order; mirrored by ndt_synth_room_scene().
    """
    ax, ay, bx, by = [], [], [], []

    def seg(x1, y1, x2, y2):
        ax.append(x1); ay.append(y1); bx.append(x2); by.append(y2)

    seg(x0, y0, x0 + L, y0)
    seg(x0 + L, y0, x0 + L, y0 + L)
    seg(x0 + L, y0 + L, x0, y0 + L)
    seg(x0, y0 + L, x0, y0)
    k = 0
    margin = 0.12 * L
    span = L - 2.0 * margin
    ext = 0.2 * L
    hmin = 0.006 * L
    hspan = 0.024 * L
    for _ in range(N_SEG):
        cx = x0 + margin + _u(seed, k) * span; k += 1
        cy = y0 + margin + _u(seed, k) * span; k += 1
        dx = (_u(seed, k) - 0.5) * ext; k += 1
        dy = (_u(seed, k) - 0.5) * ext; k += 1
        seg(cx - dx, cy - dy, cx + dx, cy + dy)
    for b in range(N_BOX):
        cx = x0 + margin + _u(seed, k) * span; k += 1
        cy = y0 + margin + _u(seed, k) * span; k += 1
        hx = hmin + _u(seed, k) * hspan; k += 1
        hy = hmin + _u(seed, k) * hspan; k += 1
        if b % 2 == 0:  # axis-aligned box
            seg(cx - hx, cy - hy, cx + hx, cy - hy)
            seg(cx + hx, cy - hy, cx + hx, cy + hy)
            seg(cx + hx, cy + hy, cx - hx, cy + hy)
            seg(cx - hx, cy + hy, cx - hx, cy - hy)
        else:           # diamond
            seg(cx - hx, cy, cx, cy - hy)
            seg(cx, cy - hy, cx + hx, cy)
            seg(cx + hx, cy, cx, cy + hy)
            seg(cx, cy + hy, cx - hx, cy)
    f = lambda v: np.asarray(v, dtype=np.float64)
    return Scene2D(f(ax), f(ay), f(bx), f(by))


def _cumlen(sc: Scene2D):
    dx = sc.bx - sc.ax
    dy = sc.by - sc.ay
    ln = np.sqrt(dx * dx + dy * dy)
    cum = np.zeros(sc.n + 1, dtype=np.float64)
    acc = 0.0
    for i in range(sc.n):  # sequential, like the C++ twin
        acc = acc + float(ln[i])
        cum[i + 1] = acc
    return ln, cum


def sample_scene(sc: Scene2D, n: int, seed: int, sigma: float = 0.01, first: int = 0):
    """n points uniformly by arc length (random order) + bounded quasi-normal noise.

    Point i uses counters 16*i .. 16*i+8 of stream ``seed``: one position uniform and
    4+4 uniforms for an Irwin-Hall(4) noise sample per axis (variance sigma^2).
    Returns float64 world coordinates (x, y).
    """
    ln, cum = _cumlen(sc)
    total = cum[-1]
    i = np.arange(first, first + n, dtype=np.uint64) * np.uint64(16)
    s = uniform01(seed, i) * total
    k = np.searchsorted(cum, s, side="right") - 1
    k = np.clip(k, 0, sc.n - 1)
    t = (s - cum[k]) / ln[k]
    x = sc.ax[k] + t * (sc.bx[k] - sc.ax[k])
    y = sc.ay[k] + t * (sc.by[k] - sc.ay[k])
    nx = ((uniform01(seed, i + np.uint64(1)) + uniform01(seed, i + np.uint64(2)))
          + (uniform01(seed, i + np.uint64(3)) + uniform01(seed, i + np.uint64(4))) - 2.0)
    ny = ((uniform01(seed, i + np.uint64(5)) + uniform01(seed, i + np.uint64(6)))
          + (uniform01(seed, i + np.uint64(7)) + uniform01(seed, i + np.uint64(8))) - 2.0)
    x = x + nx * (_SQRT3 * sigma)
    y = y + ny * (_SQRT3 * sigma)
    return x, y


def to_source_frame(xw, yw, pose):
    """World points -> source frame so that aligning them recovers ``pose`` (tx,ty,theta)."""
    tx, ty, th = pose
    c, s = math.cos(th), math.sin(th)
    dx = xw - tx
    dy = yw - ty
    return c * dx + s * dy, (-s) * dx + c * dy


def lidar_scan2d(sc: Scene2D, pose, n_beams: int = 1440, fov: float = 2.0 * math.pi, seed: int = 0,
                 sigma_r: float = 0.01, max_range: float = 30.0):
    """What a planar lidar at `pose` (x, y, heading) sees of the scene: n_beams ranges over `fov`
    (first beam at heading - fov/2, increment fov/n_beams), nearest segment hit per beam
    (occlusion), Irwin-Hall(4) range noise of standard deviation sigma_r, +inf where nothing
    within max_range is hit.  Returns (ranges float32 [n_beams], angle_min, angle_inc) in the
    sensor frame - the input format of ndt2d_polar_to_points_dev.  Point density falls with
    range and surfaces hide each other, unlike sample_scene()'s uniform arc-length sampling."""
    px, py, th = pose
    inc = fov / n_beams
    a_min = -0.5 * fov
    ang = th + a_min + inc * np.arange(n_beams, dtype=np.float64)
    dx, dy = np.cos(ang)[:, None], np.sin(ang)[:, None]                    # [beams, 1]
    ex, ey = (sc.bx - sc.ax)[None, :], (sc.by - sc.ay)[None, :]            # [1, segments]
    wx, wy = (sc.ax - px)[None, :], (sc.ay - py)[None, :]
    den = dx * ey - dy * ex
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (wx * ey - wy * ex) / den                                      # distance along the beam
        u = (wx * dy - wy * dx) / den                                      # position along the segment
    hit = (np.abs(den) > 1e-12) & (t > 1e-6) & (u >= 0.0) & (u <= 1.0)
    t = np.where(hit, t, np.inf)
    r = t.min(axis=1)
    i = np.arange(n_beams, dtype=np.uint64) * np.uint64(8)
    noise = ((uniform01(seed, i) + uniform01(seed, i + np.uint64(1)))
             + (uniform01(seed, i + np.uint64(2)) + uniform01(seed, i + np.uint64(3))) - 2.0) * (_SQRT3 * sigma_r)
    r = np.where(r <= max_range, r + noise, np.inf)
    return r.astype(np.float32), a_min, inc


def scan_points(ranges, angle_min, angle_inc, range_min: float = 0.05, range_max: float = 30.0):
    """Host restatement of ndt2d_polar_to_points_dev (float64 trig, float32 result)."""
    r = np.asarray(ranges, dtype=np.float64)
    ang = angle_min + angle_inc * np.arange(r.size, dtype=np.float64)
    ok = np.isfinite(r) & (r >= range_min) & (r <= range_max)
    x = np.where(ok, r * np.cos(ang), np.nan).astype(np.float32)
    y = np.where(ok, r * np.sin(ang), np.nan).astype(np.float32)
    return x, y


T_STAR = (0.10, -0.08, 0.01)
SIGMA = 0.03


def make_pair(config: int, n_tgt: int | None = None, n_src: int | None = None,
              pair_index: int = 0, sigma: float = SIGMA):
    """Scan pairs of BASELINE.json's configs.  Returns a dict with float32
    SoA arrays tx, ty (target), sx, sy (source, in the sensor frame), the initial guess
    ``init`` and the generating pose ``pose`` (tx, ty, theta).

    1: 8 m room, 1k/1k points        2: 50 m room, 100k/100k
    3: 200 m submap (4x4 rooms) 1M points vs a 100k-point scan taken in room (2,1)
    4: loop-closure candidate ``pair_index``: 50 m room, 100k/100k, random offset
    """
    err = T_STAR
    if config == 1:
        L, S, nt, ns = 8.0, 1, 1000, 1000
    elif config == 2:
        L, S, nt, ns = 50.0, 2, 100_000, 100_000
    elif config == 3:
        L, S, nt, ns = 50.0, 3, 1_000_000, 100_000
    elif config == 4:
        L, S, nt, ns = 50.0, 9000 + pair_index, 100_000, 100_000
        r = uniform01(7000 + pair_index, np.arange(3, dtype=np.uint64))
        err = (float((r[0] - 0.5) * 0.2), float((r[1] - 0.5) * 0.2), float((r[2] - 0.5) * 0.02))
    else:
        raise ValueError("config must be 1..4 (the 3D config 5 lives in synth3d)")
    nt = n_tgt or nt
    ns = n_src or ns
    if config == 3:
        tiles = 4
        half = 0.5 * tiles * L
        tgt_scene = None
        for j in range(tiles):
            for i in range(tiles):
                r_ = room_scene(S + 1000 * (j * tiles + i), L, i * L - half, j * L - half)
                tgt_scene = r_ if tgt_scene is None else tgt_scene.concat(r_)
        i, j = 2, 1
        src_scene = room_scene(S + 1000 * (j * tiles + i), L, i * L - half, j * L - half)
        sensor = (i * L - half + 0.5 * L, j * L - half + 0.5 * L)
    else:
        tgt_scene = src_scene = room_scene(S, L, -0.5 * L, -0.5 * L)
        sensor = (0.0, 0.0)
    init = (sensor[0], sensor[1], 0.0)
    pose = (sensor[0] + err[0], sensor[1] + err[1], err[2])
    xt, yt = sample_scene(tgt_scene, nt, seed=S * 7919 + 11, sigma=sigma)
    xs, ys = sample_scene(src_scene, ns, seed=S * 7919 + 12, sigma=sigma)
    xs, ys = to_source_frame(xs, ys, pose)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return {"tx": f(xt), "ty": f(yt), "sx": f(xs), "sy": f(ys), "pose": pose, "init": init,
            "cell": 0.5, "config": config}
