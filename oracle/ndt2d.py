"""CPU float64 ORACLE for the 2D NDT scan-matching hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (gtsam_ndt_amd/, csrc/) never does and has no CPU fallback.

PARITY UNPINNED.  The reference checkout is empty: /root/reference/README.md:1
("# GTSAM-NDT") is its only line; there is no NDT source, test, fixture or golden
vector to follow or to pin against.  This oracle therefore
restates the *published* algorithm, with this repo's concrete choices frozen in
DESIGN.md §2 ("Algorithm contract"):

  * P. Biber, W. Strasser, "The Normal Distributions Transform: A New Approach to
    Laser Scan Matching", IROS 2003 - 2D grid, per-cell mean/covariance, small
    eigenvalue clamp (0.001 of the large one), score sum exp(-q' S^-1 q / 2),
    SE(2) Jacobian.
  * M. Magnusson, "The Three-Dimensional Normal-Distributions Transform", PhD thesis,
    Orebro 2009 - d1/d2 score form, Newton Hessian terms.
  * Welford 1962 / Chan, Golub, LeVeque 1979 - streaming / mergeable moments.

It is validated by finite differences of its own score, known-transform recovery and
an independent scipy optimiser (tests/test_oracle2d.py), not by the reference.

Stages: a1 cell key, a2 per-cell moments, a3 cell finalise,
a4 transform+lookup, a5 score, a6 Jacobian/g/H terms, a7 reduction, a8 solve/update,
a9 result.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# status codes shared with include/ndt_hip.h
NDT_OK = 0
NDT_NOT_CONVERGED = 1
NDT_DEGENERATE_HESSIAN = 2
NDT_TOO_FEW_HITS = 3
NDT_TOO_FEW_CELLS = 4

HESSIAN_GN = 0
HESSIAN_NEWTON = 1


@dataclass
class NdtParams:
    """Mirror of ndt2d_params (include/ndt_hip.h)."""
    cell_size: float = 0.5
    min_points: int = 3
    eig_ratio: float = 1e-3
    d1: float = 1.0
    d2: float = 1.0
    hessian_mode: int = HESSIAN_GN
    max_iterations: int = 100
    fixed_iterations: int = 0          # >0: exactly that many GN updates, no convergence test
    eps_trans: float = 1e-5
    eps_rot: float = 1e-5
    step_max_trans: float = 0.5
    step_max_rot: float = 0.2
    min_hits: int = 3
    overlap: int = 1                   # 1: single grid; 4: Biber's four grids shifted by half a cell
    line_search: int = 0               # >0: backtracking, at most this many halvings per step
    step_scale: float = 1.0            # over-relaxation factor on the solved step


@dataclass
class Grid2D:
    ox: np.float32
    oy: np.float32
    inv_c: np.float32
    W: int
    H: int
    count: np.ndarray            # int64 [W*H]
    mean: np.ndarray             # float64 [W*H,2]  absolute coordinates
    icov: np.ndarray             # float64 [W*H,3]  (a,b,c) of [[a,b],[b,c]] = Sigma^-1
    valid: np.ndarray            # bool [W*H]
    n_valid: int = 0
    _rec32: tuple | None = field(default=None, repr=False)

    def records32(self):
        """(mean32 [n,2], icov32 [n,3]) as the device stores them (float32)."""
        if self._rec32 is None:
            self._rec32 = (self.mean.astype(np.float32), self.icov.astype(np.float32))
        return self._rec32


# ----------------------------------------------------------------------------- a1
OVERLAP_SHIFTS = ((0.0, 0.0), (0.5, 0.0), (0.0, 0.5), (0.5, 0.5))   # in cells (Biber & Strasser 2003)


def grid_geometry(tx: np.ndarray, ty: np.ndarray, cell: float, shift=(0.0, 0.0), extra: int = 0, bounds=None):
    """Origin/extent rule.  One guard cell below the minimum; the
    extent is whatever the float32 key formula yields for the maximum, plus one.  With
    overlapping grids the origin moves down by `shift` cells and every grid gets one `extra`
    column and row so that the shifted grids still cover the maximum with a guard cell.
    bounds = (xmin, ymin, xmax, ymax): the geometry ndt2d_reserve_target derives from a caller-
    chosen extent instead of the cloud's own bounding box."""
    c = float(cell)
    inv_c = np.float32(1.0 / c)
    if bounds is None:
        xmin, xmax = np.float32(tx.min()), np.float32(tx.max())
        ymin, ymax = np.float32(ty.min()), np.float32(ty.max())
    else:
        xmin, ymin, xmax, ymax = (np.float32(v) for v in bounds)
    ox0 = np.float32((math.floor(float(xmin) / c) - 1.0) * c)
    oy0 = np.float32((math.floor(float(ymin) / c) - 1.0) * c)
    kx = int(np.floor((xmax - ox0) * inv_c))      # float32 arithmetic, on the unshifted grid
    ky = int(np.floor((ymax - oy0) * inv_c))
    ox = np.float32((math.floor(float(xmin) / c) - 1.0 - shift[0]) * c)
    oy = np.float32((math.floor(float(ymin) / c) - 1.0 - shift[1]) * c)
    return ox, oy, inv_c, kx + 2 + extra, ky + 2 + extra


def cell_keys32(x: np.ndarray, y: np.ndarray, ox, oy, inv_c, W, H, interior: bool = False):
    """floorf((x-ox)*inv_c) in float32 arithmetic, exactly as the kernels do.
    interior=True is the target-side rule: a point whose cell lies on the grid's outermost ring
    counts as outside (the ring stays empty, so that the source side may clamp instead of test)."""
    x = x.astype(np.float32, copy=False)
    y = y.astype(np.float32, copy=False)
    with np.errstate(invalid="ignore"):
        fx = np.floor((x - np.float32(ox)) * np.float32(inv_c))
        fy = np.floor((y - np.float32(oy)) * np.float32(inv_c))
    lo = 1 if interior else 0
    inside = (fx >= lo) & (fx < W - lo) & (fy >= lo) & (fy < H - lo)      # NaN compares false
    ix = np.where(inside, fx, 0).astype(np.int64)
    iy = np.where(inside, fy, 0).astype(np.int64)
    key = np.where(inside, iy * W + ix, 0)
    return key, inside


# ----------------------------------------------------------------------------- a2, a3
def finalise_cell(n: int, mx: float, my: float, m2xx: float, m2xy: float, m2yy: float,
                  prm: NdtParams):
    """Moments -> (valid, a, b, c) with Sigma = M2/(n-1), eigenvalue clamp, inverse.
    Scalar float64 code; mirrored line by line by the C restatement and the HIP
    finalise kernel."""
    if n < prm.min_points or n < 2:
        return False, 0.0, 0.0, 0.0
    sxx = m2xx / (n - 1)
    sxy = m2xy / (n - 1)
    syy = m2yy / (n - 1)
    half_tr = 0.5 * (sxx + syy)
    half_df = 0.5 * (sxx - syy)
    disc = math.sqrt(half_df * half_df + sxy * sxy)
    l1 = half_tr + disc
    l2 = half_tr - disc
    if not (l1 > 0.0):
        return False, 0.0, 0.0, 0.0
    l2c = max(l2, prm.eig_ratio * l1)
    # unit eigenvector of l1: pick the better conditioned of the two formulas
    if half_df >= 0.0:
        ex, ey = half_df + disc, sxy          # (l1 - syy, sxy)
    else:
        ex, ey = sxy, disc - half_df          # (sxy, l1 - sxx)
    nrm = math.sqrt(ex * ex + ey * ey)
    if nrm > 0.0:
        ex /= nrm
        ey /= nrm
    else:                                      # isotropic: any direction
        ex, ey = 1.0, 0.0
    i1 = 1.0 / l1
    i2 = 1.0 / l2c
    # Sigma^-1 = i2*I + (i1-i2) e e^T
    d = i1 - i2
    a = i2 + d * ex * ex
    b = d * ex * ey
    c = i2 + d * ey * ey
    return True, a, b, c


def build_grids(tx: np.ndarray, ty: np.ndarray, prm: NdtParams, bounds=None):
    """The list of grids an alignment uses: one, or Biber's four half-cell-shifted ones."""
    if prm.overlap <= 1:
        return [build_grid(tx, ty, prm, bounds=bounds)]
    return [build_grid(tx, ty, prm, shift=sh, extra=1, bounds=bounds) for sh in OVERLAP_SHIFTS]


def build_grid(tx: np.ndarray, ty: np.ndarray, prm: NdtParams, shift=(0.0, 0.0), extra: int = 0, bounds=None) -> Grid2D:
    """Rows a1-a3.  Per-cell mean and centred second moment in float64 (two-pass form,
    equal to Welford's result up to float64 rounding), then finalise.  Points outside the grid's
    interior (non-finite, beyond `bounds`, or rounded onto the outermost ring) are ignored, as
    ndt2d_set_target / ndt2d_add_target_points ignore them."""
    tx = np.ascontiguousarray(tx, dtype=np.float32)
    ty = np.ascontiguousarray(ty, dtype=np.float32)
    fin = np.isfinite(tx) & np.isfinite(ty)
    ox, oy, inv_c, W, H = grid_geometry(tx[fin], ty[fin], prm.cell_size, shift, extra, bounds)
    key, inside = cell_keys32(tx, ty, ox, oy, inv_c, W, H, interior=True)
    inside &= fin
    tx, ty, key = tx[inside], ty[inside], key[inside]
    nc = W * H
    x = tx.astype(np.float64)
    y = ty.astype(np.float64)
    count = np.bincount(key, minlength=nc).astype(np.int64)
    nz = np.maximum(count, 1)
    mx = np.bincount(key, weights=x, minlength=nc) / nz
    my = np.bincount(key, weights=y, minlength=nc) / nz
    dx = x - mx[key]
    dy = y - my[key]
    # second pass re-centres the mean (removes the float64 summation residue)
    cx = np.bincount(key, weights=dx, minlength=nc) / nz
    cy = np.bincount(key, weights=dy, minlength=nc) / nz
    mx = mx + cx
    my = my + cy
    dx = x - mx[key]
    dy = y - my[key]
    m2xx = np.bincount(key, weights=dx * dx, minlength=nc)
    m2xy = np.bincount(key, weights=dx * dy, minlength=nc)
    m2yy = np.bincount(key, weights=dy * dy, minlength=nc)
    mean = np.zeros((nc, 2))
    icov = np.zeros((nc, 3))
    valid = np.zeros(nc, dtype=bool)
    for k in np.nonzero(count >= max(prm.min_points, 2))[0]:
        ok, a, b, c = finalise_cell(int(count[k]), mx[k], my[k], m2xx[k], m2xy[k], m2yy[k], prm)
        if ok:
            valid[k] = True
            mean[k] = (mx[k], my[k])
            icov[k] = (a, b, c)
    return Grid2D(ox, oy, inv_c, W, H, count, mean, icov, valid, int(valid.sum()))


def welford_cell(xs: np.ndarray, ys: np.ndarray):
    """Textbook streaming Welford (n, mean, M2) for one cell - used by the tests to show
    the two-pass form above and the device's fixed-point sums agree with it."""
    n = 0
    mx = my = 0.0
    m2xx = m2xy = m2yy = 0.0
    for x, y in zip(xs.astype(np.float64), ys.astype(np.float64)):
        n += 1
        dx = x - mx
        dy = y - my
        mx += dx / n
        my += dy / n
        m2xx += dx * (x - mx)
        m2xy += dx * (y - my)
        m2yy += dy * (y - my)
    return n, mx, my, m2xx, m2xy, m2yy


# ----------------------------------------------------------------------------- a4-a7
def _fma32(a, b, c):
    """float32 fma emulated through float64 (product exact; one extra rounding, rare)."""
    return (a.astype(np.float64) * np.float64(b) + np.asarray(c, dtype=np.float64)).astype(np.float32)


def transform(sx, sy, pose, mirror32: bool):
    tx, ty, th = (float(v) for v in pose)
    c, s = math.cos(th), math.sin(th)
    if mirror32:
        x = sx.astype(np.float32, copy=False)
        y = sy.astype(np.float32, copy=False)
        c32, s32 = np.float32(c), np.float32(s)
        # px = fmaf(c, x, fmaf(-s, y, tx));  py = fmaf(s, x, fmaf(c, y, ty))
        px = _fma32(x, c32, _fma32(y, -s32, np.float32(tx)))
        py = _fma32(x, s32, _fma32(y, c32, np.float32(ty)))
        jx = _fma32(x, -s32, (-c32) * y)
        jy = _fma32(x, c32, (-s32) * y)
        return px, py, jx.astype(np.float64), jy.astype(np.float64)
    x = sx.astype(np.float64)
    y = sy.astype(np.float64)
    px = c * x - s * y + tx
    py = s * x + c * y + ty
    return px, py, -s * x - c * y, c * x - s * y


def evaluate(grid, sx, sy, pose, prm: NdtParams, mirror32: bool = False):
    """One grid or a list of grids (overlapping grids: the terms of all grids are summed and
    n_hit counts point-grid hits)."""
    if isinstance(grid, (list, tuple)):
        H = np.zeros((3, 3)); g = np.zeros(3); sc = 0.0; nh = 0
        for gr in grid:
            Hi, gi, si, ni = evaluate_one(gr, sx, sy, pose, prm, mirror32)
            H += Hi; g += gi; sc += si; nh += ni
        return H, g, sc, nh
    return evaluate_one(grid, sx, sy, pose, prm, mirror32)


def evaluate_one(grid: Grid2D, sx, sy, pose, prm: NdtParams, mirror32: bool = False):
    """Score, gradient and Hessian of f(p) = -sum_i d1 exp(-d2/2 q_i' S^-1 q_i) at pose p.

    Returns H (3x3), g (3), score (= -f, to be maximised), n_hit.
    mirror32=True mirrors the device's float32 transform/key/records so that per-stage
    parity can be checked tightly; mirror32=False is the float64 truth.
    """
    px, py, jx, jy = transform(sx, sy, pose, mirror32)
    if mirror32:
        key, inside = cell_keys32(px, py, grid.ox, grid.oy, grid.inv_c, grid.W, grid.H)
        mean, icov = (r.astype(np.float64) for r in grid.records32())
        px = px.astype(np.float64)
        py = py.astype(np.float64)
    else:
        fx = (px - float(grid.ox)) * float(grid.inv_c)
        fy = (py - float(grid.oy)) * float(grid.inv_c)
        ix = np.floor(fx).astype(np.int64)
        iy = np.floor(fy).astype(np.int64)
        inside = (ix >= 0) & (ix < grid.W) & (iy >= 0) & (iy < grid.H)
        key = np.where(inside, iy * grid.W + ix, 0)
        mean, icov = grid.mean, grid.icov
    hit = inside & grid.valid[key]
    k = key[hit]
    qx = px[hit] - mean[k, 0]
    qy = py[hit] - mean[k, 1]
    a, b, c = icov[k, 0], icov[k, 1], icov[k, 2]
    jx, jy = jx[hit], jy[hit]
    vx = a * qx + b * qy
    vy = b * qx + c * qy
    m = qx * vx + qy * vy
    s = prm.d1 * np.exp(-0.5 * prm.d2 * m)
    w = s * prm.d2
    vt = vx * jx + vy * jy
    ux = a * jx + b * jy
    uy = b * jx + c * jy
    g = np.array([np.sum(w * vx), np.sum(w * vy), np.sum(w * vt)])
    hxx, hxy, hyy = np.sum(w * a), np.sum(w * b), np.sum(w * c)
    hxt, hyt = np.sum(w * ux), np.sum(w * uy)
    htt = np.sum(w * (jx * ux + jy * uy))
    if prm.hessian_mode == HESSIAN_NEWTON:
        wd = w * prm.d2
        hxx -= np.sum(wd * vx * vx)
        hxy -= np.sum(wd * vx * vy)
        hyy -= np.sum(wd * vy * vy)
        hxt -= np.sum(wd * vx * vt)
        hyt -= np.sum(wd * vy * vt)
        htt -= np.sum(wd * vt * vt)
        # d2p'/dtheta2 = -(R p) = (-jy, jx) rotated: R p = (jy, -jx)
        htt += np.sum(w * (vx * (-jy) + vy * jx))
    Hm = np.array([[hxx, hxy, hxt], [hxy, hyy, hyt], [hxt, hyt, htt]])
    return Hm, g, float(np.sum(s)), int(hit.sum())


# ----------------------------------------------------------------------------- a8
LM_ATTEMPTS = 12


def solve3(H: np.ndarray, g: np.ndarray):
    """Solve (H + lam*diag(H)) d = -g by LDL^T; lam = 0, 1e-6, 1e-5, ... 1e4 on failure.
    Scalar float64 in a fixed order (mirrored in C and HIP).  Returns (d, ok)."""
    h00, h01, h02 = float(H[0, 0]), float(H[0, 1]), float(H[0, 2])
    h11, h12, h22 = float(H[1, 1]), float(H[1, 2]), float(H[2, 2])
    d0 = max(abs(h00), 1e-12)
    d1 = max(abs(h11), 1e-12)
    d2 = max(abs(h22), 1e-12)
    lam = 0.0
    for attempt in range(LM_ATTEMPTS):
        a00 = h00 + lam * d0
        a11 = h11 + lam * d1
        a22 = h22 + lam * d2
        ok = a00 > 1e-12 * d0
        if ok:                      # LDL^T: pivots a00, p1, p2 (= squared Cholesky diagonal)
            r0 = 1.0 / a00
            l10 = h01 * r0
            l20 = h02 * r0
            p1 = a11 - l10 * h01
            ok = p1 > 1e-12 * d1
        if ok:
            r1 = 1.0 / p1
            t = h12 - l20 * h01
            l21 = t * r1
            p2 = a22 - l20 * h02 - l21 * t
            ok = p2 > 1e-12 * d2
        if ok:
            r2 = 1.0 / p2
            z0 = -float(g[0])
            z1 = -float(g[1]) - l10 * z0
            z2 = -float(g[2]) - l20 * z0 - l21 * z1
            x2 = z2 * r2
            x1 = z1 * r1 - l21 * x2
            x0 = z0 * r0 - l10 * x1 - l20 * x2
            if math.isfinite(x0) and math.isfinite(x1) and math.isfinite(x2):
                return np.array([x0, x1, x2]), True
        lam = 1e-6 if lam == 0.0 else lam * 10.0
    return np.zeros(3), False


def wrap_angle(t: float) -> float:
    pi = math.pi
    if t > pi or t <= -pi:
        t = t - 2.0 * pi * math.floor((t + pi) / (2.0 * pi))
        if t <= -pi:
            t += 2.0 * pi
    return t


LS_TOL = 1e-3   # a trial is rejected when its score is below (1 - LS_TOL) x the score it started from:
                # above the cell-flip noise of the score, so float32 and float64 evaluations agree on it


def gn_update(pose, H, g, n_hit, it, prm: NdtParams, score: float = 0.0, ls: dict | None = None):
    """One a8 step.  Returns (new_pose, iterations, status, done).

    With prm.line_search > 0 and a state dict `ls` (backtracking line search, SURVEY 8f rank 3;
    [BUILD]: step halving, Armijo constant 0): the evaluation handed in is first judged as a
    TRIAL of the previous step - if it scored worse than the pose that step started from (or
    left the map), the step is halved and re-tried from that pose, at most prm.line_search
    times; every trial costs one evaluation and counts as one iteration.  Convergence is only
    tested on accepted evaluations."""
    use_ls = prm.line_search > 0 and ls is not None
    if use_ls and ls.get("valid") and ls["trials"] < prm.line_search and \
            (n_hit < prm.min_hits or score < ls["score"] - LS_TOL * abs(ls["score"])):
        ls["alpha"] *= 0.5
        ls["trials"] += 1
        b, st, a = ls["base"], ls["step"], ls["alpha"]
        pose = (b[0] + a * st[0], b[1] + a * st[1], wrap_angle(b[2] + a * st[2]))
        it += 1
        if prm.fixed_iterations > 0:
            return pose, it, NDT_OK, it >= prm.fixed_iterations
        if it >= prm.max_iterations:
            return pose, it, NDT_NOT_CONVERGED, True
        return pose, it, NDT_OK, False
    if n_hit < prm.min_hits:
        return pose, it, NDT_TOO_FEW_HITS, True
    d, ok = solve3(H, g)
    if not ok:
        return pose, it, NDT_DEGENERATE_HESSIAN, True
    d = d * prm.step_scale
    nt = math.sqrt(d[0] * d[0] + d[1] * d[1])
    nr = abs(d[2])
    alpha = 1.0
    if nt > prm.step_max_trans:
        alpha = prm.step_max_trans / nt
    if nr * alpha > prm.step_max_rot:
        alpha = prm.step_max_rot / nr
    d = d * alpha
    if use_ls:
        ls.update(valid=True, base=pose, step=(float(d[0]), float(d[1]), float(d[2])), score=float(score),
                  alpha=1.0, trials=0)
    pose = (pose[0] + d[0], pose[1] + d[1], wrap_angle(pose[2] + d[2]))
    it += 1
    if prm.fixed_iterations > 0:
        return pose, it, NDT_OK, it >= prm.fixed_iterations
    if nt * alpha < prm.eps_trans and nr * alpha < prm.eps_rot:
        return pose, it, NDT_OK, True
    if it >= prm.max_iterations:
        return pose, it, NDT_NOT_CONVERGED, True
    return pose, it, NDT_OK, False


def align(grid: Grid2D, sx, sy, init_pose, prm: NdtParams, mirror32: bool = False,
          trace: list | None = None):
    """Full Gauss-Newton loop (rows a4-a9).  H, g, score, n_hit in the result are those
    of the LAST evaluation, i.e. at the pose before the final update."""
    pose = tuple(float(v) for v in init_pose)
    it = 0
    n_valid = sum(g.n_valid for g in grid) if isinstance(grid, (list, tuple)) else grid.n_valid
    if n_valid < 1:
        return {"pose": pose, "H": np.zeros((3, 3)), "g": np.zeros(3), "score": 0.0,
                "n_hit": 0, "iterations": 0, "status": NDT_TOO_FEW_CELLS}
    ls = {} if prm.line_search > 0 else None
    while True:
        H, g, score, n_hit = evaluate(grid, sx, sy, pose, prm, mirror32)
        if trace is not None:
            trace.append({"pose": pose, "H": H.copy(), "g": g.copy(), "score": score, "n_hit": n_hit})
        pose, it, status, done = gn_update(pose, H, g, n_hit, it, prm, score, ls)
        if done:
            return {"pose": pose, "H": H, "g": g, "score": score, "n_hit": n_hit,
                    "iterations": it, "status": status}
