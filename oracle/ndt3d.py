"""CPU float64 ORACLE for the 3D NDT SE(3) variant.
TEST INFRASTRUCTURE ONLY - same rules as oracle/ndt2d.py: only tests/, smoke() and bench.py's
checker legs may import it; the product has no CPU path.

PARITY UNPINNED: the reference checkout holds no 3D (or any) NDT code
(/root/reference/README.md:1 is its only line).  This restates M. Magnusson, "The
Three-Dimensional Normal-Distributions Transform", PhD thesis, Orebro 2009 (3D voxel grid,
3x3 covariance with small-eigenvalue clamp, SE(3) Euler-angle derivatives) with this repo's
choices: R = Rz(yaw) Ry(pitch) Rx(roll), pose = (tx,ty,tz,roll,pitch,
yaw) updated additively, Gauss-Newton Hessian sum w J' S^-1 J, single-cell lookup, cyclic
Jacobi eigen-decomposition with a fixed number of sweeps so the device can mirror it.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from .ndt2d import (NDT_DEGENERATE_HESSIAN, NDT_NOT_CONVERGED, NDT_OK, NDT_TOO_FEW_CELLS,
                    NDT_TOO_FEW_HITS, wrap_angle)

JACOBI_SWEEPS = 6


@dataclass
class Ndt3Params:
    """Mirror of ndt3d_params (include/ndt_hip.h)."""
    cell_size: float = 1.0
    min_points: int = 5
    eig_ratio: float = 1e-3
    d1: float = 1.0
    d2: float = 1.0
    max_iterations: int = 100
    fixed_iterations: int = 0
    eps_trans: float = 1e-5
    eps_rot: float = 1e-5
    step_max_trans: float = 1.0
    step_max_rot: float = 0.2
    min_hits: int = 6
    line_search: int = 0               # as NdtParams.line_search (oracle/ndt2d.py gn_update)
    step_scale: float = 1.0            # as NdtParams.step_scale
    hessian_mode: int = 0              # 0: Gauss-Newton; 1: full Newton Hessian (Magnusson 2009, eq. 6.13)


@dataclass
class Grid3D:
    o: np.ndarray                # float32 [3] origin
    inv_c: np.float32
    dims: tuple                  # (W, H, D)
    count: np.ndarray            # int64 [ncell]
    mean: np.ndarray             # float64 [ncell,3]
    icov: np.ndarray             # float64 [ncell,6]  xx xy xz yy yz zz of Sigma^-1
    valid: np.ndarray
    n_valid: int = 0
    _rec32: tuple | None = field(default=None, repr=False)

    def records32(self):
        if self._rec32 is None:
            self._rec32 = (self.mean.astype(np.float32), self.icov.astype(np.float32))
        return self._rec32


def grid_geometry3(p: np.ndarray, cell: float):
    """Same rule as 2D per axis: one guard cell below the float32 minimum, extent from the
    float32 key of the maximum, plus one."""
    c = float(cell)
    inv_c = np.float32(1.0 / c)
    o = np.zeros(3, np.float32)
    dims = []
    for a in range(3):
        mn, mx = np.float32(p[:, a].min()), np.float32(p[:, a].max())
        o[a] = np.float32((math.floor(float(mn) / c) - 1.0) * c)
        dims.append(int(np.floor((mx - o[a]) * inv_c)) + 2)
    return o, inv_c, tuple(dims)


def cell_keys3(p32: np.ndarray, o, inv_c, dims):
    idx = np.floor((p32.astype(np.float32) - o.astype(np.float32)) * np.float32(inv_c)).astype(np.int64)
    inside = np.all((idx >= 0) & (idx < np.array(dims)), axis=1)
    key = np.where(inside, (idx[:, 2] * dims[1] + idx[:, 1]) * dims[0] + idx[:, 0], 0)
    return key, inside


def jacobi_eig3(S: np.ndarray):
    """Cyclic Jacobi on symmetric 3x3 matrices, vectorised over the leading axis.
    S: [n,3,3] -> (eigenvalues [n,3], eigenvectors [n,3,3] columns).  Fixed JACOBI_SWEEPS sweeps
    over (0,1), (0,2), (1,2); the device runs the same rotations in the same order."""
    A = S.copy()
    n = A.shape[0]
    V = np.tile(np.eye(3), (n, 1, 1))
    for _ in range(JACOBI_SWEEPS):
        for p, q in ((0, 1), (0, 2), (1, 2)):
            apq = A[:, p, q]
            app, aqq = A[:, p, p], A[:, q, q]
            nz = np.abs(apq) > 1e-300
            tau = np.where(nz, (aqq - app) / (2.0 * np.where(nz, apq, 1.0)), 0.0)
            with np.errstate(over="ignore"):          # tau^2 may overflow to inf: then t -> 0, as on the device
                t = np.where(nz, np.where(tau >= 0, 1.0, -1.0) / (np.abs(tau) + np.sqrt(1.0 + tau * tau)), 0.0)
            c = 1.0 / np.sqrt(1.0 + t * t)
            s = t * c
            r = 3 - p - q
            arp, arq = A[:, r, p].copy(), A[:, r, q].copy()
            A[:, p, p] = app - t * apq
            A[:, q, q] = aqq + t * apq
            A[:, p, q] = 0.0
            A[:, q, p] = 0.0
            A[:, r, p] = A[:, p, r] = c * arp - s * arq
            A[:, r, q] = A[:, q, r] = s * arp + c * arq
            vp, vq = V[:, :, p].copy(), V[:, :, q].copy()
            V[:, :, p] = c[:, None] * vp - s[:, None] * vq
            V[:, :, q] = s[:, None] * vp + c[:, None] * vq
    return np.stack([A[:, 0, 0], A[:, 1, 1], A[:, 2, 2]], axis=1), V


def finalise_cells3(n: np.ndarray, M2: np.ndarray, prm: Ndt3Params):
    """M2 [k,6] (xx xy xz yy yz zz) -> (ok [k], icov [k,6])."""
    k = n.shape[0]
    S = np.zeros((k, 3, 3))
    den = np.maximum(n - 1, 1).astype(np.float64)
    for (i, j), col in zip(((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)), range(6)):
        S[:, i, j] = S[:, j, i] = M2[:, col] / den
    lam, V = jacobi_eig3(S)
    lmax = lam.max(axis=1)
    ok = (n >= max(prm.min_points, 2)) & (lmax > 0.0)
    lam_c = np.maximum(lam, prm.eig_ratio * lmax[:, None])
    inv = 1.0 / np.where(ok[:, None], lam_c, 1.0)
    C = np.einsum("nik,nk,njk->nij", V, inv, V)
    icov = np.stack([C[:, 0, 0], C[:, 0, 1], C[:, 0, 2], C[:, 1, 1], C[:, 1, 2], C[:, 2, 2]], axis=1)
    icov[~ok] = 0.0
    return ok, icov


def build_grid3(tx, ty, tz, prm: Ndt3Params) -> Grid3D:
    P32 = np.stack([np.asarray(tx, np.float32), np.asarray(ty, np.float32), np.asarray(tz, np.float32)], axis=1)
    o, inv_c, dims = grid_geometry3(P32, prm.cell_size)
    key, inside = cell_keys3(P32, o, inv_c, dims)
    assert inside.all()
    nc = dims[0] * dims[1] * dims[2]
    P = P32.astype(np.float64)
    count = np.bincount(key, minlength=nc).astype(np.int64)
    nz = np.maximum(count, 1)[:, None]
    mean = np.stack([np.bincount(key, weights=P[:, a], minlength=nc) for a in range(3)], axis=1) / nz
    d = P - mean[key]
    mean = mean + np.stack([np.bincount(key, weights=d[:, a], minlength=nc) for a in range(3)], axis=1) / nz
    d = P - mean[key]
    M2 = np.stack([np.bincount(key, weights=d[:, i] * d[:, j], minlength=nc)
                   for i, j in ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))], axis=1)
    cand = np.nonzero(count >= max(prm.min_points, 2))[0]
    valid = np.zeros(nc, bool)
    icov = np.zeros((nc, 6))
    if cand.size:
        ok, ic = finalise_cells3(count[cand], M2[cand], prm)
        valid[cand] = ok
        icov[cand] = ic
    mean[~valid] = 0.0
    return Grid3D(o, inv_c, dims, count, mean, icov, valid, int(valid.sum()))


def rot_and_derivs(roll, pitch, yaw):
    """R = Rz Ry Rx and dR/droll, dR/dpitch, dR/dyaw."""
    ca, sa = math.cos(roll), math.sin(roll)
    cb, sb = math.cos(pitch), math.sin(pitch)
    cg, sg = math.cos(yaw), math.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
    Ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
    Rz = np.array([[cg, -sg, 0], [sg, cg, 0], [0, 0, 1]])
    dRx = np.array([[0, 0, 0], [0, -sa, -ca], [0, ca, -sa]])
    dRy = np.array([[-sb, 0, cb], [0, 0, 0], [-cb, 0, -sb]])
    dRz = np.array([[-sg, -cg, 0], [cg, -sg, 0], [0, 0, 0]])
    return Rz @ Ry @ Rx, Rz @ Ry @ dRx, Rz @ dRy @ Rx, dRz @ Ry @ Rx


def rot_second_derivs(roll, pitch, yaw):
    """The six second derivatives of R = Rz Ry Rx: {(a, b): d2R / da db}, a <= b over (0 roll, 1 pitch, 2 yaw)."""
    ca, sa = math.cos(roll), math.sin(roll)
    cb, sb = math.cos(pitch), math.sin(pitch)
    cg, sg = math.cos(yaw), math.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
    Ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
    Rz = np.array([[cg, -sg, 0], [sg, cg, 0], [0, 0, 1]])
    dRx = np.array([[0, 0, 0], [0, -sa, -ca], [0, ca, -sa]])
    dRy = np.array([[-sb, 0, cb], [0, 0, 0], [-cb, 0, -sb]])
    dRz = np.array([[-sg, -cg, 0], [cg, -sg, 0], [0, 0, 0]])
    ddRx = np.array([[0, 0, 0], [0, -ca, sa], [0, -sa, -ca]])
    ddRy = np.array([[-cb, 0, -sb], [0, 0, 0], [sb, 0, -cb]])
    ddRz = np.array([[-cg, sg, 0], [-sg, -cg, 0], [0, 0, 0]])
    return {(0, 0): Rz @ Ry @ ddRx, (0, 1): Rz @ dRy @ dRx, (0, 2): dRz @ Ry @ dRx,
            (1, 1): Rz @ ddRy @ Rx, (1, 2): dRz @ dRy @ Rx, (2, 2): ddRz @ Ry @ Rx}


def evaluate3(grid: Grid3D, sx, sy, sz, pose, prm: Ndt3Params, mirror32: bool = False):
    """H (6x6; Gauss-Newton, or the full Newton Hessian with prm.hessian_mode = 1), g (6), score, n_hit
    of f = -sum d1 exp(-d2/2 q'S^-1 q)."""
    P = np.stack([np.asarray(sx), np.asarray(sy), np.asarray(sz)], axis=1)
    R, Ra, Rb, Rg = rot_and_derivs(*pose[3:])
    t = np.array(pose[:3], dtype=np.float64)
    if mirror32:
        R32 = R.astype(np.float32).astype(np.float64)
        P64 = P.astype(np.float32).astype(np.float64)
        Pw = (P64 @ R32.T + t.astype(np.float32).astype(np.float64)).astype(np.float32)
        key, inside = cell_keys3(Pw, grid.o, grid.inv_c, grid.dims)
        mean, icov = (r.astype(np.float64) for r in grid.records32())
        Pw = Pw.astype(np.float64)
        Ra, Rb, Rg = (M.astype(np.float32).astype(np.float64) for M in (Ra, Rb, Rg))
    else:
        P64 = P.astype(np.float64)
        Pw = P64 @ R.T + t
        idx = np.floor((Pw - grid.o.astype(np.float64)) * float(grid.inv_c)).astype(np.int64)
        inside = np.all((idx >= 0) & (idx < np.array(grid.dims)), axis=1)
        key = np.where(inside, (idx[:, 2] * grid.dims[1] + idx[:, 1]) * grid.dims[0] + idx[:, 0], 0)
        mean, icov = grid.mean, grid.icov
    hit = inside & grid.valid[key]
    k = key[hit]
    q = Pw[hit] - mean[k]
    ic = icov[k]
    C = np.empty((k.size, 3, 3))
    C[:, 0, 0], C[:, 0, 1], C[:, 0, 2] = ic[:, 0], ic[:, 1], ic[:, 2]
    C[:, 1, 0], C[:, 1, 1], C[:, 1, 2] = ic[:, 1], ic[:, 3], ic[:, 4]
    C[:, 2, 0], C[:, 2, 1], C[:, 2, 2] = ic[:, 2], ic[:, 4], ic[:, 5]
    v = np.einsum("nij,nj->ni", C, q)
    m = np.einsum("ni,ni->n", q, v)
    s = prm.d1 * np.exp(-0.5 * prm.d2 * m)
    w = s * prm.d2
    p = P64[hit]
    J = np.zeros((k.size, 3, 6))
    J[:, 0, 0] = J[:, 1, 1] = J[:, 2, 2] = 1.0
    J[:, :, 3] = p @ Ra.T
    J[:, :, 4] = p @ Rb.T
    J[:, :, 5] = p @ Rg.T
    g = np.einsum("n,nik,ni->k", w, J, v)
    H = np.einsum("n,nik,nij,njl->kl", w, J, C, J)
    if prm.hessian_mode == 1:
        # Newton: - d2 (J'v)(J'v)' per point, and v' d2p'/dp_k dp_l on the rotation block.  The latter is
        # linear in M = sum w v p' (3x3), which is what the device accumulates: sum_ab R_kl[a,b] M[a,b].
        Jv = np.einsum("nik,ni->nk", J, v)
        H = H - prm.d2 * np.einsum("n,nk,nl->kl", w, Jv, Jv)
        M = np.einsum("n,na,nb->ab", w, v, p)
        dd = rot_second_derivs(*pose[3:])
        if mirror32:
            dd = {k_: m_.astype(np.float64) for k_, m_ in dd.items()}
        for (a, b), Rab in dd.items():
            t2 = float(np.sum(Rab * M))
            H[3 + a, 3 + b] += t2
            if a != b:
                H[3 + b, 3 + a] += t2
    return H, g, float(s.sum()), int(hit.sum())


LM_ATTEMPTS = 12


def solve_ldl(H: np.ndarray, g: np.ndarray):
    """(H + lam*diag|H|) d = -g by LDL^T, lam = 0, 1e-6, 1e-5, ...; any size (6 here).
    Scalar float64 in a fixed order (mirrored on the device)."""
    n = H.shape[0]
    dg = [max(abs(float(H[i, i])), 1e-12) for i in range(n)]
    lam = 0.0
    for _ in range(LM_ATTEMPTS):
        L = [[0.0] * n for _ in range(n)]
        D = [0.0] * n
        ok = True
        for j in range(n):
            p = float(H[j, j]) + lam * dg[j]
            for k in range(j):
                p -= L[j][k] * L[j][k] * D[k]
            if not (p > 1e-12 * dg[j]):
                ok = False
                break
            D[j] = p
            for i in range(j + 1, n):
                a = float(H[i, j])
                for k in range(j):
                    a -= L[i][k] * L[j][k] * D[k]
                L[i][j] = a / p
        if ok:
            z = [0.0] * n
            for i in range(n):
                a = -float(g[i])
                for k in range(i):
                    a -= L[i][k] * z[k]
                z[i] = a
            x = [0.0] * n
            for i in reversed(range(n)):
                a = z[i] / D[i]
                for k in range(i + 1, n):
                    a -= L[k][i] * x[k]
                x[i] = a
            if all(math.isfinite(v) for v in x):
                return np.array(x), True
        lam = 1e-6 if lam == 0.0 else lam * 10.0
    return np.zeros(n), False


def gn_update3(pose, H, g, n_hit, it, prm: Ndt3Params, score: float = 0.0, ls: dict | None = None):
    """6-DoF twin of oracle/ndt2d.py gn_update(), backtracking line search included."""
    from .ndt2d import LS_TOL
    use_ls = prm.line_search > 0 and ls is not None
    if use_ls and ls.get("valid") and ls["trials"] < prm.line_search and \
            (n_hit < prm.min_hits or score < ls["score"] - LS_TOL * abs(ls["score"])):
        ls["alpha"] *= 0.5
        ls["trials"] += 1
        b, st, a = ls["base"], ls["step"], ls["alpha"]
        pose = tuple(b[i] + a * st[i] for i in range(3)) + tuple(wrap_angle(b[i] + a * st[i]) for i in range(3, 6))
        it += 1
        if prm.fixed_iterations > 0:
            return pose, it, NDT_OK, it >= prm.fixed_iterations
        if it >= prm.max_iterations:
            return pose, it, NDT_NOT_CONVERGED, True
        return pose, it, NDT_OK, False
    if n_hit < prm.min_hits:
        return pose, it, NDT_TOO_FEW_HITS, True
    d, ok = solve_ldl(H, g)
    if not ok:
        return pose, it, NDT_DEGENERATE_HESSIAN, True
    d = d * prm.step_scale
    nt = math.sqrt(d[0] ** 2 + d[1] ** 2 + d[2] ** 2)
    nr = math.sqrt(d[3] ** 2 + d[4] ** 2 + d[5] ** 2)
    alpha = 1.0
    if nt > prm.step_max_trans:
        alpha = prm.step_max_trans / nt
    if nr * alpha > prm.step_max_rot:
        alpha = prm.step_max_rot / nr
    d = d * alpha
    if use_ls:
        ls.update(valid=True, base=pose, step=tuple(float(v) for v in d), score=float(score), alpha=1.0, trials=0)
    pose = (pose[0] + d[0], pose[1] + d[1], pose[2] + d[2],
            wrap_angle(pose[3] + d[3]), wrap_angle(pose[4] + d[4]), wrap_angle(pose[5] + d[5]))
    it += 1
    if prm.fixed_iterations > 0:
        return pose, it, NDT_OK, it >= prm.fixed_iterations
    if nt * alpha < prm.eps_trans and nr * alpha < prm.eps_rot:
        return pose, it, NDT_OK, True
    if it >= prm.max_iterations:
        return pose, it, NDT_NOT_CONVERGED, True
    return pose, it, NDT_OK, False


def align3(grid: Grid3D, sx, sy, sz, init_pose, prm: Ndt3Params, mirror32: bool = False, trace=None):
    pose = tuple(float(v) for v in init_pose)
    it = 0
    if grid.n_valid < 1:
        return {"pose": pose, "H": np.zeros((6, 6)), "g": np.zeros(6), "score": 0.0, "n_hit": 0,
                "iterations": 0, "status": NDT_TOO_FEW_CELLS}
    ls = {} if prm.line_search > 0 else None
    while True:
        H, g, score, n_hit = evaluate3(grid, sx, sy, sz, pose, prm, mirror32)
        if trace is not None:
            trace.append({"pose": pose, "H": H.copy(), "g": g.copy(), "score": score, "n_hit": n_hit})
        pose, it, status, done = gn_update3(pose, H, g, n_hit, it, prm, score, ls)
        if done:
            return {"pose": pose, "H": H, "g": g, "score": score, "n_hit": n_hit, "iterations": it,
                    "status": status}
