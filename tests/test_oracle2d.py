"""Self-validation of the CPU oracle (it has no reference to be pinned against: the reference
checkout is empty, SURVEY.md section 8c, so these are the checks that make it trustworthy):
finite differences of its own score, known-transform recovery, an independent optimiser,
Welford equivalence, and the small linear-algebra pieces against numpy."""
import math

import numpy as np
import pytest

from gtsam_ndt_amd import synth
from oracle import ndt2d as o


@pytest.fixture(scope="module")
def cfg1():
    d = synth.make_pair(1)
    prm = o.NdtParams()
    return d, prm, o.build_grid(d["tx"], d["ty"], prm)


def _score(grid, d, pose, prm):
    return o.evaluate(grid, d["sx"], d["sy"], pose, prm)[2]


def _keys(grid, d, pose):
    px, py, _, _ = o.transform(d["sx"], d["sy"], pose, False)
    ix = np.floor((px - float(grid.ox)) * float(grid.inv_c)).astype(np.int64)
    iy = np.floor((py - float(grid.oy)) * float(grid.inv_c)).astype(np.int64)
    return iy * 100000 + ix


def _clean_stencil(grid, d, pose, h):
    """True when no source point changes cell anywhere in the +-h finite-difference stencil
    (the NDT score is only piecewise smooth)."""
    k0 = _keys(grid, d, pose)
    for k in range(3):
        for sgn in (-1.0, 1.0):
            e = np.zeros(3); e[k] = sgn * h
            if not np.array_equal(k0, _keys(grid, d, np.asarray(pose) + e)):
                return False
    return True


def test_gradient_matches_finite_differences(cfg1):
    d, prm, grid = cfg1
    rng = np.random.default_rng(1)
    h = 1e-6
    checked = 0
    while checked < 5:
        pose = np.array(d["pose"]) + rng.normal(0, [0.02, 0.02, 0.003])
        if not _clean_stencil(grid, d, pose, h):
            continue
        checked += 1
        _, g, _, n0 = o.evaluate(grid, d["sx"], d["sy"], pose, prm)
        fd = np.zeros(3)
        for k in range(3):
            e = np.zeros(3); e[k] = h
            sp = o.evaluate(grid, d["sx"], d["sy"], pose + e, prm)
            sm = o.evaluate(grid, d["sx"], d["sy"], pose - e, prm)
            assert sp[3] == n0 == sm[3]          # no point changed cell inside the stencil
            fd[k] = -(sp[2] - sm[2]) / (2 * h)   # g is the gradient of -score
        assert np.allclose(g, fd, rtol=1e-6, atol=1e-6 * np.abs(g).max())


def test_newton_hessian_matches_finite_differences_of_gradient(cfg1):
    d, _, grid = cfg1
    prm = o.NdtParams(hessian_mode=o.HESSIAN_NEWTON)
    h = 1e-6
    rng = np.random.default_rng(2)
    pose = np.array(d["pose"]) + np.array([0.01, -0.02, 0.002])
    while not _clean_stencil(grid, d, pose, h):
        pose = np.array(d["pose"]) + rng.normal(0, [0.02, 0.02, 0.003])
    H, _, _, n0 = o.evaluate(grid, d["sx"], d["sy"], pose, prm)
    fd = np.zeros((3, 3))
    for k in range(3):
        e = np.zeros(3); e[k] = h
        gp = o.evaluate(grid, d["sx"], d["sy"], pose + e, prm)
        gm = o.evaluate(grid, d["sx"], d["sy"], pose - e, prm)
        assert gp[3] == n0 == gm[3]
        fd[:, k] = (gp[1] - gm[1]) / (2 * h)
    assert np.allclose(H, fd, rtol=1e-4, atol=1e-6 * np.abs(H).max())
    assert np.allclose(H, H.T)


def test_gauss_newton_hessian_is_the_weighted_normal_matrix(cfg1):
    """GN mode: H = sum_i w_i J_i' S_i^-1 J_i, computed here point by point."""
    d, prm, grid = cfg1
    pose = d["pose"]
    H, g, score, n_hit = o.evaluate(grid, d["sx"], d["sy"], pose, prm)
    c, s = math.cos(pose[2]), math.sin(pose[2])
    Hs = np.zeros((3, 3)); gs = np.zeros(3); sc = 0.0; hits = 0
    for x, y in zip(d["sx"].astype(np.float64), d["sy"].astype(np.float64)):
        p = np.array([c * x - s * y + pose[0], s * x + c * y + pose[1]])
        ix = math.floor((p[0] - float(grid.ox)) * float(grid.inv_c))
        iy = math.floor((p[1] - float(grid.oy)) * float(grid.inv_c))
        if not (0 <= ix < grid.W and 0 <= iy < grid.H):
            continue
        k = iy * grid.W + ix
        if not grid.valid[k]:
            continue
        a, b, cc = grid.icov[k]
        S = np.array([[a, b], [b, cc]])
        q = p - grid.mean[k]
        J = np.array([[1, 0, -s * x - c * y], [0, 1, c * x - s * y]])
        w = math.exp(-0.5 * q @ S @ q)
        Hs += w * J.T @ S @ J; gs += w * J.T @ S @ q; sc += w; hits += 1
    assert hits == n_hit
    assert np.allclose(H, Hs, rtol=1e-10) and np.allclose(g, gs, rtol=1e-9, atol=1e-9) and abs(sc - score) < 1e-9
    assert np.all(np.linalg.eigvalsh(H) > 0)


def test_known_transform_recovery():
    """Aligning a scan against an independent sample of the same surfaces recovers the
    generating pose up to sampling noise (tolerance tightens with the point count)."""
    for cfg, tol in ((1, 1.5e-2), (2, 2e-3)):
        d = synth.make_pair(cfg)
        prm = o.NdtParams()
        r = o.align(o.build_grid(d["tx"], d["ty"], prm), d["sx"], d["sy"], d["init"], prm)
        assert r["status"] == o.NDT_OK
        assert np.abs(np.array(r["pose"]) - np.array(d["pose"])).max() < tol


def test_independent_optimiser_agrees(cfg1):
    """scipy's Nelder-Mead on the same score, started at the oracle's answer, does not move
    away from it (the fixed point of the GN iteration is a local maximum of the score)."""
    from scipy.optimize import minimize
    d, prm, grid = cfg1
    r = o.align(grid, d["sx"], d["sy"], d["init"], prm)
    f = lambda p: -_score(grid, d, p, prm)
    m = minimize(f, np.array(r["pose"]), method="Nelder-Mead",
                 options={"xatol": 1e-7, "fatol": 1e-10, "initial_simplex":
                          np.array(r["pose"]) + 1e-3 * np.vstack([np.zeros(3), np.eye(3)])})
    assert np.abs(m.x - np.array(r["pose"])).max() < 2e-3
    assert f(m.x) <= f(np.array(r["pose"])) + 1e-9


def test_two_pass_moments_equal_welford(cfg1):
    d, prm, grid = cfg1
    key, _ = o.cell_keys32(d["tx"], d["ty"], grid.ox, grid.oy, grid.inv_c, grid.W, grid.H)
    for k in np.nonzero(grid.valid)[0][:40]:
        sel = key == k
        n, mx, my, m2xx, m2xy, m2yy = o.welford_cell(d["tx"][sel], d["ty"][sel])
        assert n == grid.count[k]
        assert abs(mx - grid.mean[k, 0]) < 1e-12 and abs(my - grid.mean[k, 1]) < 1e-12
        ok, a, b, c = o.finalise_cell(n, mx, my, m2xx, m2xy, m2yy, prm)
        assert ok and np.allclose([a, b, c], grid.icov[k], rtol=1e-9)


def test_finalise_cell_inverts_clamped_covariance():
    prm = o.NdtParams()
    rng = np.random.default_rng(3)
    for _ in range(200):
        A = rng.normal(size=(2, 2))
        S = A @ A.T * rng.uniform(1e-4, 1.0)
        if rng.uniform() < 0.3:
            S = np.outer(A[0], A[0]) + 1e-9 * np.eye(2)      # nearly singular -> clamp acts
        n = 10
        ok, a, b, c = o.finalise_cell(n, 0.0, 0.0, S[0, 0] * (n - 1), S[0, 1] * (n - 1), S[1, 1] * (n - 1), prm)
        assert ok
        w, V = np.linalg.eigh(S)
        w[0] = max(w[0], prm.eig_ratio * w[1])
        ref = np.linalg.inv(V @ np.diag(w) @ V.T)
        assert np.allclose([[a, b], [b, c]], ref, rtol=1e-7, atol=1e-9 * np.abs(ref).max())
    assert o.finalise_cell(2, 0, 0, 1.0, 0.0, 1.0, prm)[0] is False     # n < min_points
    assert o.finalise_cell(5, 0, 0, 0.0, 0.0, 0.0, prm)[0] is False     # all points coincide


def test_solve3_and_angle_wrap():
    rng = np.random.default_rng(5)
    for _ in range(100):
        A = rng.normal(size=(3, 3)); H = A @ A.T + 1e-3 * np.eye(3); g = rng.normal(size=3)
        d, ok = o.solve3(H, g)
        assert ok and np.allclose(d, np.linalg.solve(H, -g), rtol=1e-9, atol=1e-12)
    d, ok = o.solve3(np.diag([1.0, -1.0, 1.0]), np.ones(3))            # indefinite -> damped
    assert ok and np.isfinite(d).all()
    d, ok = o.solve3(np.zeros((3, 3)), np.ones(3))                      # flat: damped gradient step
    assert ok and np.isfinite(d).all()
    d, ok = o.solve3(np.full((3, 3), np.nan), np.ones(3))               # hopeless
    assert not ok
    for t in (-7.0, -math.pi, -3.0, 0.0, 3.0, math.pi, 7.0, 100.0):
        w = o.wrap_angle(t)
        assert -math.pi < w <= math.pi and abs(math.sin(w) - math.sin(t)) < 1e-12


def test_mirror32_mode_tracks_truth(cfg1):
    d, prm, grid = cfg1
    Ht, gt, st, nt = o.evaluate(grid, d["sx"], d["sy"], d["pose"], prm)
    Hm, gm, sm, nm = o.evaluate(grid, d["sx"], d["sy"], d["pose"], prm, mirror32=True)
    assert abs(nt - nm) <= 1 and abs(st - sm) / st < 1e-3 and np.abs(Ht - Hm).max() / np.abs(Ht).max() < 1e-3


def test_edge_statuses():
    prm = o.NdtParams()
    d = synth.make_pair(1)
    g = o.build_grid(np.array([0.0, 10.0], np.float32), np.array([0.0, 10.0], np.float32), prm)
    assert o.align(g, d["sx"], d["sy"], (0, 0, 0), prm)["status"] == o.NDT_TOO_FEW_CELLS
    g = o.build_grid(d["tx"], d["ty"], prm)
    r = o.align(g, d["sx"] + 1000.0, d["sy"], (0, 0, 0), prm)
    assert r["status"] == o.NDT_TOO_FEW_HITS and r["iterations"] == 0
    r = o.align(g, d["sx"], d["sy"], d["init"], o.NdtParams(max_iterations=3))
    assert r["status"] == o.NDT_NOT_CONVERGED and r["iterations"] == 3
    r = o.align(g, d["sx"], d["sy"], d["init"], o.NdtParams(fixed_iterations=7))
    assert r["status"] == o.NDT_OK and r["iterations"] == 7
