"""Self-validation of the 3D oracle (parity unpinned: no reference code exists)."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth3d
from oracle import ndt3d as o


@pytest.fixture(scope="module")
def small():
    d = synth3d.make_pair3d(n_elev=32, n_azim=512)
    prm = o.Ndt3Params()
    return d, prm, o.build_grid3(d["tx"], d["ty"], d["tz"], prm)


def test_jacobi_matches_lapack():
    rng = np.random.default_rng(0)
    A = rng.normal(size=(200, 3, 3))
    S = A @ A.transpose(0, 2, 1) * rng.uniform(1e-4, 10.0, size=(200, 1, 1))
    S[:20] = np.einsum("ni,nj->nij", A[:20, 0], A[:20, 0])         # rank one
    lam, V = o.jacobi_eig3(S)
    rec = np.einsum("nik,nk,njk->nij", V, lam, V)
    assert np.abs(rec - S).max() < 1e-12 * np.abs(S).max()
    assert np.abs(np.sort(lam, axis=1) - np.linalg.eigvalsh(S)).max() < 1e-11 * np.abs(S).max()
    assert np.abs(np.einsum("nki,nkj->nij", V, V) - np.eye(3)).max() < 1e-12


def test_rotation_derivatives():
    ang = np.array([0.3, -0.2, 0.7])
    R, Ra, Rb, Rg = o.rot_and_derivs(*ang)
    assert np.allclose(R, synth3d.rotation(*ang)) and np.allclose(R @ R.T, np.eye(3))
    h = 1e-6
    for k, D in enumerate((Ra, Rb, Rg)):
        e = np.zeros(3); e[k] = h
        fd = (o.rot_and_derivs(*(ang + e))[0] - o.rot_and_derivs(*(ang - e))[0]) / (2 * h)
        assert np.allclose(D, fd, atol=1e-9)


def test_gradient_matches_finite_differences(small):
    d, prm, g = small
    pose = np.array(d["pose"]) + np.array([0.01, -0.01, 0.005, 0.001, -0.001, 0.002])
    H, gr, s, n = o.evaluate3(g, d["sx"], d["sy"], d["sz"], pose, prm)
    h = 1e-7
    fd = np.zeros(6)
    for k in range(6):
        e = np.zeros(6); e[k] = h
        sp = o.evaluate3(g, d["sx"], d["sy"], d["sz"], pose + e, prm)
        sm = o.evaluate3(g, d["sx"], d["sy"], d["sz"], pose - e, prm)
        fd[k] = -(sp[2] - sm[2]) / (2 * h)
    assert np.allclose(gr, fd, rtol=1e-4, atol=1e-5 * np.abs(gr).max())
    assert np.allclose(H, H.T) and np.all(np.linalg.eigvalsh(H) > 0)


def test_solve_ldl_matches_numpy():
    rng = np.random.default_rng(2)
    for _ in range(50):
        A = rng.normal(size=(6, 6)); H = A @ A.T + 1e-3 * np.eye(6); g = rng.normal(size=6)
        d, ok = o.solve_ldl(H, g)
        assert ok and np.allclose(d, np.linalg.solve(H, -g), rtol=1e-8, atol=1e-11)
    assert not o.solve_ldl(np.full((6, 6), np.nan), np.ones(6))[1]


def test_known_transform_recovery(small):
    d, prm, g = small
    r = o.align3(g, d["sx"], d["sy"], d["sz"], d["init"], prm)
    assert r["status"] == 0
    e = np.abs(np.array(r["pose"]) - np.array(d["pose"]))
    assert e[:3].max() < 2e-2 and e[3:].max() < 3e-3
    m = o.align3(g, d["sx"], d["sy"], d["sz"], d["init"], prm, mirror32=True)
    assert np.abs(np.array(m["pose"]) - np.array(r["pose"])).max() < 1e-4


def test_newton_hessian_matches_finite_differences_of_the_gradient(small):
    """hessian_mode = 1: the full Newton Hessian (second derivatives of R through M = sum w v p') is the
    derivative of the gradient; the Gauss-Newton form is not."""
    d, _, g = small
    prm = o.Ndt3Params(hessian_mode=1)
    pose = np.array(d["pose"]) + np.array([0.01, -0.01, 0.005, 0.001, -0.001, 0.002])
    H = o.evaluate3(g, d["sx"], d["sy"], d["sz"], pose, prm)[0]
    Hgn = o.evaluate3(g, d["sx"], d["sy"], d["sz"], pose, o.Ndt3Params())[0]
    h = 1e-6
    fd = np.zeros((6, 6))
    for k in range(6):
        e = np.zeros(6); e[k] = h
        fd[:, k] = (o.evaluate3(g, d["sx"], d["sy"], d["sz"], pose + e, prm)[1]
                    - o.evaluate3(g, d["sx"], d["sy"], d["sz"], pose - e, prm)[1]) / (2 * h)
    assert np.allclose(H, H.T)
    assert np.abs(H - fd).max() / np.abs(H).max() < 1e-5
    assert np.abs(Hgn - fd).max() / np.abs(H).max() > 1e-2
    # the structural rules the device uses for the six second derivatives of R
    ang = (0.3, -0.2, 0.7)
    R, Ra, Rb, Rg = o.rot_and_derivs(*ang)
    dd = o.rot_second_derivs(*ang)
    col = lambda M: np.stack([np.zeros(3), M[:, 2], -M[:, 1]], axis=1)
    row = lambda M: np.stack([-M[1], M[0], np.zeros(3)], axis=0)
    import math
    u, w = np.array([-math.sin(ang[2]), math.cos(ang[2]), 0.0]), np.array([0.0, math.cos(ang[0]), -math.sin(ang[0])])
    assert np.allclose(dd[(0, 0)], col(Ra)) and np.allclose(dd[(0, 1)], col(Rb)) and np.allclose(dd[(0, 2)], row(Ra))
    assert np.allclose(dd[(1, 1)], -R + np.outer(u, w)) and np.allclose(dd[(1, 2)], row(Rb)) and np.allclose(dd[(2, 2)], row(Rg))


def test_3d_oracle_reduces_to_the_2d_oracle_on_planar_data():
    """Two independently written restatements (2D: closed-form eigen-decomposition, 2x3 Jacobian; 3D: cyclic
    Jacobi, 3x6 Jacobian, Euler angles) must agree when the clouds lie in the plane z = 0: the (x, y, yaw)
    block of the 3D Hessian and gradient equals the 2D ones, the other block is zero, and the alignments
    take the same steps."""
    from gtsam_ndt_amd import synth
    from oracle import ndt2d as o2
    d = synth.make_pair(2, n_tgt=30000, n_src=20000)
    z, zs = np.zeros_like(d["tx"]), np.zeros_like(d["sx"])
    p2 = o2.NdtParams(min_points=5)
    p3 = o.Ndt3Params(cell_size=0.5, min_points=5, step_max_trans=0.5, min_hits=3)
    g2 = o2.build_grid(d["tx"], d["ty"], p2)
    g3 = o.build_grid3(d["tx"], d["ty"], z, p3)
    assert g3.n_valid == g2.n_valid and g3.dims[:2] == (g2.W, g2.H)
    idx, other = [0, 1, 5], [2, 3, 4]
    for pose in (d["init"], d["pose"]):
        H2, gr2, s2, n2 = o2.evaluate(g2, d["sx"], d["sy"], pose, p2)
        H3, gr3, s3, n3 = o.evaluate3(g3, d["sx"], d["sy"], zs, (pose[0], pose[1], 0.0, 0.0, 0.0, pose[2]), p3)
        assert n3 == n2 and abs(s3 - s2) <= 1e-12 * s2
        assert np.abs(H3[np.ix_(idx, idx)] - H2).max() <= 1e-12 * np.abs(H2).max()
        assert np.abs(gr3[idx] - gr2).max() <= 1e-11 * np.sqrt(np.abs(np.diag(H2)) * s2).max()
        assert np.abs(H3[np.ix_(idx, other)]).max() == 0.0 and np.abs(gr3[other]).max() == 0.0
    r2 = o2.align(g2, d["sx"], d["sy"], d["init"], p2)
    r3 = o.align3(g3, d["sx"], d["sy"], zs, (d["init"][0], d["init"][1], 0.0, 0.0, 0.0, d["init"][2]), p3)
    assert r2["status"] == r3["status"] == 0 and r2["iterations"] == r3["iterations"]
    assert np.abs(np.array(r3["pose"])[idx] - np.array(r2["pose"])).max() < 1e-12 and np.abs(np.array(r3["pose"])[other]).max() == 0.0


def test_independent_optimiser_agrees_3d(small):
    """scipy's BFGS with numerical gradients, on the score alone (it never sees the oracle's gradient, Hessian or update
    rule), started 2 mm / 1 mrad from the oracle's answer, returns to it: the fixed point of the Gauss-Newton iteration
    is the local maximum of the score."""
    from scipy.optimize import minimize
    d, prm, grid = small
    r = o.align3(grid, d["sx"], d["sy"], d["sz"], d["init"], prm)
    assert r["status"] == 0
    f = lambda p: -o.evaluate3(grid, d["sx"], d["sy"], d["sz"], tuple(p), prm)[2]
    x0 = np.array(r["pose"]) + np.array([2e-3, -2e-3, 1e-3, 1e-3, -1e-3, 1e-3])
    m = minimize(f, x0, method="BFGS", options={"gtol": 1e-4, "eps": 1e-6, "maxiter": 200})
    # (the score is piecewise smooth - points change voxel - so a quasi-Newton search on numerical gradients stalls a few
    # tenths of a millimetre out; it started ten times farther away and must not find a better optimum)
    assert np.abs(m.x[:3] - np.array(r["pose"])[:3]).max() < 5e-4 and np.abs(m.x[3:] - np.array(r["pose"])[3:]).max() < 1e-4, (m.x, r["pose"])
    assert f(np.array(r["pose"])) <= f(m.x) + 1e-9 * abs(f(m.x))          # and what it found is no better than the oracle's optimum
