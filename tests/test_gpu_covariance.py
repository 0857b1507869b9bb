"""Covariance calibration behind the GTSAM factor adapter.

Monte-Carlo over noise realisations of BOTH scans of a pair (same scene, same true pose, fresh
sampling and noise every time): the empirical covariance of the converged pose against the calibrated
one, S H^-1 S (ndt2d_calibrated_covariance), for the Gauss-Newton and the Newton form of H.  Also
shows what the calibration is for: H^-1 alone underestimates the scatter 4x to 18x in variance.
All alignments run on the GPU (one loop-closure batch of R pairs per case)."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth

gpu = pytest.mark.gpu

R = 240
TRUE_POSE = (0.10, -0.08, 0.01)
# (scene seed, room size, points per scan, noise sigma): the config-2 scene at its full density, and a
# smaller, sparser room
CASES = [(2, 50.0, 100_000, 0.03), (7, 30.0, 20_000, 0.03)]


def _monte_carlo(scene_seed, L, n, sigma, mode):
    import torch
    from gtsam_ndt_amd import synth_dev
    from gtsam_ndt_amd.matcher import NdtBatch2D
    scene = synth.room_scene(scene_seed, L, -0.5 * L, -0.5 * L)
    dev = torch.device("cuda:0")
    tx, ty, sx, sy = (torch.empty(R * n, dtype=torch.float32, device=dev) for _ in range(4))
    for r in range(R):
        sl = slice(r * n, (r + 1) * n)
        synth_dev.sample_scene(scene, n, seed=1000 + 2 * r, sigma=sigma, out=(tx[sl], ty[sl]))
        synth_dev.sample_scene(scene, n, seed=1001 + 2 * r, sigma=sigma, pose=TRUE_POSE, out=(sx[sl], sy[sl]))
    off = torch.arange(R + 1, dtype=torch.int64, device=dev) * n
    init = torch.zeros((R, 3), dtype=torch.float64, device=dev)
    with NdtBatch2D() as b:                    # the estimator: Gauss-Newton iterations to convergence
        rows = b.decode(b.align_dev(tx, ty, off, sx, sy, off, init))
    assert all(r.status == 0 for r in rows)
    est = np.array([r.pose for r in rows])
    if mode == 1:
        # the Newton form of the Hessian at the converged poses: one evaluation per pair, started there
        # (Newton *iterations* from 0.1 m off are chaotic; the form of H is what is calibrated)
        at = torch.from_numpy(est.copy()).to(dev)
        with NdtBatch2D(hessian_mode=1, fixed_iterations=1) as b:
            rows = b.decode(b.align_dev(tx, ty, off, sx, sy, off, at))
    return rows, est, np.cov(est.T)


@gpu
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", [0, 1])
def test_calibrated_covariance_matches_the_empirical_scatter(gpu_lib, case, mode):
    rows, est, C = _monte_carlo(*case, mode)
    assert np.abs(est.mean(0) - np.array(TRUE_POSE)).max() < 2e-3                 # and the estimator is unbiased at that level
    worst_lo, worst_hi, raw = np.inf, 0.0, []
    for r in rows[:60]:                        # per realisation, as a caller gets it
        ev = np.linalg.eigvals(np.linalg.solve(r.covariance(mode), C)).real
        worst_lo, worst_hi = min(worst_lo, ev.min()), max(worst_hi, ev.max())
        raw.append(np.linalg.eigvals(np.linalg.solve(np.linalg.inv(r.H), C)).real.min())
    # every direction of the calibrated covariance within a factor 3 of the empirical one
    assert 1.0 / 3.0 < worst_lo and worst_hi < 3.0, (case, mode, worst_lo, worst_hi)
    # the uncalibrated H^-1 is too small in EVERY direction: by more than 1.8x (Newton form) / 4x (Gauss-Newton
    # form) in variance in its best direction, 8x / 18x in its worst
    assert min(raw) > (1.8 if mode == 1 else 4.0), (case, mode, min(raw))


def test_calibration_function_is_s_hinv_s(ndt_lib):
    import ctypes as C
    from gtsam_ndt_amd import _lib as L
    rng = np.random.default_rng(3)
    A = rng.normal(size=(3, 3))
    H = A @ A.T + 3.0 * np.eye(3)
    for mode, kt, kr in ((0, 10.0, 18.0), (1, 3.9, 7.5)):
        out = np.zeros(9)
        st = ndt_lib.ndt2d_calibrated_covariance(H.ctypes.data_as(C.POINTER(C.c_double)), mode,
                                                 out.ctypes.data_as(C.POINTER(C.c_double)))
        S = np.diag(np.sqrt([kt, kt, kr]))
        assert st == 0 and np.allclose(out.reshape(3, 3), S @ np.linalg.inv(H) @ S, rtol=1e-12)
    bad = np.diag([1.0, -1.0, 1.0])
    out = np.ones(9)
    assert ndt_lib.ndt2d_calibrated_covariance(bad.ctypes.data_as(C.POINTER(C.c_double)), 0,
                                               out.ctypes.data_as(C.POINTER(C.c_double))) == L.NDT_DEGENERATE_HESSIAN
    assert not out.any()


@gpu
def test_3d_inverse_hessian_against_the_empirical_scatter(gpu_lib):
    """The same Monte-Carlo in 3D (config-5 scans; the pair is placed at a random pose in the room every time, so the
    beams sample other surface points, fresh range noise; all alignments in one ndt3d_batch call).  Unlike 2D, the
    plain H^-1 of the Gauss-Newton form is a fair covariance at this density (48 points per occupied voxel): every
    direction within a factor 0.2 .. 5 of the empirical one, so MatchResult3::covariance needs no inflation there
    (docs/ALGORITHM.md section 2.9 has the sparse and noisy cases, where it does)."""
    import math
    import torch
    from gtsam_ndt_amd import synth3d, synth_dev
    from gtsam_ndt_amd.matcher import NdtBatch3D
    R3, n_elev, n_azim = 160, 64, 2048
    n = n_elev * n_azim
    rel = tuple(0.5 * np.array(synth3d.T_STAR_3D))
    R_rel, t_rel = synth3d.rotation(*rel[3:]), np.array(rel[:3])
    euler = lambda M: (math.atan2(M[2, 1], M[2, 2]), -math.asin(M[2, 0]), math.atan2(M[1, 0], M[0, 0]))
    rng = np.random.default_rng(5)
    dev = torch.device("cuda:0")
    t = [torch.empty(R3 * n, dtype=torch.float32, device=dev) for _ in range(3)]
    s = [torch.empty(R3 * n, dtype=torch.float32, device=dev) for _ in range(3)]
    for r in range(R3):
        A = (rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(-0.05, 0.05), rng.uniform(-0.01, 0.01),
             rng.uniform(-0.01, 0.01), rng.uniform(-0.3, 0.3))
        RA = synth3d.rotation(*A[3:])
        B = tuple(np.array(A[:3]) + RA @ t_rel) + euler(RA @ R_rel)          # B = A o rel: the relative pose is `rel` every time
        sl = slice(r * n, (r + 1) * n)
        synth_dev.lidar_scan3d(5000 + 2 * r, A, n_elev, n_azim, 0.02, out=tuple(c[sl] for c in t))
        synth_dev.lidar_scan3d(5001 + 2 * r, B, n_elev, n_azim, 0.02, out=tuple(c[sl] for c in s))
    off = torch.arange(R3 + 1, dtype=torch.int64, device=dev) * n
    init = torch.zeros((R3, 6), dtype=torch.float64, device=dev)
    with NdtBatch3D() as b:
        rows = b.decode(b.align_dev(t, off, s, off, init))
    assert all(r.status == 0 for r in rows)
    est = np.array([r.pose for r in rows])
    assert np.abs(est.mean(0) - np.array(rel)).max() < 2e-3
    C = np.cov(est.T)
    lo, hi = np.inf, 0.0
    for r in rows[:40]:
        ev = np.linalg.eigvals(np.linalg.solve(np.linalg.inv(r.H), C)).real
        lo, hi = min(lo, ev.min()), max(hi, ev.max())
    assert 0.2 < lo and hi < 5.0, (lo, hi)
