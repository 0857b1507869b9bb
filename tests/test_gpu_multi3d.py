"""ndt3d_align_multi_scan_dev / ndt3d_align_multi_start_dev: many 3D alignments against one cached voxel grid in
one launch chain (k_multi_solve3 + k_multi_body3).  Every start must equal its single-call alignment bit for bit
(same thread -> point assignment, same reductions, same update), in fixed and in converged mode."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth3d

pytestmark = pytest.mark.gpu

POSES = [(0.30, -0.20, 0.05, 0.01, -0.01, 0.03), (-0.25, 0.15, -0.04, -0.008, 0.012, -0.02),
         (0.10, 0.28, 0.02, 0.0, 0.015, 0.035), (-0.12, -0.22, 0.06, 0.012, 0.0, -0.03),
         (0.22, 0.05, -0.03, -0.01, -0.012, 0.015), (0.05, -0.05, 0.0, 0.0, 0.0, 0.01),
         (0.0, 0.0, 0.0, 0.0, 0.0, 0.0), (0.15, 0.15, 0.03, 0.005, 0.005, -0.01)]
SHAPES = [(32, 512), (16, 256), (32, 300), (24, 384), (32, 256), (16, 512), (32, 512), (20, 400)]


@pytest.fixture(scope="module")
def world():
    import torch
    target = synth3d.lidar_scan(101, (0.0,) * 6, 32, 1024, 0.02)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    t = tuple(f(target[:, a]) for a in range(3))
    scans = []
    for k, (p, (e, a)) in enumerate(zip(POSES, SHAPES)):
        s = synth3d.lidar_scan(300 + k, p, e, a, 0.02)
        scans.append(tuple(torch.from_numpy(f(s[:, c])).cuda() for c in range(3)))
    return t, scans


def _equal(a, b):
    assert a.status == b.status and a.iterations == b.iterations and a.n_hit == b.n_hit, (a, b)
    assert a.pose == b.pose and np.array_equal(a.H, b.H) and np.array_equal(a.g, b.g) and a.score == b.score


@pytest.mark.parametrize("kw", [dict(), dict(fixed_iterations=12), dict(line_search=4, step_scale=1.5)])
def test_multi_scan_equals_single_calls(gpu_lib, world, kw):
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    t, scans = world
    inits = [(0.0,) * 6] * len(scans)
    with NdtMatcher3D(**kw) as m:
        m.set_target(*t)
        single = [m.align(*s, i) for s, i in zip(scans, inits)]
        for count in (1, 3, len(scans)):
            multi = m.align_multi_scan(scans[:count], inits[:count])
            for a, b in zip(multi, single):
                _equal(a, b)
        again = m.align_multi_scan(scans, inits)                      # the context is reusable, results reproducible
        for a, b in zip(again, single):
            _equal(a, b)
    assert sum(r.status == 0 for r in single) >= 6 or "fixed_iterations" in kw
    if not kw:                                                        # and they are the right answers
        for r, p in zip(single, POSES):
            if r.status == 0:
                e = np.abs(np.array(r.pose) - np.array(p))
                assert e[:3].max() < 0.03 and e[3:].max() < 5e-3, (r.pose, p)


def test_multi_start_newton_and_a_start_that_misses(gpu_lib, world):
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    t, scans = world
    s = scans[0]
    with NdtMatcher3D() as g:
        g.set_target(*t)
        opt = np.array(g.align(*s, (0.0,) * 6).pose)
    rng = np.random.default_rng(4)
    starts = [tuple(opt + 2e-3 * rng.uniform(-1, 1, 6) * np.array([1, 1, 1, 0.2, 0.2, 0.2])) for _ in range(5)]
    starts.append((300.0, 300.0, 0.0, 0.0, 0.0, 0.0))                 # far outside the map: NDT_TOO_FEW_HITS at once
    starts.append(tuple(opt))
    for mode in (0, 1):
        with NdtMatcher3D(hessian_mode=mode) as m:
            m.set_target(*t)
            single = [m.align(*s, p) for p in starts]
            multi = m.align_multi_start(*s, starts)
            for a, b in zip(multi, single):
                _equal(a, b)
            assert multi[5].status == 3 and multi[5].iterations == 0
            assert multi[6].status == 0


def test_multi_scan_64_scans_fixed_mode(gpu_lib, world):
    """The full width of the chain: 64 starts (8 scans x 8 guesses), 6 fixed iterations."""
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    t, scans = world
    rng = np.random.default_rng(9)
    pick = [scans[k % len(scans)] for k in range(64)]
    inits = [tuple(0.02 * rng.uniform(-1, 1, 6) * np.array([1, 1, 1, 0.1, 0.1, 0.1])) for _ in range(64)]
    with NdtMatcher3D(fixed_iterations=6) as m:
        m.set_target(*t)
        multi = m.align_multi_scan(pick, inits)
        for k in (0, 7, 31, 63):
            _equal(multi[k], m.align(*pick[k], inits[k]))
    assert all(r.iterations == 6 for r in multi)
