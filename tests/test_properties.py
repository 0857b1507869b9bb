"""Property tests: invariances the domain offers, checked on the oracle on
CPU and on the HIP path on the GPU, at sizes up to BASELINE's full configurations."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth
from oracle import ndt2d as o


def _rigid(x, y, pose):
    c, s = np.cos(pose[2]), np.sin(pose[2])
    return (c * x - s * y + pose[0]).astype(np.float32), (s * x + c * y + pose[1]).astype(np.float32)


def test_oracle_source_permutation_invariance():
    d = synth.make_pair(1)
    prm = o.NdtParams()
    g = o.build_grid(d["tx"], d["ty"], prm)
    perm = np.random.default_rng(3).permutation(len(d["sx"]))
    a = o.align(g, d["sx"], d["sy"], d["init"], prm)
    b = o.align(g, d["sx"][perm], d["sy"][perm], d["init"], prm)
    assert np.abs(np.array(a["pose"]) - np.array(b["pose"])).max() < 1e-9


def test_oracle_joint_translation_leaves_relative_pose():
    """Shifting both clouds by a whole number of cells leaves the relative pose unchanged
    (the grid is anchored to multiples of the cell size)."""
    d = synth.make_pair(1)
    prm = o.NdtParams()
    shift = (7.0, -3.5)            # multiples of 0.5 m
    a = o.align(o.build_grid(d["tx"], d["ty"], prm), d["sx"], d["sy"], d["init"], prm)
    tx, ty = d["tx"] + np.float32(shift[0]), d["ty"] + np.float32(shift[1])
    # source frame unchanged; the pose that maps it onto the shifted target is shifted too
    b = o.align(o.build_grid(tx, ty, prm), d["sx"], d["sy"], (d["init"][0] + shift[0], d["init"][1] + shift[1], 0.0), prm)
    e = np.array(b["pose"]) - np.array(a["pose"]) - np.array([shift[0], shift[1], 0.0])
    assert np.abs(e).max() < 2e-4


@pytest.mark.gpu
def test_gpu_source_permutation_and_chunking(gpu_lib):
    """Any source order gives the same converged pose (to reduction rounding), at config 2 size."""
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(2)
    perm = np.random.default_rng(5).permutation(len(d["sx"]))
    with NdtMatcher2D() as m:
        m.set_target(d["tx"], d["ty"])
        a = m.align(d["sx"], d["sy"], d["init"])
        b = m.align(d["sx"][perm], d["sy"][perm], d["init"])
    assert a.status == 0 == b.status
    assert np.abs(np.array(a.pose) - np.array(b.pose)).max() < 1e-6
    assert a.n_hit == b.n_hit


@pytest.mark.gpu
def test_gpu_full_size_round_trip_config3(gpu_lib):
    """Full BASELINE config 3 (1M-point target, 100k-point source): move the source by a known
    rigid transform and the recovered pose composes back (size-independent property; the oracle
    is not needed at this size)."""
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(3)
    extra = (0.03, -0.02, 0.004)
    # source' = extra^-1 (source): aligning source' must give pose o extra
    c, s = np.cos(extra[2]), np.sin(extra[2])
    dx, dy = d["sx"] - np.float32(extra[0]), d["sy"] - np.float32(extra[1])
    sx2, sy2 = (c * dx + s * dy).astype(np.float32), (-s * dx + c * dy).astype(np.float32)
    with NdtMatcher2D() as m:
        info = m.set_target(d["tx"], d["ty"])
        assert info.n_points == 1_000_000
        a = m.align(d["sx"], d["sy"], d["init"])
        b = m.align(sx2, sy2, d["init"])
    assert a.status == 0 == b.status
    ca, sa = np.cos(a.pose[2]), np.sin(a.pose[2])
    comp = (a.pose[0] + ca * extra[0] - sa * extra[1], a.pose[1] + sa * extra[0] + ca * extra[1], a.pose[2] + extra[2])
    e = np.abs(np.array(b.pose) - np.array(comp))
    assert e[0] < 2e-4 and e[1] < 2e-4 and e[2] < 2e-5
    assert np.abs(np.array(a.pose) - np.array(d["pose"])).max() < 2e-3       # known transform recovered


@pytest.mark.gpu
def test_gpu_incremental_submap_full_size(gpu_lib):
    """1M-point submap built in ten 100k-point additions equals the one-shot build bit for bit."""
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(3)
    n = len(d["tx"])
    order = np.argsort(np.arange(n) % 10, kind="stable")
    with NdtMatcher2D() as m:
        m.set_target(d["tx"], d["ty"])
        full = m.grid()
        # seed the extent with the extreme points, then stream the rest in ten chunks
        ext = np.unique([np.argmin(d["tx"]), np.argmax(d["tx"]), np.argmin(d["ty"]), np.argmax(d["ty"])])
        rest = np.setdiff1d(order, ext, assume_unique=False)
        m.set_target(d["tx"][ext], d["ty"][ext])
        for chunk in np.array_split(rest, 10):
            assert m.add_target_points(d["tx"][chunk], d["ty"][chunk]) == 0
        inc = m.grid()
    for u, v in zip(full, inc):
        np.testing.assert_array_equal(u, v)


def test_oracle_is_invariant_under_whole_cell_translations():
    """Moving the target and the pose by whole cells moves the grid with them: the cell statistics,
    score, gradient and Hessian are unchanged (the origin is snapped to multiples of the cell size,
    so nothing depends on where the map sits).  Exact multiples of 0.5 m in float32 keep the points
    exactly representable."""
    from oracle import ndt2d as o
    d = synth.make_pair(2, n_tgt=8000, n_src=4000)
    prm = o.NdtParams()
    g0 = o.build_grid(d["tx"], d["ty"], prm)
    H0, gr0, s0, n0 = o.evaluate(g0, d["sx"], d["sy"], d["pose"], prm)
    for kx, ky in ((8, -6), (-40, 64), (1, 1)):
        dx, dy = np.float32(0.5 * kx), np.float32(0.5 * ky)
        tx, ty = d["tx"] + dx, d["ty"] + dy
        exact = np.all((tx - dx) == d["tx"]) and np.all((ty - dy) == d["ty"])          # the shift was lossless
        g1 = o.build_grid(tx, ty, prm)
        assert (g1.W, g1.H, g1.n_valid) == (g0.W, g0.H, g0.n_valid)
        np.testing.assert_array_equal(g1.count, g0.count)
        pose = (d["pose"][0] + float(dx), d["pose"][1] + float(dy), d["pose"][2])
        H1, gr1, s1, n1 = o.evaluate(g1, d["sx"], d["sy"], pose, prm)
        assert n1 == n0
        tol = 1e-9 if exact else 1e-4
        assert abs(s1 - s0) <= tol * s0 and np.abs(H1 - H0).max() <= tol * np.abs(H0).max()
        assert np.abs(gr1 - gr0).max() <= tol * np.sqrt(np.abs(np.diag(H0)).max() * s0)


def test_oracle_rotation_of_both_clouds_by_a_quarter_turn():
    """A quarter turn maps the cell lattice onto itself: aligning the rotated pair from the rotated
    initial guess ends at the rotated pose (to the convergence wobble)."""
    from oracle import ndt2d as o
    d = synth.make_pair(2, n_tgt=8000, n_src=8000)
    prm = o.NdtParams()
    r0 = o.align(o.build_grid(d["tx"], d["ty"], prm), d["sx"], d["sy"], d["init"], prm)
    # world frame rotated by +90 degrees: (x, y) -> (-y, x); the source frame is left alone
    tx, ty = (-d["ty"]).astype(np.float32), d["tx"].copy()
    init = (-d["init"][1], d["init"][0], d["init"][2] + np.pi / 2)
    r1 = o.align(o.build_grid(tx, ty, prm), d["sx"], d["sy"], init, prm)
    assert r0["status"] == r1["status"] == o.NDT_OK
    want = np.array([-r0["pose"][1], r0["pose"][0], o.wrap_angle(r0["pose"][2] + np.pi / 2)])
    assert np.abs(np.array(r1["pose"]) - want).max() < 5e-4


@pytest.mark.gpu
def test_gpu_whole_cell_translation_full_size(gpu_lib):
    """Config 3 (1M-point submap) moved by whole cells: identical cell counts, the converged pose
    moves by exactly the shift (to the float32 resolution of the larger coordinates)."""
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(3)
    dx, dy = np.float32(64.0), np.float32(-32.0)                 # 128 and -64 cells of 0.5 m
    with NdtMatcher2D() as m:
        m.set_target(d["tx"], d["ty"])
        c0 = m.grid()[0]
        info0 = m.grid_info()
        a = m.align(d["sx"], d["sy"], d["init"])
        m.set_target(d["tx"] + dx, d["ty"] + dy)
        c1 = m.grid()[0]
        info1 = m.grid_info()
        b = m.align(d["sx"], d["sy"], (d["init"][0] + 64.0, d["init"][1] - 32.0, d["init"][2]))
    assert (info0.width, info0.height, info0.n_valid) == (info1.width, info1.height, info1.n_valid)
    assert info1.ox - info0.ox == 64.0 and info1.oy - info0.oy == -32.0
    lossless = np.all((d["tx"] + dx) - dx == d["tx"]) and np.all((d["ty"] + dy) - dy == d["ty"])
    if lossless:
        np.testing.assert_array_equal(c0, c1)
    else:                                                        # a few points round across a cell edge
        assert np.abs(c0.astype(np.int64) - c1.astype(np.int64)).sum() <= 2e-4 * c0.sum()
    assert a.status == 0 == b.status
    e = np.abs(np.array(b.pose) - (np.array(a.pose) + np.array([64.0, -32.0, 0.0])))
    assert e[0] < 2e-4 and e[1] < 2e-4 and e[2] < 2e-5
