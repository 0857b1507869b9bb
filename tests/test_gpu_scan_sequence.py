"""The path as a SLAM front end drives it: range/bearing scans of a moving planar lidar (ray
cast against the scene: range-dependent density, occlusion) -> ndt2d_polar_to_points_dev ->
scan-to-submap alignment -> incremental submap update, scan after scan.  Each step is compared
with the CPU oracle given the very same float32 points."""
import math

import numpy as np
import pytest

from gtsam_ndt_amd import synth

pytestmark = pytest.mark.gpu

POSES = [(8.0 + 1.5 * k, 10.0 + 0.8 * k, 0.1 * k) for k in range(6)]


def _world(x, y, pose):
    c, s = np.float32(math.cos(pose[2])), np.float32(math.sin(pose[2]))
    return ((c * x - s * y) + np.float32(pose[0])).astype(np.float32), ((s * x + c * y) + np.float32(pose[1])).astype(np.float32)


def test_scan_to_submap_sequence_matches_oracle(gpu_lib):
    import torch
    from gtsam_ndt_amd.matcher import NdtMatcher2D, polar_to_points
    from oracle import ndt2d as o
    sc = synth.room_scene(4242, 30.0)
    prm = o.NdtParams()
    tx = ty = None
    with NdtMatcher2D() as m:
        for k, p in enumerate(POSES):
            r, a0, da = synth.lidar_scan2d(sc, p, n_beams=7200, seed=100 + k)
            dx, dy = polar_to_points(torch.from_numpy(r).cuda(), a0, da, 0.05, 30.0)     # device conversion
            x, y = dx.cpu().numpy(), dy.cpu().numpy()
            hx, hy = synth.scan_points(r, a0, da)
            assert np.array_equal(np.isnan(x), np.isnan(hx))
            np.testing.assert_allclose(x[~np.isnan(x)], hx[~np.isnan(hx)], rtol=0, atol=4e-6)
            np.testing.assert_allclose(y[~np.isnan(y)], hy[~np.isnan(hy)], rtol=0, atol=4e-6)
            if k == 0:
                ok = ~np.isnan(x)
                tx, ty = _world(x[ok], y[ok], p)
                lo, hi = (tx.min(), ty.min()), (tx.max(), ty.max())
                m.set_target(tx, ty)
                continue
            guess = (p[0] + 0.05, p[1] - 0.04, p[2] + 0.01)             # odometry-grade initial guess
            # the device gets the scan as converted, NaN points (no return) included: they are ignored
            got = m.align(dx, dy, guess)
            ok = ~np.isnan(x)
            ref = o.align(o.build_grid(tx, ty, prm), x[ok], y[ok], guess, prm)
            assert got.status == 0 == ref["status"]
            e = np.abs(np.array(got.pose) - np.array(ref["pose"]))
            assert e.max() < 1e-4, (k, got.pose, ref["pose"])                # BASELINE: 1e-4 m / 1e-4 rad
            assert abs(got.iterations - ref["iterations"]) <= 3 and got.n_hit == ref["n_hit"]
            assert np.abs(np.array(got.pose) - np.array(p)).max() < 5e-3     # and it is the right answer
            # submap update with the estimate (the oracle's, so that both sides keep identical inputs)
            wx, wy = _world(x[ok], y[ok], ref["pose"])
            inb = (wx >= lo[0]) & (wx <= hi[0]) & (wy >= lo[1]) & (wy <= hi[1])
            assert m.add_target_points(wx[inb], wy[inb]) == 0
            tx, ty = np.concatenate([tx, wx[inb]]), np.concatenate([ty, wy[inb]])
        # the incrementally grown submap is the grid a rebuild from all points gives
        g = o.build_grid(tx, ty, prm)
        count, mean, icov = m.grid()
        np.testing.assert_array_equal(count.astype(np.int64), g.count)
        info = m.grid_info()
        assert (info.width, info.height, info.n_valid) == (g.W, g.H, g.n_valid)


def test_device_side_submap_update_equals_host_side(gpu_lib):
    """ndt2d_add_target_points_dev with a pose = transforming on the host (same float32 operation
    order) and merging through the host entry point: identical grids, identical outside counts."""
    import torch
    from gtsam_ndt_amd.matcher import NdtMatcher2D, polar_to_points
    sc = synth.room_scene(4242, 30.0)
    r0, a0, da = synth.lidar_scan2d(sc, POSES[0], n_beams=7200, seed=100)
    x0, y0 = synth.scan_points(r0, a0, da)
    ok0 = ~np.isnan(x0)
    tx, ty = _world(x0[ok0], y0[ok0], POSES[0])
    r1, a1, da1 = synth.lidar_scan2d(sc, POSES[3], n_beams=7200, seed=103)
    dx, dy = polar_to_points(torch.from_numpy(r1).cuda(), a1, da1, 0.05, 30.0)
    x1, y1 = dx.cpu().numpy(), dy.cpu().numpy()
    pose = (POSES[3][0] + 0.003, POSES[3][1] - 0.002, POSES[3][2] + 0.0004)
    wx, wy = _world(x1, y1, pose)                                   # NaN beams stay NaN
    with NdtMatcher2D() as a, NdtMatcher2D() as b:
        a.set_target(tx, ty)
        b.set_target(tx, ty)
        out_a = a.add_target_points(wx, wy)
        out_b = b.add_target_points(dx, dy, pose=pose)
        assert out_a == out_b
        ca, ma, ia = a.grid()
        cb, mb, ib = b.grid()
        assert ca.sum() > 1.5 * ok0.sum()                           # the second scan did go in
        np.testing.assert_array_equal(ca, cb)
        np.testing.assert_array_equal(ma, mb)
        np.testing.assert_array_equal(ia, ib)
        # without a pose the device points are merged as they are
        c0 = a.add_target_points(torch.from_numpy(tx).cuda(), torch.from_numpy(ty).cuda())
        d0 = b.add_target_points(tx, ty)
        assert c0 == d0 == 0
        np.testing.assert_array_equal(a.grid()[0], b.grid()[0])


def test_reserved_extent_then_incremental_fill_equals_one_shot_build(gpu_lib):
    """ndt2d_reserve_target: an empty grid over a chosen extent.  With the cloud's own bounding
    box it has the geometry ndt2d_set_target derives, so filling it chunk by chunk ends in the
    identical grid; before any cell is valid an alignment reports NDT_TOO_FEW_CELLS."""
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(2, n_tgt=60000, n_src=20000)
    tx, ty = d["tx"], d["ty"]
    with NdtMatcher2D() as a, NdtMatcher2D() as b:
        ia = a.set_target(tx, ty)
        ib = b.reserve_target(tx.min(), ty.min(), tx.max(), ty.max())
        assert (ia.width, ia.height, ia.ox, ia.oy) == (ib.width, ib.height, ib.ox, ib.oy) and ib.n_valid == 0
        r = b.align(d["sx"], d["sy"], d["init"])
        assert r.status == L.NDT_TOO_FEW_CELLS and r.iterations == 0
        for part in np.array_split(np.arange(tx.size), 5):
            assert b.add_target_points(tx[part], ty[part]) == 0
        assert b.grid_info().n_valid == ia.n_valid
        for u, v in zip(a.grid(), b.grid()):
            np.testing.assert_array_equal(u, v)
        ra, rb = a.align(d["sx"], d["sy"], d["init"]), b.align(d["sx"], d["sy"], d["init"])
        assert ra.pose == rb.pose and ra.iterations == rb.iterations
        # a larger reserved region: the same cloud lands in it whole, points beyond it are counted
        ic = b.reserve_target(tx.min() - 20.0, ty.min() - 20.0, tx.max() + 20.0, ty.max() + 20.0)
        assert ic.width > ia.width and ic.height > ia.height
        assert b.add_target_points(tx, ty) == 0
        assert b.add_target_points(tx[:100] + np.float32(500.0), ty[:100]) == 100
        rc = b.align(d["sx"], d["sy"], d["init"])
        assert rc.status == 0 and np.abs(np.array(rc.pose) - np.array(ra.pose)).max() < 2e-3   # other cell phase
        with pytest.raises(L.NdtError):
            b.reserve_target(1.0, 0.0, 0.0, 1.0)
        with pytest.raises(L.NdtError):
            b.reserve_target(0.0, 0.0, float("nan"), 1.0)
