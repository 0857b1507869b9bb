"""The 3D path as a SLAM front end drives it: 64-beam scans of a moving sensor (ray cast against the box room),
scan-to-submap alignment, incremental voxel-grid update with the estimated pose - on the device, scan after scan
(ndt3d_reserve_target, ndt3d_align_dev, ndt3d_add_target_points_dev).  Each estimate is checked against the
generating pose, each merge against the host entry point fed with a float32 restatement of the device's
transform (grids bit for bit), the final voxel counts against numpy."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth3d

pytestmark = pytest.mark.gpu

POSES = [(0.0, 0.0, 0.0, 0.0, 0.0, 0.0), (0.6, 0.3, 0.02, 0.004, -0.006, 0.05), (1.3, 0.5, -0.01, -0.005, 0.004, 0.11),
         (1.9, 1.1, 0.03, 0.006, 0.002, 0.16), (2.4, 1.8, 0.0, -0.003, -0.005, 0.22)]


def _world(x, y, z, pose):
    """ndt3d_add_target_points_dev's float32 restatement: R in float64 rounded to float32, rows summed left to right."""
    R = synth3d.rotation(*pose[3:]).astype(np.float32)
    t = np.asarray(pose[:3], np.float32)
    out = []
    for r in range(3):
        a, b, c = R[r, 0] * x, R[r, 1] * y, R[r, 2] * z
        out.append((((a + b) + c) + t[r]).astype(np.float32))
    return out


def _scan(k, pose):
    p = synth3d.lidar_scan(200 + k, pose, n_elev=32, n_azim=1024, sigma=0.02)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return f(p[:, 0]), f(p[:, 1]), f(p[:, 2])          # sensor frame; the sensor sits SENSOR_Z above the pose's origin


def test_scan_to_submap_sequence_3d(gpu_lib):
    import torch
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    from oracle import ndt3d as o
    prm = o.Ndt3Params()
    lo, hi = (-22.0, -22.0, -3.0), (22.0, 22.0, 6.0)        # the room in the first sensor's frame, with margin
    cuda = lambda a: torch.from_numpy(a).cuda()
    T = None
    with NdtMatcher3D() as m, NdtMatcher3D() as host:
        info = m.reserve_target(lo, hi)
        host.reserve_target(lo, hi)
        assert info.n_valid == 0 and m.align(*_scan(0, POSES[0]), POSES[0]).status == 4      # NDT_TOO_FEW_CELLS
        for k, p in enumerate(POSES):
            x, y, z = _scan(k, p)
            if k == 0:
                assert m.add_target_points(cuda(x), cuda(y), cuda(z), pose=p) == 0
                T = _world(x, y, z, p)
                assert host.add_target_points(*T) == 0
            else:
                guess = tuple(np.array(p) + np.array([0.05, -0.04, 0.01, 0.002, -0.002, 0.01]))      # odometry-grade guess
                got = m.align(cuda(x), cuda(y), cuda(z), guess)
                # (the oracle derives its grid from the points' bounding box, the submap lives in the reserved extent:
                # another voxelisation of the same map, so the estimate is checked against the generating pose)
                assert got.status == 0
                e = np.abs(np.array(got.pose) - np.array(p))
                assert e[:3].max() < 0.02 and e[3:].max() < 3e-3, (k, got.pose, p)
                # ... and the same grid on both sides for the merge: transform on the device vs on the host
                est = got.pose
                out_d = m.add_target_points(cuda(x), cuda(y), cuda(z), pose=est)
                W = _world(x, y, z, est)
                out_h = host.add_target_points(*W)
                assert out_d == out_h
                T = [np.concatenate([a, b]) for a, b in zip(T, W)]
            for u, v in zip(m.grid(), host.grid()):
                np.testing.assert_array_equal(u, v)
        # the incrementally grown submap holds exactly the points' voxel counts (extent = the reserved box)
        count, mean, icov = m.grid()
        gi = m.grid_info()
        P = np.stack(T, axis=1)
        idx = np.floor((P - np.array([gi.ox, gi.oy, gi.oz], np.float32)) * np.float32(gi.inv_cell)).astype(np.int64)
        inside = np.all((idx >= 0) & (idx < np.array([gi.width, gi.height, gi.depth])), axis=1)
        key = (idx[:, 2] * gi.height + idx[:, 1]) * gi.width + idx[:, 0]
        ref = np.bincount(key[inside], minlength=gi.width * gi.height * gi.depth)
        np.testing.assert_array_equal(count.astype(np.int64), ref)
        x, y, z = _scan(9, POSES[2])
        H, g, s, nh = m.evaluate(x, y, z, POSES[2])
        assert nh > 0.8 * x.size


def test_reserved_voxel_grid_filled_in_chunks_equals_one_shot_build(gpu_lib):
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    d = synth3d.make_pair3d(n_elev=32, n_azim=512)
    t = (d["tx"], d["ty"], d["tz"])
    with NdtMatcher3D() as a, NdtMatcher3D() as b:
        ia = a.set_target(*t)
        ib = b.reserve_target([c.min() for c in t], [c.max() for c in t])
        assert (ia.width, ia.height, ia.depth, ia.ox, ia.oy, ia.oz) == (ib.width, ib.height, ib.depth, ib.ox, ib.oy, ib.oz)
        for part in np.array_split(np.arange(t[0].size), 4):
            assert b.add_target_points(*(c[part] for c in t)) == 0
        assert b.grid_info().n_valid == ia.n_valid
        for u, v in zip(a.grid(), b.grid()):
            np.testing.assert_array_equal(u, v)
        ra, rb = a.align(d["sx"], d["sy"], d["sz"], d["init"]), b.align(d["sx"], d["sy"], d["sz"], d["init"])
        assert ra.pose == rb.pose and np.array_equal(ra.H, rb.H)


def test_range_image_to_points_kernel(gpu_lib):
    """ndt3d_range_image_to_points_dev against the float64 direction formula of synth3d.lidar_scan, with no-return
    cells; the converted scan aligns like the host-converted one."""
    import torch
    from gtsam_ndt_amd.matcher import NdtMatcher3D, range_image_to_points
    n_elev, n_azim = 32, 1024
    el = np.deg2rad(np.linspace(-24.0, 20.0, n_elev))
    az0, inc = 0.5 * (2.0 * np.pi / n_azim), 2.0 * np.pi / n_azim
    p = synth3d.lidar_scan(300, POSES[1], n_elev=n_elev, n_azim=n_azim, sigma=0.02)         # ring-major, azimuth inner
    rng = np.linalg.norm(p, axis=1).astype(np.float32).reshape(n_elev, n_azim)
    rng[3, 10:20] = np.nan; rng[7, 5] = 0.01; rng[9, 100] = 500.0                            # no return, too near, too far
    x, y, z = (t.cpu().numpy() for t in range_image_to_points(torch.from_numpy(rng).cuda(), el, az0, inc, 0.05, 100.0))
    E, A = np.meshgrid(el, az0 + inc * np.arange(n_azim), indexing="ij")
    r64 = rng.astype(np.float64)
    ref = np.stack([r64 * np.cos(E) * np.cos(A), r64 * np.cos(E) * np.sin(A), r64 * np.sin(E)], axis=-1).reshape(-1, 3)
    bad = ~(np.isfinite(rng) & (rng >= 0.05) & (rng <= 100.0)).reshape(-1)
    assert bad.sum() == 12 and np.array_equal(np.isnan(x), bad) and np.array_equal(np.isnan(z), bad)
    got = np.stack([x, y, z], axis=1)
    np.testing.assert_allclose(got[~bad], ref[~bad], rtol=0, atol=4e-6)
    d = synth3d.make_pair3d(n_elev=n_elev, n_azim=n_azim, pose=POSES[1])
    with NdtMatcher3D() as m:
        m.set_target(d["tx"], d["ty"], d["tz"])
        guess = tuple(np.array(POSES[1]) + np.array([0.05, -0.04, 0.01, 0.002, -0.002, 0.01]))
        a = m.align(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(z).cuda(), guess)
    e = np.abs(np.array(a.pose) - np.array(POSES[1]))
    assert a.status == 0 and e[:3].max() < 0.02 and e[3:].max() < 3e-3, a.pose
