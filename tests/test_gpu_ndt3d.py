"""3D SE(3) variant on the GPU vs the CPU oracle (BASELINE config 5; parity unpinned: the
oracle is this repo's own, the reference holds no code)."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth3d

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small():
    from oracle import ndt3d as o
    d = synth3d.make_pair3d(n_elev=32, n_azim=512)
    prm = o.Ndt3Params()
    return d, prm, o.build_grid3(d["tx"], d["ty"], d["tz"], prm)


def test_grid3d_parity(gpu_lib, small):
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    d, prm, g = small
    with NdtMatcher3D() as m:
        info = m.set_target(d["tx"], d["ty"], d["tz"])
        assert (info.width, info.height, info.depth) == g.dims and info.n_valid == g.n_valid
        assert (info.ox, info.oy, info.oz) == tuple(g.o)
        count, mean, icov = m.grid()
    np.testing.assert_array_equal(count.astype(np.int64), g.count)          # bit-exact integer work
    v = g.valid
    np.testing.assert_array_equal(icov[:, 0] != 0, v)
    np.testing.assert_allclose(mean[v], g.mean[v], rtol=0, atol=4e-6)
    nrm = np.linalg.norm(g.icov[v], axis=1, keepdims=True)
    assert np.max(np.abs(icov[v] - g.icov[v]) / nrm) < 2e-5


def test_evaluate3d_parity(gpu_lib, small):
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    from oracle import ndt3d as o
    d, prm, g = small
    with NdtMatcher3D() as m:
        m.set_target(d["tx"], d["ty"], d["tz"])
        for pose in (d["init"], d["pose"]):
            H, gr, s, nh = m.evaluate(d["sx"], d["sy"], d["sz"], pose)
            Hm, gm, sm, nm = o.evaluate3(g, d["sx"], d["sy"], d["sz"], pose, prm, mirror32=True)
            Ht, gt, st, nt = o.evaluate3(g, d["sx"], d["sy"], d["sz"], pose, prm)
            assert abs(nh - nm) <= 3 and abs(nh - nt) <= 5
            sc = np.sqrt(np.outer(np.diag(Hm), np.diag(Hm)))
            assert np.max(np.abs(H - Hm) / sc) < 2e-4
            assert abs(s - sm) / sm < 2e-4 and abs(s - st) / st < 5e-3
            gs = np.sqrt(np.diag(Hm) * sm)
            assert np.max(np.abs(gr - gm) / gs) < 1e-3


def test_align3d_converged_pose_parity(gpu_lib, small):
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    from oracle import ndt3d as o
    d, prm, g = small
    ref = o.align3(g, d["sx"], d["sy"], d["sz"], d["init"], prm)
    with NdtMatcher3D() as m:
        m.set_target(d["tx"], d["ty"], d["tz"])
        r = m.align(d["sx"], d["sy"], d["sz"], d["init"])
        r2 = m.align(d["sx"], d["sy"], d["sz"], d["init"])
    assert r.status == 0 == ref["status"]
    e = np.abs(np.array(r.pose) - np.array(ref["pose"]))
    assert e[:3].max() < 1e-4 and e[3:].max() < 1e-4             # BASELINE.json: 1e-4 m / 1e-4 rad
    assert abs(r.iterations - ref["iterations"]) <= 3
    assert r.pose == r2.pose and np.array_equal(r.H, r2.H)        # deterministic


def test_align3d_full_config5(gpu_lib):
    """The full 64 x 2048 = 131072-point pair of BASELINE config 5."""
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    from oracle import ndt3d as o
    d = synth3d.make_pair3d()
    prm = o.Ndt3Params(fixed_iterations=10)
    ref = o.align3(o.build_grid3(d["tx"], d["ty"], d["tz"], prm), d["sx"], d["sy"], d["sz"], d["init"], prm)
    with NdtMatcher3D(fixed_iterations=10) as m:
        m.set_target(d["tx"], d["ty"], d["tz"])
        r = m.align(d["sx"], d["sy"], d["sz"], d["init"])
    assert r.iterations == 10
    e = np.abs(np.array(r.pose) - np.array(ref["pose"]))
    assert e[:3].max() < 1e-4 and e[3:].max() < 1e-4
    with NdtMatcher3D() as m:
        m.set_target(d["tx"], d["ty"], d["tz"])
        r = m.align(d["sx"], d["sy"], d["sz"], d["init"])
    assert r.status == 0
    e = np.abs(np.array(r.pose) - np.array(d["pose"]))
    assert e[:3].max() < 5e-3 and e[3:].max() < 1e-3             # recovers the generating pose


def test_set_target3d_from_device_arrays(gpu_lib, small):
    import torch
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    d, prm, g = small
    with NdtMatcher3D() as a, NdtMatcher3D() as b:
        ia = a.set_target(d["tx"], d["ty"], d["tz"])
        ib = b.set_target(*(torch.from_numpy(d[k]).cuda() for k in ("tx", "ty", "tz")))
        assert (ia.width, ia.height, ia.depth, ia.n_valid) == (ib.width, ib.height, ib.depth, ib.n_valid)
        for u, v in zip(a.grid(), b.grid()):
            np.testing.assert_array_equal(u, v)


def test_incremental_target3d_equals_rebuild(gpu_lib, small):
    """ndt3d_add_target_points: the voxel grid built in pieces is bit for bit the one-shot grid (exact
    integer sums), points outside the cached extent are counted and ignored."""
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    d, prm, g = small
    n = len(d["tx"])
    ext = np.unique([f(d[k]) for k in ("tx", "ty", "tz") for f in (np.argmin, np.argmax)])
    rest = np.setdiff1d(np.arange(n), ext)
    with NdtMatcher3D() as m:
        m.set_target(d["tx"], d["ty"], d["tz"])
        full = m.grid()
        info_full = m.grid_info()
        m.set_target(d["tx"][ext], d["ty"][ext], d["tz"][ext])          # fixes the extent
        for part in np.array_split(rest, 4):
            assert m.add_target_points(d["tx"][part], d["ty"][part], d["tz"][part]) == 0
        assert m.add_target_points(d["tx"][:7] + np.float32(1e4), d["ty"][:7], d["tz"][:7]) == 7
        inc = m.grid()
        assert m.grid_info().n_valid == info_full.n_valid == g.n_valid
        r = m.align(d["sx"], d["sy"], d["sz"], d["init"])
    for u, v in zip(full, inc):
        np.testing.assert_array_equal(u, v)
    assert r.status == 0


def test_pyramid3d_widens_the_basin(gpu_lib):
    """From 1.2 m / 0.08 rad off, the 1 m grid alone ends elsewhere; 4 m -> 2 m -> 1 m recovers the pose."""
    from gtsam_ndt_amd.matcher import NdtMatcher3D, NdtPyramid3D
    d = synth3d.make_pair3d(n_elev=32, n_azim=1024)
    true = np.array(d["pose"])
    init = tuple(true + np.array([1.2, -0.9, 0.15, 0.0, 0.0, 0.08]))
    with NdtPyramid3D() as p:
        p.set_target(d["tx"], d["ty"], d["tz"])
        r = p.align(d["sx"], d["sy"], d["sz"], init)
    with NdtMatcher3D() as m:
        m.set_target(d["tx"], d["ty"], d["tz"])
        flat = m.align(d["sx"], d["sy"], d["sz"], init)
    e = np.abs(np.array(r.pose) - true)
    assert r.status == 0 and e[:3].max() < 0.03 and e[3:].max() < 5e-3, (r.pose, d["pose"])
    ef = np.abs(np.array(flat.pose) - true)
    assert ef[:3].max() > e[:3].max()


def test_async_entry_point_equals_the_synchronous_one(gpu_lib):
    """ndt3d_align_dev_async / ndt3d_align_finish: fixed-K chains enqueued back to back without a host
    round trip (the last one's result is fetched), and a converged-mode loop begun asynchronously."""
    import torch
    from gtsam_ndt_amd import synth3d
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    d = synth3d.make_pair3d(n_azim=512)
    s = [torch.from_numpy(d[k]).cuda() for k in ("sx", "sy", "sz")]
    for kw in (dict(fixed_iterations=12), dict()):
        with NdtMatcher3D(**kw) as m:
            m.set_target(d["tx"], d["ty"], d["tz"])
            want = m.align(*s, d["init"])
            for _ in range(3):
                m.align_async(*s, d["init"])
            got = m.finish()
            assert got.pose == want.pose and got.iterations == want.iterations and got.status == want.status
            assert np.array_equal(got.H, want.H)
            m.align_async(*s, d["init"])
            m.set_target(d["tx"], d["ty"], d["tz"])          # any other call finishes the loop in flight first
            assert m.align(*s, d["init"]).pose == want.pose


def test_newton_hessian_3d(gpu_lib, small):
    """hessian_mode = NDT_HESSIAN_NEWTON in 3D: the evaluation against the oracle's Newton form (float32
    mirror: tight) at poses near and 3 cm off the optimum, and Newton iterations from a start inside
    their basin - which is millimetres wide on this score (at 1 cm the Hessian is indefinite and the
    oracle's own Newton run wanders off), so the start is the Gauss-Newton optimum displaced by 2 mm:
    the "polish the converged pose" use."""
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    from oracle import ndt3d as o
    d, _, g = small
    prm = o.Ndt3Params(hessian_mode=1)
    gn = o.align3(g, d["sx"], d["sy"], d["sz"], d["init"], o.Ndt3Params())
    off3 = tuple(np.array(d["pose"]) + np.array([0.03, -0.02, 0.01, 0.002, -0.002, 0.004]))
    start = tuple(np.array(gn["pose"]) + 0.002 * np.array([1.0, -1.0, 0.5, 0.1, -0.1, 0.2]))
    with NdtMatcher3D(hessian_mode=1) as m:
        m.set_target(d["tx"], d["ty"], d["tz"])
        for pose in (gn["pose"], off3):
            H, gr, s, nh = m.evaluate(d["sx"], d["sy"], d["sz"], pose)
            Hm, gm, sm, nm = o.evaluate3(g, d["sx"], d["sy"], d["sz"], pose, prm, mirror32=True)
            Hgn = o.evaluate3(g, d["sx"], d["sy"], d["sz"], pose, o.Ndt3Params(), mirror32=True)[0]
            assert abs(nh - nm) <= 3
            sc = np.sqrt(np.outer(np.diag(Hgn), np.diag(Hgn)))
            assert np.max(np.abs(H - Hm) / sc) < 3e-4
            assert np.max(np.abs(Hm - Hgn) / sc) > 0.05                      # and it is not the Gauss-Newton one
            assert abs(s - sm) / sm < 2e-4
        r = m.align(d["sx"], d["sy"], d["sz"], start)
    ref = o.align3(g, d["sx"], d["sy"], d["sz"], start, prm)
    assert r.status == 0 == ref["status"] and ref["iterations"] <= 8 and abs(r.iterations - ref["iterations"]) <= 2
    e = np.abs(np.array(r.pose) - np.array(ref["pose"]))
    assert e[:3].max() < 1e-4 and e[3:].max() < 1e-4
    assert np.abs(np.array(r.pose) - np.array(gn["pose"]))[:3].max() < 1e-4          # the same optimum as Gauss-Newton's
    prm3 = o.Ndt3Params(hessian_mode=1, fixed_iterations=3)
    ref3 = o.align3(g, d["sx"], d["sy"], d["sz"], start, prm3)
    with NdtMatcher3D(hessian_mode=1, fixed_iterations=3) as m:
        m.set_target(d["tx"], d["ty"], d["tz"])
        r3 = m.align(d["sx"], d["sy"], d["sz"], start)
    assert r3.iterations == 3 and np.abs(np.array(r3.pose) - np.array(ref3["pose"])).max() < 1e-4


def test_3d_kernels_reduce_to_the_2d_kernels_on_planar_data(gpu_lib):
    """The 3D path (k_iterate3, 3D voxel build with Jacobi) and the 2D path (k_iterate, closed-form finalise)
    are separate kernels; on clouds in the plane z = 0 they must describe the same problem: same valid cells,
    the same (x, y, yaw) Hessian block and score, and the same converged pose (both within 1e-4 of the oracle)."""
    from gtsam_ndt_amd import synth
    from gtsam_ndt_amd.matcher import NdtMatcher2D, NdtMatcher3D
    from oracle import ndt2d as o2
    d = synth.make_pair(2, n_tgt=30000, n_src=20000)
    z, zs = np.zeros_like(d["tx"]), np.zeros_like(d["sx"])
    idx, other = [0, 1, 5], [2, 3, 4]
    with NdtMatcher2D(min_points=5) as m2, \
            NdtMatcher3D(cell_size=0.5, min_points=5, step_max_trans=0.5, min_hits=3) as m3:
        i2 = m2.set_target(d["tx"], d["ty"])
        i3 = m3.set_target(d["tx"], d["ty"], z)
        assert i3.n_valid == i2.n_valid and (i3.width, i3.height, i3.depth) == (i2.width, i2.height, 3)
        for pose in (d["init"], d["pose"]):
            H2, g2, s2, n2 = m2.evaluate(d["sx"], d["sy"], pose)
            H3, g3, s3, n3 = m3.evaluate(d["sx"], d["sy"], zs, (pose[0], pose[1], 0.0, 0.0, 0.0, pose[2]))
            assert n3 == n2 and abs(s3 - s2) <= 2e-5 * s2
            assert np.abs(H3[np.ix_(idx, idx)] - H2).max() <= 2e-5 * np.abs(H2).max()
            assert np.abs(H3[np.ix_(idx, other)]).max() <= 1e-6 * np.abs(H2).max() and np.abs(g3[other]).max() <= 1e-3
        r2 = m2.align(d["sx"], d["sy"], d["init"])
        r3 = m3.align(d["sx"], d["sy"], zs, (d["init"][0], d["init"][1], 0.0, 0.0, 0.0, d["init"][2]))
    prm = o2.NdtParams(min_points=5)
    ref = o2.align(o2.build_grid(d["tx"], d["ty"], prm), d["sx"], d["sy"], d["init"], prm)
    assert r2.status == r3.status == ref["status"] == 0
    assert np.abs(np.array(r3.pose)[idx] - np.array(r2.pose)).max() < 2e-5
    assert np.abs(np.array(r3.pose)[idx] - np.array(ref["pose"])).max() < 1e-4 and np.abs(np.array(r3.pose)[other]).max() < 1e-6
