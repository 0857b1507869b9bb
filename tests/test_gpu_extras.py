"""Rows either side of the hot path: Magnusson score constants, the
range/bearing -> Cartesian kernel, and the coarse-to-fine pyramid."""
import math

import numpy as np
import pytest

from gtsam_ndt_amd import synth


def test_magnusson_constants_formula(ndt_lib):
    from gtsam_ndt_amd.matcher import magnusson_constants
    from gtsam_ndt_amd import _lib as L
    import ctypes as C
    for p_o, c, dim in ((0.55, 1.0, 3), (0.3, 0.5, 2), (0.1, 2.0, 3)):
        c1 = 10.0 * (1.0 - p_o); c2 = p_o / c ** dim; d3 = -math.log(c2)
        md1 = -math.log(c1 + c2) - d3
        md2 = -2.0 * math.log((-math.log(c1 * math.exp(-0.5) + c2) - d3) / md1)
        d1, d2 = magnusson_constants(p_o, c, dim)
        assert abs(d1 + md1) < 1e-12 and abs(d2 - md2) < 1e-12 and d1 > 0 and d2 > 0
    d1, d2 = C.c_double(), C.c_double()
    assert ndt_lib.ndt_magnusson_constants(1.5, 1.0, 3, C.byref(d1), C.byref(d2)) == L.NDT_ERR_INVALID_ARG
    assert ndt_lib.ndt_magnusson_constants(0.5, 1.0, 4, C.byref(d1), C.byref(d2)) == L.NDT_ERR_INVALID_ARG


@pytest.mark.gpu
def test_polar_to_points_kernel(gpu_lib):
    import torch
    from gtsam_ndt_amd.matcher import polar_to_points
    n = 100_003
    rng = np.random.default_rng(0)
    r = rng.uniform(0.05, 30.0, n).astype(np.float32)
    r[::97] = np.inf; r[5::101] = np.nan; r[7::103] = 0.01; r[9::107] = 99.0
    a0, da = -math.pi, 2 * math.pi / n
    x, y = polar_to_points(torch.from_numpy(r).cuda(), a0, da, 0.05, 30.0)
    x, y = x.cpu().numpy(), y.cpu().numpy()
    ang = a0 + np.arange(n) * da
    ok = np.isfinite(r) & (r >= 0.05) & (r <= 30.0)
    assert np.array_equal(np.isnan(x), ~ok) and np.array_equal(np.isnan(y), ~ok)
    np.testing.assert_allclose(x[ok], (r[ok].astype(np.float64) * np.cos(ang[ok])).astype(np.float32), rtol=0, atol=4e-6)
    np.testing.assert_allclose(y[ok], (r[ok].astype(np.float64) * np.sin(ang[ok])).astype(np.float32), rtol=0, atol=4e-6)


@pytest.mark.gpu
def test_magnusson_score_aligns_like_oracle(gpu_lib):
    """d1, d2 from the mixture constants flow through kernels and oracle alike."""
    from gtsam_ndt_amd.matcher import NdtMatcher2D, magnusson_constants
    from oracle import ndt2d as o
    d = synth.make_pair(2, n_tgt=30000, n_src=30000)
    d1, d2 = magnusson_constants(0.3, 0.5, 2)
    prm = o.NdtParams(d1=d1, d2=d2)
    ref = o.align(o.build_grid(d["tx"], d["ty"], prm), d["sx"], d["sy"], d["init"], prm)
    with NdtMatcher2D(d1=d1, d2=d2) as m:
        m.set_target(d["tx"], d["ty"])
        r = m.align(d["sx"], d["sy"], d["init"])
    assert r.status == 0 == ref["status"]
    assert np.abs(np.array(r.pose) - np.array(ref["pose"])).max() < 1e-4
    assert abs(r.score - ref["score"]) / ref["score"] < 1e-3


@pytest.mark.gpu
def test_pyramid_widens_the_basin(gpu_lib):
    """From a 0.3 m / 0.05 rad offset the 0.5 m grid alone does not converge; the coarse-to-fine
    pyramid does, to the pose the oracle's pyramid reaches."""
    from gtsam_ndt_amd.matcher import NdtMatcher2D, NdtPyramid2D, PYRAMID_LEVELS
    from oracle import ndt2d as o
    sc = synth.room_scene(2, 50.0, -25.0, -25.0)
    xt, yt = synth.sample_scene(sc, 60000, 2 * 7919 + 11, 0.03)
    xs, ys = synth.sample_scene(sc, 60000, 2 * 7919 + 12, 0.03)
    pose = (0.30, -0.20, 0.05)
    xs, ys = synth.to_source_frame(xs, ys, pose)
    f = lambda a: a.astype(np.float32)
    xt, yt, xs, ys = map(f, (xt, yt, xs, ys))
    with NdtMatcher2D() as m:
        m.set_target(xt, yt)
        single = m.align(xs, ys, (0, 0, 0))
    assert single.status != 0 or np.abs(np.array(single.pose) - np.array(pose)).max() > 5e-3
    with NdtPyramid2D() as p:
        p.set_target(xt, yt)
        r = p.align(xs, ys, (0, 0, 0))
    assert r.status == 0 and np.abs(np.array(r.pose) - np.array(pose)).max() < 3e-3
    # the oracle running the same schedule
    cur = (0.0, 0.0, 0.0)
    for mult, er in PYRAMID_LEVELS:
        prm = o.NdtParams(cell_size=0.5 * mult, eig_ratio=er, eps_trans=1e-3, eps_rot=1e-4, max_iterations=30,
                          step_max_trans=0.5 * mult)
        cur = o.align(o.build_grid(xt, yt, prm), xs, ys, cur, prm)["pose"]
    prm = o.NdtParams()
    ref = o.align(o.build_grid(xt, yt, prm), xs, ys, cur, prm)
    assert ref["status"] == 0
    assert np.abs(np.array(r.pose) - np.array(ref["pose"])).max() < 1e-4


@pytest.mark.gpu
def test_async_converged_mode(gpu_lib):
    """ndt2d_align_dev_async in converged mode: begun here, finished (kept fed) by align_finish;
    any other call on the handle finishes a loop in flight first."""
    import torch
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(2, n_tgt=30000, n_src=30000)
    sx, sy = torch.from_numpy(d["sx"]).cuda(), torch.from_numpy(d["sy"]).cuda()
    d2 = synth.make_pair(4, pair_index=3, n_tgt=20000, n_src=20000)
    with NdtMatcher2D() as m:
        m.set_target(d["tx"], d["ty"])
        ref = m.align(sx, sy, d["init"])
        assert ref.status == 0 and ref.iterations > 16                 # longer than the two chunks begun by async
        m.align_async(sx, sy, d["init"])
        got = m.finish()
        assert got.pose == ref.pose and got.iterations == ref.iterations and np.array_equal(got.H, ref.H)
        # two asynchronous calls back to back: the second finishes the first, then runs
        m.align_async(sx, sy, d["init"])
        m.align_async(sx, sy, (0.05, 0.02, 0.0))
        second = m.finish()
        assert second.status == 0 and np.abs(np.array(second.pose) - np.array(ref.pose)).max() < 5e-4
        # a target change with a loop in flight: the loop is finished against the old target first
        m.align_async(sx, sy, d["init"])
        m.set_target(d2["tx"], d2["ty"])
        late = m.finish()
        assert late.pose == ref.pose and late.iterations == ref.iterations
        # evaluation with a loop in flight
        m.set_target(d["tx"], d["ty"])
        m.align_async(sx, sy, d["init"])
        H, g, s, nh = m.evaluate(d["sx"], d["sy"], ref.pose)
        assert nh > 0 and s > 0


@pytest.mark.gpu
@pytest.mark.parametrize("n_src,kw", [(360, {}), (1000, {}), (2048, {}), (2049, {}), (4096, {}), (1000, dict(hessian_mode=1)),
                                      (1500, dict(overlap_grids=4)), (777, dict(line_search=3)), (900, dict(step_scale=2.5)),
                                      (1000, dict(fixed_iterations=7))])
def test_short_scan_kernel_equals_launch_per_iteration_path(gpu_lib, n_src, kw):
    """Scans of up to 4096 points run the whole loop in one workgroup (k_align_small); the result
    must be what the general path (one launch per iteration, tuning short_scan_kernel = 0) returns, up to
    the float32 summation order, and what the oracle returns."""
    import torch
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    from oracle import ndt2d as o
    d = synth.make_pair(2, n_tgt=50000, n_src=n_src)
    sx, sy = torch.from_numpy(d["sx"]).cuda(), torch.from_numpy(d["sy"]).cuda()
    res = {}
    for no_small in ("0", "1"):
        with NdtMatcher2D(tuning={"short_scan_kernel": 0 if no_small == "1" else 1}, **kw) as m:
            m.set_target(d["tx"], d["ty"])
            res[no_small] = (m.align(sx, sy, d["init"]), m.align(d["sx"], d["sy"], d["init"]),
                             m.evaluate(d["sx"], d["sy"], d["pose"]))
            m.align_async(sx, sy, d["init"])
            res[no_small] += (m.finish(),)
    small, general = res["0"], res["1"]
    assert small[0].pose == small[1].pose == small[3].pose            # device / host / async entry points
    a, b = small[0], general[0]
    if kw.get("hessian_mode", 0) == 0:        # Newton trajectories on sparse scans amplify rounding
        assert a.status == b.status and abs(a.iterations - b.iterations) <= 1
        assert np.abs(np.array(a.pose) - np.array(b.pose)).max() < 2e-6
        assert a.n_hit == b.n_hit
    Hs, gs, ss, ns_ = small[2]
    Hg, gg, sg, ng = general[2]
    assert ns_ == ng and abs(ss - sg) <= 2e-6 * sg and np.abs(Hs - Hg).max() <= 2e-6 * np.abs(Hg).max()
    okw = dict(kw)
    if "overlap_grids" in okw:
        okw["overlap"] = okw.pop("overlap_grids")
    prm = o.NdtParams(**okw)
    grid = o.build_grids(d["tx"], d["ty"], prm) if prm.overlap == 4 else o.build_grid(d["tx"], d["ty"], prm)
    if kw.get("hessian_mode", 0) == 0:
        # the float32-mirror oracle follows the device evaluation point for point: tight; the float64
        # oracle may settle a few cell flips away on these mid-density scans (its own float32 mirror
        # does too): loose
        r32 = o.align(grid, d["sx"], d["sy"], d["init"], prm, mirror32=True)
        r64 = o.align(grid, d["sx"], d["sy"], d["init"], prm)
        assert a.status == r32["status"] == r64["status"]
        assert np.abs(np.array(a.pose) - np.array(r32["pose"])).max() < 1e-6
        assert np.abs(np.array(a.pose) - np.array(r64["pose"])).max() < 5e-4


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"launch_graphs": 0}, {"chunk_launches": 2}, {"chunk_launches": 64}])
def test_launch_chain_drivers_agree(gpu_lib, env):
    """The converged-mode loop gives the same result however the host feeds it: plain stream
    launches with a poll per chunk, or graph replays of 2, 8 (default) or 64 launches."""
    import torch
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(2, n_tgt=30000, n_src=30000)
    sx, sy = torch.from_numpy(d["sx"]).cuda(), torch.from_numpy(d["sy"]).cuda()
    with NdtMatcher2D() as m:
        m.set_target(d["tx"], d["ty"])
        ref = m.align(sx, sy, d["init"])
    with NdtMatcher2D(tuning=dict(env)) as m:
        m.set_target(d["tx"], d["ty"])
        for src in ((sx, sy), (d["sx"], d["sy"])):
            for _ in range(3):
                r = m.align(*src, d["init"])
                assert r.status == ref.status and r.iterations == ref.iterations
                assert r.pose == ref.pose and np.array_equal(r.H, ref.H)
        m.align_async(sx, sy, d["init"])
        r = m.finish()
        assert r.pose == ref.pose and r.iterations == ref.iterations


@pytest.mark.gpu
def test_wide_workgroups_for_large_scans_equal_the_narrow_ones(gpu_lib):
    """Scans of >= 300k points run k_iterate with 1024-thread workgroups (loads in flight); the
    result is the 256-thread one up to the float32 summation order."""
    import torch
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(3, n_src=400_000)
    tx, ty, sx, sy = (torch.from_numpy(d[k]).cuda() for k in ("tx", "ty", "sx", "sy"))
    res = {}
    for off in ("0", "1"):
        with NdtMatcher2D(tuning={"wide_threshold": 0} if off == "1" else {}) as m:
            m.set_target(tx, ty)
            res[off] = (m.align(sx, sy, d["init"]), m.align(sx, sy, d["init"]))
    for off in ("0", "1"):
        assert res[off][0].pose == res[off][1].pose                     # deterministic
    a, b = res["0"][0], res["1"][0]
    assert a.status == b.status == 0 and abs(a.iterations - b.iterations) <= 1 and a.n_hit == b.n_hit
    assert np.abs(np.array(a.pose) - np.array(b.pose)).max() < 2e-6
    assert np.abs(np.array(a.pose) - np.array(d["pose"])).max() < 2e-3
