"""Seeded differential sweep: ragged, clustered, duplicated, NaN-laced clouds at random offsets and
cell sizes through the device build and evaluation against the oracle, and the batch kernel
against the single-pair kernel on the same inputs (two independent device implementations)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cloud(rng, n, centre, spread):
    k = int(rng.integers(1, 6))
    blobs = rng.normal(size=(k, 2)) * spread + centre
    which = rng.integers(0, k, n)
    aniso = rng.uniform(0.02, 1.0, size=(k, 2))
    p = blobs[which] + rng.normal(size=(n, 2)) * aniso[which]
    if n > 8 and rng.random() < 0.5:                       # exact duplicates
        p[rng.integers(0, n, n // 8)] = p[rng.integers(0, n, n // 8)]
    if n > 4 and rng.random() < 0.5:                       # a straight wall: rank-1 cells -> eigenvalue clamp
        m = n // 4
        p[:m, 0] = centre[0] + np.linspace(-spread, spread, m)
        p[:m, 1] = centre[1]
    x, y = p[:, 0].astype(np.float32), p[:, 1].astype(np.float32)
    if rng.random() < 0.4:                                 # no-return beams
        bad = rng.integers(0, n, max(1, n // 20))
        x[bad] = np.nan
        y[bad[::2]] = np.inf
    return x, y


def _cases(n_cases, seed):
    rng = np.random.default_rng(seed)
    for i in range(n_cases):
        centre = rng.uniform(-800, 800, 2) if rng.random() < 0.5 else rng.uniform(-5, 5, 2)
        spread = float(rng.uniform(0.5, 20.0))
        cell = float(rng.choice([0.1, 0.25, 0.3, 0.5, 0.75, 1.0, 2.0, 3.0]))
        nt = int(rng.choice([3, 17, 64, 257, 1000, 4099, 20000]))
        ns = int(rng.choice([1, 5, 63, 64, 65, 1000, 8191]))
        tx, ty = _cloud(rng, nt, centre, spread)
        sx, sy = _cloud(rng, ns, centre, spread)
        pose = (float(rng.normal(0, 0.3)), float(rng.normal(0, 0.3)), float(rng.normal(0, 0.05)))
        kw = dict(cell_size=cell, min_points=int(rng.integers(2, 7)), eig_ratio=float(rng.choice([1e-3, 1e-2, 0.1])),
                  hessian_mode=int(rng.integers(0, 2)))
        yield i, tx, ty, sx, sy, pose, kw


def test_random_clouds_build_and_evaluate_like_the_oracle(gpu_lib):
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
    from oracle import ndt2d as o
    checked = 0
    for i, tx, ty, sx, sy, pose, kw in _cases(60, seed=20261004):
        finite = np.isfinite(tx) & np.isfinite(ty)
        if finite.sum() == 0:
            continue
        prm = o.NdtParams(**kw)
        g = o.build_grid(tx[finite], ty[finite], prm)
        with NdtMatcher2D(**kw) as m:
            info = m.set_target(tx, ty)                                  # non-finite points are ignored
            assert (info.width, info.height, info.ox, info.oy) == (g.W, g.H, g.ox, g.oy), (i, kw)
            count, mean, icov = m.grid()
            np.testing.assert_array_equal(count.astype(np.int64), g.count, err_msg=f"case {i}")
            assert info.n_valid == g.n_valid, (i, kw)
            np.testing.assert_array_equal(icov[:, 0] != 0, g.valid, err_msg=f"case {i}")
            if g.n_valid == 0:
                r = m.align(sx, sy, pose)
                assert r.status == L.NDT_TOO_FEW_CELLS
                continue
            H, gr, score, n_hit = m.evaluate(sx, sy, pose)
            ok = np.isfinite(sx) & np.isfinite(sy)
            Hm, gm, sm, nm = o.evaluate(g, sx[ok], sy[ok], pose, prm, mirror32=True)
            assert abs(n_hit - nm) <= 2, (i, n_hit, nm)
            if nm > 2 and n_hit == nm:
                hs = max(np.abs(Hm).max(), 1e-30)
                assert np.abs(H - Hm).max() / hs < 1e-3, (i, kw)
                assert abs(score - sm) <= 1e-3 * max(sm, 1e-6), (i, score, sm)
            # the batch kernel (grid in LDS) evaluates the same pair at the same pose: one update,
            # H/g/score are those of the evaluation at the start pose
            with NdtBatch2D(fixed_iterations=1, **kw) as b:
                rb = b.align([(tx, ty)], [(sx, sy)], [pose])[0]
            if rb.status in (L.NDT_OK, L.NDT_NOT_CONVERGED):
                assert rb.n_hit == n_hit, (i, rb.n_hit, n_hit)
                hs = max(np.abs(H).max(), 1e-30)
                assert np.abs(rb.H - H).max() / hs < 1e-4, (i, kw)
                assert abs(rb.score - score) <= 1e-4 * max(score, 1e-6)
            else:
                assert rb.status in (L.NDT_TOO_FEW_HITS, L.NDT_DEGENERATE_HESSIAN), (i, rb.status)
            checked += 1
    assert checked >= 40


def test_random_clouds_3d(gpu_lib):
    """The 3D build (binned tiles, Jacobi finalise) and evaluation on random clustered clouds."""
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtBatch3D, NdtMatcher3D
    from oracle import ndt3d as o3
    rng = np.random.default_rng(77)
    checked = n_batch = 0
    for i in range(24):
        centre = rng.uniform(-300, 300, 3) if i % 2 else rng.uniform(-3, 3, 3)
        spread = float(rng.uniform(1.0, 15.0))
        cell = float(rng.choice([0.5, 1.0, 1.5, 2.0]))
        nt = int(rng.choice([5, 100, 1000, 5000, 30000]))
        ns = int(rng.choice([1, 64, 65, 1000, 4097]))
        k = int(rng.integers(1, 5))
        blobs = rng.normal(size=(k, 3)) * spread + centre
        sc = rng.uniform(0.05, 1.5, size=(k, 3))

        def cloud(n):
            w = rng.integers(0, k, n)
            p = (blobs[w] + rng.normal(size=(n, 3)) * sc[w]).astype(np.float32)
            if n > 8:                                                      # a flat floor patch: rank-2 voxels
                p[: n // 4, 2] = np.float32(centre[2])
            return p[:, 0].copy(), p[:, 1].copy(), p[:, 2].copy()

        tx, ty, tz = cloud(nt)
        sx, sy, sz = cloud(ns)
        kw = dict(cell_size=cell, min_points=int(rng.integers(3, 8)), eig_ratio=float(rng.choice([1e-3, 1e-2])))
        prm = o3.Ndt3Params(**kw)
        g = o3.build_grid3(tx, ty, tz, prm)
        pose = tuple(rng.normal(0, 0.2, 3)) + tuple(rng.normal(0, 0.03, 3))
        with NdtMatcher3D(**kw) as m:
            info = m.set_target(tx, ty, tz)
            assert (info.width, info.height, info.depth) == g.dims, (i, kw)
            count, mean, icov = m.grid()
            np.testing.assert_array_equal(count.astype(np.int64), g.count, err_msg=f"case {i}")
            assert info.n_valid == g.n_valid, (i, kw)
            if g.n_valid == 0:
                assert m.align(sx, sy, sz, pose).status == L.NDT_TOO_FEW_CELLS
                continue
            H, gr, s, nh = m.evaluate(sx, sy, sz, pose)
            Hm, gm, sm, nm = o3.evaluate3(g, sx, sy, sz, pose, prm, mirror32=True)
            assert abs(nh - nm) <= 2, (i, nh, nm)
            if nm > 3 and nh == nm:
                hs = max(np.abs(Hm).max(), 1e-30)
                assert np.abs(H - Hm).max() / hs < 2e-3, (i, kw)
                assert abs(s - sm) <= 2e-3 * max(sm, 1e-6)
            # the 3D batch kernel (voxel grid in LDS, or in global memory when the grid outgrows the carve) evaluates the same
            # pair at the same pose - map-frame sums against the single-pair kernel's Jacobian form - with NaN-laced copies
            # of the clouds: no-return points must change nothing
            def laced(c, every):
                out = []
                for a in c:
                    b = np.concatenate([a, np.full(max(1, a.size // every), np.nan, np.float32)])
                    out.append(b)
                out[1][-1] = np.inf
                return tuple(out)
            with NdtBatch3D(fixed_iterations=1, **kw) as b:
                rb, rl = b.align([(tx, ty, tz), laced((tx, ty, tz), 9)], [(sx, sy, sz), laced((sx, sy, sz), 7)], [pose, pose])
            assert rb.status == rl.status and rb.n_hit == rl.n_hit and rb.pose == rl.pose and np.array_equal(rb.H, rl.H), i
            if rb.status in (L.NDT_OK, L.NDT_NOT_CONVERGED):
                assert rb.n_hit == nh, (i, rb.n_hit, nh)
                hs = max(np.abs(H).max(), 1e-30)
                assert np.abs(rb.H - H).max() / hs < 2e-4, (i, kw)
                assert abs(rb.score - s) <= 1e-4 * max(s, 1e-6)
                n_batch += 1
            else:
                assert rb.status in (L.NDT_TOO_FEW_HITS, L.NDT_DEGENERATE_HESSIAN), (i, rb.status)
            checked += 1
    assert checked >= 12 and n_batch >= 8


def test_random_clouds_three_iterations_batch_vs_single_pair(gpu_lib):
    """Three fixed iterations with random options (Hessian form, over-relaxation, line search, step
    limits) on random clouds: the four loop drivers - k_align_small / k_iterate behind the single-pair
    handle, the 256- and 1024-thread variants of k_batch - must land on the same pose."""
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
    rng = np.random.default_rng(424242)
    compared = 0
    for i, tx, ty, sx, sy, pose, kw in _cases(120, seed=777):
        # a source that overlaps the target: a noisy subset of it, a pose near the identity
        fin = np.flatnonzero(np.isfinite(tx) & np.isfinite(ty))
        if fin.size < 20:
            continue
        pick = rng.choice(fin, size=min(len(sx), fin.size), replace=False)
        sx = (tx[pick] + rng.normal(0, 0.02, pick.size)).astype(np.float32)
        sy = (ty[pick] + rng.normal(0, 0.02, pick.size)).astype(np.float32)
        pose = (float(rng.normal(0, 0.05)), float(rng.normal(0, 0.05)), float(rng.normal(0, 0.005)))
        kw = dict(kw, fixed_iterations=3, step_scale=float(rng.choice([1.0, 2.0, 3.0])),
                  line_search=int(rng.choice([0, 0, 2])), step_max_trans=float(rng.choice([0.5, 0.05])))
        with NdtMatcher2D(**kw) as m:
            info = m.set_target(tx, ty)
            if info.n_valid == 0:
                continue
            a = m.align(sx, sy, pose)
        with NdtBatch2D(**kw) as b:
            c = b.align([(tx, ty)], [(sx, sy)], [pose])[0]
        if c.status == L.NDT_ERR_CAPACITY:
            continue
        assert a.status == c.status, (i, kw, a.status, c.status)
        if a.status in (L.NDT_OK, L.NDT_NOT_CONVERGED) and a.n_hit > 10:
            assert a.iterations == c.iterations == 3
            scale = max(1.0, np.abs(np.array(a.pose[:2])).max())
            assert np.abs(np.array(a.pose) - np.array(c.pose)).max() < 2e-4 * scale, (i, kw, a.pose, c.pose)
            compared += 1
    assert compared >= 40


def test_random_clouds_3d_on_one_reused_handle(gpu_lib):
    """The same kind of clouds through ONE handle: from the second build on the geometry is decided on the device
    (ndt3d_api.hpp set_target3_single_sync), with grids that shrink, grow past the storage (fallback inside the call),
    move far away and change cell population; dense blobs put thousands of points into single tiles (shared tiles: slabs,
    lane-private LDS copies).  Geometry, every voxel's count and the number of valid voxels must be the oracle's."""
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    from oracle import ndt3d as o3
    rng = np.random.default_rng(4242)
    kw = dict(cell_size=1.0, min_points=5, eig_ratio=1e-2)
    prm = o3.Ndt3Params(**kw)
    with NdtMatcher3D(**kw) as m:
        for i in range(20):
            centre = rng.uniform(-200, 200, 3) if i % 3 == 0 else rng.uniform(-3, 3, 3)
            spread = float(rng.choice([0.5, 2.0, 6.0, 12.0]))
            n = int(rng.choice([7, 300, 3000, 30000, 120000]))
            k = int(rng.integers(1, 4))
            blobs = rng.normal(size=(k, 3)) * spread + centre
            sc = rng.uniform(0.05, 1.0, size=(k, 3))
            w = rng.integers(0, k, n)
            p = (blobs[w] + rng.normal(size=(n, 3)) * sc[w]).astype(np.float32)
            if n > 50:
                p[::17, 0] = np.nan                                        # no-return points
            fin = np.isfinite(p).all(axis=1)
            g = o3.build_grid3(p[fin, 0].copy(), p[fin, 1].copy(), p[fin, 2].copy(), prm)
            info = m.set_target(p[:, 0].copy(), p[:, 1].copy(), p[:, 2].copy())
            assert (info.width, info.height, info.depth) == g.dims, i
            assert (info.ox, info.oy, info.oz) == tuple(np.float32(v) for v in g.o), i
            count, _, _ = m.grid()
            np.testing.assert_array_equal(count.astype(np.int64), g.count, err_msg=f"case {i}")
            assert info.n_valid == g.n_valid, i


def test_random_pairs_through_the_overlapping_grid_batch(gpu_lib):
    """The batch with four overlapping grids against the single-pair path with the same option on random clouds: one
    evaluation at the start pose (the four grids themselves: hits, score, Hessian)."""
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
    checked = n_ok = 0
    for i, tx, ty, sx, sy, pose, kw in _cases(40, seed=991):
        if (np.isfinite(tx) & np.isfinite(ty)).sum() == 0:
            continue
        kw = dict(kw, overlap_grids=4)
        with NdtMatcher2D(**kw) as m:
            info = m.set_target(tx, ty)
            if info.n_valid == 0 or 4 * info.width * info.height > (1 << 18):
                continue
            H, gr, score, n_hit = m.evaluate(sx, sy, pose)
        with NdtBatch2D(fixed_iterations=1, **kw) as b:
            rb = b.align([(tx, ty)], [(sx, sy)], [pose])[0]
        if rb.status in (L.NDT_OK, L.NDT_NOT_CONVERGED):
            assert rb.n_hit == n_hit, (i, rb.n_hit, n_hit)
            hs = max(np.abs(H).max(), 1e-30)
            assert np.abs(rb.H - H).max() / hs < 1e-4, (i, kw)
            assert abs(rb.score - score) <= 1e-4 * max(score, 1e-6)
            n_ok += 1
        else:       # (random sources against random targets mostly miss the map: the status must say so)
            assert rb.status in (L.NDT_TOO_FEW_HITS, L.NDT_DEGENERATE_HESSIAN), (i, rb.status)
            assert n_hit < 3 or rb.status == L.NDT_DEGENERATE_HESSIAN, (i, n_hit, rb.status)
        checked += 1
    assert checked >= 15 and n_ok >= 3
