"""The grid's outermost ring of cells stays empty, whatever the caller adds.

The source-side lookup clamps cell keys onto the grid instead of testing them, so every out-of-range,
NaN or dead-lane lookup reads a ring cell - which therefore must never become valid.  Two ways a
point could get there: ndt2d_reserve_target + ndt2d_add_target_points with points beyond the
reserved box, and a boundary point that float32 rounding of (x - ox) * inv_c puts into cell 0
with a cell size that is not a power of two.  Both count as outside the grid, on the device and
in the oracle (oracle/ndt2d.py cell_keys32(interior=True))."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth

pytestmark = pytest.mark.gpu

BOX = (-20.0, -20.0, 20.0, 20.0)       # inside the 50 m room of config 2: points lie beyond it on all sides


def _ring(a, W, H):
    a = a.reshape(H, W)
    return np.concatenate([a[0, :], a[-1, :], a[:, 0], a[:, -1]])


@pytest.mark.parametrize("mode", [0, 1])
def test_reserved_grid_with_points_beyond_the_box(gpu_lib, mode):
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    from oracle import ndt2d as o
    d = synth.make_pair(2, n_tgt=60000, n_src=20000)
    prm = o.NdtParams(hessian_mode=mode)
    g = o.build_grid(d["tx"], d["ty"], prm, bounds=BOX)
    with NdtMatcher2D(hessian_mode=mode) as m:
        info = m.reserve_target(*BOX)
        assert (info.width, info.height) == (g.W, g.H) and info.ox == g.ox and info.oy == g.oy
        # in two portions, the second through the merge path
        half = len(d["tx"]) // 2
        n_out = m.add_target_points(d["tx"][:half], d["ty"][:half]) + m.add_target_points(d["tx"][half:], d["ty"][half:])
        count, mean, icov = m.grid()
        np.testing.assert_array_equal(count.astype(np.int64), g.count)
        assert n_out == len(d["tx"]) - int(g.count.sum()) and n_out > 1000
        assert _ring(count, g.W, g.H).sum() == 0 and not _ring(icov[:, 0] != 0, g.W, g.H).any()
        # points within one cell of the box would have landed on the ring
        fx = np.floor((d["tx"] - g.ox) * g.inv_c); fy = np.floor((d["ty"] - g.oy) * g.inv_c)
        on_ring = ((fx == 0) | (fx == g.W - 1) | (fy == 0) | (fy == g.H - 1)) & (fx >= 0) & (fx < g.W) & (fy >= 0) & (fy < g.H)
        assert on_ring.sum() > 100
        # source: the whole room (a third of it outside the grid), NaN "no return" points, far-away points
        sx, sy = d["sx"].copy(), d["sy"].copy()
        sx[::7] = np.nan
        sy[3::11] = np.nan
        sx[5::13] = 1e6
        sy[6::17] = -np.inf
        for n in (len(sx), 2000):                           # k_iterate and the one-workgroup kernel
            for pose in (d["init"], d["pose"]):
                H, gr, score, n_hit = m.evaluate(sx[:n], sy[:n], pose)
                fin = np.isfinite(sx[:n]) & np.isfinite(sy[:n])
                Hm, gm, sm, nm = o.evaluate(g, sx[:n][fin], sy[:n][fin], pose, prm, mirror32=True)
                assert np.isfinite(H).all() and np.isfinite(gr).all() and np.isfinite(score)
                assert abs(n_hit - nm) <= 2
                assert abs(score - sm) / sm < 2e-5
                assert np.abs(H - Hm).max() / np.abs(Hm).max() < 5e-5
        r = m.align(sx, sy, d["init"])
        fin = np.isfinite(sx) & np.isfinite(sy)
        ref = o.align(g, sx[fin], sy[fin], d["init"], prm)
        assert r.status == 0 == ref["status"]
        assert np.abs(np.array(r.pose) - np.array(ref["pose"])).max() < 1e-4
        assert abs(r.n_hit - ref["n_hit"]) <= 3


def _shift_min_to(a, value):
    """float32 cloud moved so that its minimum is exactly `value`."""
    return (a - a.min() + np.float32(value)).astype(np.float32)


def test_boundary_point_rounded_into_cell_zero_is_outside(gpu_lib):
    """Cell size 0.3 with the cloud's minimum at exactly -60.0: float32((x - ox) * inv_c) = 0.9999974,
    cell 0 - the ring.  The point is left out (device and oracle), the ring stays empty."""
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
    from oracle import ndt2d as o
    d = synth.make_pair(1, n_tgt=20000, n_src=20000)
    tx, ty = _shift_min_to(d["tx"], -60.0), _shift_min_to(d["ty"], -58.8)
    dx, dy = float(tx[0] - d["tx"][0]), float(ty[0] - d["ty"][0])
    init = (d["init"][0] + dx, d["init"][1] + dy, d["init"][2])
    prm = o.NdtParams(cell_size=0.3)
    g = o.build_grid(tx, ty, prm)
    ox = np.float32((np.floor(-60.0 / 0.3) - 1.0) * 0.3)
    assert np.float32((np.float32(-60.0) - ox) * np.float32(1.0 / 0.3)) < 1.0          # the premise
    assert g.count.sum() < len(tx)
    with NdtMatcher2D(cell_size=0.3) as m:
        info = m.set_target(tx, ty)
        count, mean, icov = m.grid()
        np.testing.assert_array_equal(count.astype(np.int64), g.count)
        assert _ring(count, g.W, g.H).sum() == 0 and info.n_valid == g.n_valid
        r = m.align(d["sx"], d["sy"], init)
    ref = o.align(g, d["sx"], d["sy"], init, prm)
    assert r.status == 0 == ref["status"] and np.abs(np.array(r.pose) - np.array(ref["pose"])).max() < 1e-4
    assert abs(r.n_hit - ref["n_hit"]) <= 3
    with NdtBatch2D(cell_size=0.3) as b:                      # the LDS-built grid follows the same rule
        rb = b.align([(tx, ty)], [(d["sx"], d["sy"])], [init])[0]
    assert rb.status == 0 and np.abs(np.array(rb.pose) - np.array(r.pose)).max() < 2e-5 and abs(rb.n_hit - r.n_hit) <= 2
