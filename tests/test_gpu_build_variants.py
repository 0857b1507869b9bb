"""The chunk-sorted grid build (round 3: k_bounds_parts -> k_chunk_sort -> k_tile_gather, ndt2d_build_sorted.hpp) against the
scattered-atomics build (k_accumulate + k_finalise) on the inputs its moving parts care about: every chunk size (256 x 4 / 8 /
16 points), clouds smaller than a wave, array slices that are not 16-byte aligned (the scalar bounds loop), NaN and infinite
points, a cloud that sits in one cell / one tile (a tile shared by eight workgroups, lists handed to the last arriver), an
extent with more tiles than the run table may hold (the round-1 binned build takes over) and more than the tile histogram
holds (the atomic build), and submap updates onto shared and partial tiles.  Exact integer sums: every grid must be bit for
bit the atomic build's, with the same count of valid cells and of points outside the grid."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)


def _build(Matcher, variant, x, y, adds=(), **kw):
    import torch
    with Matcher(tuning={"binned_build": variant}, **kw) as m:
        info = m.set_target(x, y)
        outs = []
        for ax, ay, pose in adds:
            if pose is None:
                outs.append(m.add_target_points(ax, ay))
            else:
                outs.append(m.add_target_points(torch.as_tensor(ax).cuda(), torch.as_tensor(ay).cuda(), pose=pose))
        gi = m.grid_info()
        return m.grid(), (info.width, info.height, info.n_valid), (gi.n_valid, gi.n_points), outs


def _check(Matcher, x, y, adds=(), **kw):
    ref = _build(Matcher, 0, x, y, adds, **kw)
    for variant in (1, 2):
        got = _build(Matcher, variant, x, y, adds, **kw)
        _same(ref[0], got[0])
        assert ref[1:] == got[1:], (variant, ref[1:], got[1:])
    return ref


@pytest.fixture(scope="module")
def Matcher(gpu_lib):
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    return NdtMatcher2D


@pytest.mark.parametrize("n", [1, 63, 257, 4097, 131_073, 600_000])
def test_every_chunk_size_and_ragged_tails(Matcher, n):
    rng = np.random.default_rng(n)
    # walls of a 40 m room: thin structures, tens of points per cell at the larger sizes
    t = rng.uniform(0, 160, n)
    x = np.where(t < 40, t, np.where(t < 80, 40.0, np.where(t < 120, 120 - t, 0.0))) + rng.normal(0, 0.02, n)
    y = np.where(t < 40, 0.0, np.where(t < 80, t - 40, np.where(t < 120, 40.0, 160 - t))) + rng.normal(0, 0.02, n)
    _check(Matcher, x.astype(np.float32), y.astype(np.float32))


def test_unaligned_device_slices_and_non_finite_points(Matcher):
    import torch
    rng = np.random.default_rng(7)
    n = 50_000
    x = rng.uniform(-30, 30, n + 8).astype(np.float32)
    y = (0.3 * x + rng.normal(0, 0.05, n + 8)).astype(np.float32)
    x[[5, 100, 40_000]] = [np.nan, np.inf, -np.inf]
    y[[6, 200]] = [np.nan, np.inf]
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    ref = _build(Matcher, 0, x[1:n + 1], y[3:n + 3])
    for variant in (1, 2):
        with Matcher(tuning={"binned_build": variant}) as m:
            info = m.set_target(xd[1:n + 1], yd[3:n + 3])         # 4- and 12-byte offsets: neither array 16-byte aligned
            _same(ref[0], m.grid())
            assert (info.width, info.height, info.n_valid) == ref[1]


def test_one_cell_and_one_tile(Matcher):
    """300 000 points inside one 0.5 m cell (its count stays below the 2^20 the sums are exact for), then inside one tile:
    the tile is shared by eight workgroups and the lists carry one cell / a few hundred cells."""
    rng = np.random.default_rng(3)
    n = 300_000
    x = rng.uniform(10.05, 10.45, n).astype(np.float32)
    y = rng.uniform(-3.95, -3.55, n).astype(np.float32)
    x[:2] = [0.0, 20.0]; y[:2] = [-10.0, 5.0]                       # the extent: a 40 x 30 m grid around the blob
    ref = _check(Matcher, x, y)
    assert ref[0][0].max() >= n - 2
    x2 = rng.uniform(4.0, 11.9, n).astype(np.float32)               # one 16 m tile's worth of cells
    y2 = (-6.0 + 0.5 * np.sin(x2) + rng.normal(0, 0.03, n)).astype(np.float32)
    x2[:2] = [0.0, 20.0]; y2[:2] = [-10.0, 5.0]
    _check(Matcher, x2, y2)


def test_more_tiles_than_the_run_table_or_the_histogram_hold(Matcher):
    rng = np.random.default_rng(11)
    n = 150_000
    # 1.2 km x 1.2 km at 0.5 m: 75 x 75 = 5625 tiles x 147 chunks > 2^20 table entries -> the round-1 binned build
    x = rng.uniform(0, 1200, n).astype(np.float32)
    y = (600 + 500 * np.sin(x / 90.0) + rng.normal(0, 0.05, n)).astype(np.float32)
    _check(Matcher, x, y)
    # 1.6 km x 1.6 km: 100 x 100 tiles > 8192 -> scattered atomics whatever the knob says
    x = rng.uniform(0, 1600, n).astype(np.float32)
    y = (800 + 700 * np.sin(x / 120.0) + rng.normal(0, 0.05, n)).astype(np.float32)
    _check(Matcher, x, y)


def test_submap_updates_on_shared_partial_and_untouched_tiles(Matcher):
    from gtsam_ndt_amd import synth
    d = synth.make_pair(2)                                          # 100k / 100k in a 50 m room: 4 x 4 tiles, the last row partial
    n = len(d["tx"])
    keep = np.zeros(n, bool); keep[::3] = True
    for a in (d["tx"], d["ty"]):
        keep[[np.argmin(a), np.argmax(a)]] = True
    corner = (d["tx"] < d["tx"].min() + 6) & (d["ty"] < d["ty"].min() + 6)       # an update that touches one tile only
    adds = [(d["tx"][~keep], d["ty"][~keep], None),                 # 66k points: every tile shared, cached sums added by the last arriver
            (d["tx"][corner], d["ty"][corner], None),               # one tile touched, fifteen leave at once
            (d["sx"], d["sy"], d["pose"]),                          # a scan moved into the map frame on the way in
            (d["tx"][:50] + 500.0, d["ty"][:50], None)]             # all outside: counted, nothing changes
    ref = _check(Matcher, d["tx"][keep], d["ty"][keep], adds)
    assert ref[3][0] == 0 and ref[3][3] == 50


def test_overlapping_grids_through_the_sorted_build(Matcher):
    from gtsam_ndt_amd import synth
    d = synth.make_pair(2)
    half = len(d["tx"]) // 2
    ext = np.unique([np.argmin(d["tx"]), np.argmax(d["tx"]), np.argmin(d["ty"]), np.argmax(d["ty"])])
    first = np.union1d(np.arange(half), ext)
    rest = np.setdiff1d(np.arange(len(d["tx"])), first)
    out = {}
    for variant in (0, 1, 2):
        with Matcher(overlap_grids=4, tuning={"binned_build": variant}) as m:
            nv = m.set_target(d["tx"][first], d["ty"][first]).n_valid
            m.add_target_points(d["tx"][rest], d["ty"][rest])
            r = m.align(d["sx"], d["sy"], d["init"])
            out[variant] = (nv, m.grid_info().n_valid, r.pose, r.iterations)
    assert out[0] == out[1] == out[2]


@pytest.mark.parametrize("n", [5_000, 131_073, 600_000])
def test_clouds_in_spatial_order(Matcher, n):
    """A scan in bearing order (and a submap update in the same order): whole chunks fall into one tile - few runs of
    thousands of points, the blocks-of-256 tail of the gather kernel's wave-per-run loop."""
    rng = np.random.default_rng(n + 5)
    t = np.sort(rng.uniform(0, 160, n))                    # along the walls of a 40 m room, in order
    x = np.where(t < 40, t, np.where(t < 80, 40.0, np.where(t < 120, 120 - t, 0.0))) + rng.normal(0, 0.02, n)
    y = np.where(t < 40, 0.0, np.where(t < 80, t - 40, np.where(t < 120, 40.0, 160 - t))) + rng.normal(0, 0.02, n)
    x, y = x.astype(np.float32), y.astype(np.float32)
    m = n // 3
    _check(Matcher, x, y, adds=[(x[:m] + np.float32(0.13), y[:m], None), (x[m:2 * m], y[m:2 * m], (0.2, -0.1, 0.01))])
