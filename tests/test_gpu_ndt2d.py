"""GPU parity tests proper: HIP path (through the C-ABI) vs the CPU oracle on the same seeded
inputs.  Parity is against this repo's oracle - the reference implementation is unavailable
(/root/reference/README.md:1 is the checkout's only line)."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4     # BASELINE.json north_star: converged pose within 1e-4 m / 1e-4 rad
POSE_TOL_RAD = 1e-4


@pytest.fixture(scope="module")
def oracle():
    from oracle import ndt2d
    return ndt2d


@pytest.fixture(scope="module")
def Matcher(gpu_lib):
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    return NdtMatcher2D


def _grid_parity(oracle, m, d, prm):
    g = oracle.build_grid(d["tx"], d["ty"], prm)
    info = m.set_target(d["tx"], d["ty"])
    assert (info.width, info.height) == (g.W, g.H)
    assert info.ox == g.ox and info.oy == g.oy and info.inv_cell == g.inv_c
    count, mean, icov = m.grid()
    # a1+a2: integer work is bit-exact
    np.testing.assert_array_equal(count.astype(np.int64), g.count)
    valid_dev = icov[:, 0] != 0
    np.testing.assert_array_equal(valid_dev, g.valid)
    assert info.n_valid == g.n_valid
    v = g.valid
    # a2/a3: mean to float32 resolution, Sigma^-1 to 1e-5 relative of its norm
    np.testing.assert_allclose(mean[v], g.mean[v], rtol=0, atol=2e-6 * max(1.0, np.abs(g.mean[v]).max()))
    nrm = np.linalg.norm(g.icov[v], axis=1, keepdims=True)
    assert np.max(np.abs(icov[v] - g.icov[v]) / nrm) < 1e-5
    return g


@pytest.mark.parametrize("config", [1, 2])
def test_grid_build_parity(oracle, Matcher, config):
    d = synth.make_pair(config)
    prm = oracle.NdtParams()
    with Matcher() as m:
        _grid_parity(oracle, m, d, prm)


@pytest.mark.parametrize("config,mode", [(1, 0), (1, 1), (2, 0), (2, 1)])
def test_evaluate_parity(oracle, Matcher, config, mode):
    """Rows a4-a7 at fixed poses: H, g, score vs the float32-mirror oracle (tight) and the
    float64 oracle (loose: a few boundary points may change cell)."""
    d = synth.make_pair(config)
    prm = oracle.NdtParams(hessian_mode=mode)
    with Matcher(hessian_mode=mode) as m:
        g = _grid_parity(oracle, m, d, prm)
        for pose in (d["init"], d["pose"], (d["pose"][0] + 0.02, d["pose"][1] - 0.01, d["pose"][2] + 0.002)):
            H, gr, score, n_hit = m.evaluate(d["sx"], d["sy"], pose)
            Hm, gm, sm, nm = oracle.evaluate(g, d["sx"], d["sy"], pose, prm, mirror32=True)
            Ht, gt, st, nt = oracle.evaluate(g, d["sx"], d["sy"], pose, prm, mirror32=False)
            assert abs(n_hit - nm) <= 2
            assert abs(n_hit - nt) <= max(3, int(1e-4 * len(d["sx"])))
            hs = np.abs(Hm).max()
            # gradient entries cancel: scale by the sum of magnitudes ~ sqrt(H_ii * score)
            gs = np.sqrt(np.abs(np.diag(Hm)) * max(sm, 1.0)) + 1e-30
            assert np.abs(H - Hm).max() / hs < 2e-5
            assert np.max(np.abs(gr - gm) / gs) < 2e-4
            assert abs(score - sm) / sm < 2e-5
            assert np.abs(H - Ht).max() / hs < 2e-3
            assert abs(score - st) / st < 2e-3


# Newton mode is aligned on the dense configs only: on the 1k-point scene its damped steps
# through indefinite Hessians are chaotic at rounding level (the oracle's own C and numpy
# forms part ways there), so only its per-evaluation parity is checked on config 1.
@pytest.mark.parametrize("config,mode", [(1, 0), (2, 0), (2, 1), (3, 0), (4, 0)])
def test_align_converged_pose_parity(oracle, Matcher, config, mode):
    """The headline parity number: converged SE(2) pose within 1e-4 m / 1e-4 rad of the CPU
    oracle on identical synthetic scans."""
    d = synth.make_pair(config)
    prm = oracle.NdtParams(hessian_mode=mode)
    ref = oracle.align(oracle.build_grid(d["tx"], d["ty"], prm), d["sx"], d["sy"], d["init"], prm)
    with Matcher(hessian_mode=mode) as m:
        m.set_target(d["tx"], d["ty"])
        r = m.align(d["sx"], d["sy"], d["init"])
    assert ref["status"] == 0 and r.status == 0
    e = np.abs(np.array(r.pose) - np.array(ref["pose"]))
    assert e[0] < POSE_TOL_M and e[1] < POSE_TOL_M and e[2] < POSE_TOL_RAD, (r.pose, ref["pose"])
    if mode == 0:   # Newton's tail on the discontinuous score stops by chance; GN's does not
        assert abs(r.iterations - ref["iterations"]) <= 3
    assert abs(r.n_hit - ref["n_hit"]) <= max(3, int(1e-4 * len(d["sx"])))
    assert abs(r.score - ref["score"]) / ref["score"] < 1e-3


def test_fixed_iterations_trace(oracle, Matcher):
    """fixed-K mode applies exactly K updates and follows the oracle's trajectory."""
    d = synth.make_pair(2)
    for K in (1, 5, 30):
        prm = oracle.NdtParams(fixed_iterations=K)
        ref = oracle.align(oracle.build_grid(d["tx"], d["ty"], prm), d["sx"], d["sy"], d["init"], prm)
        with Matcher(fixed_iterations=K) as m:
            m.set_target(d["tx"], d["ty"])
            r = m.align(d["sx"], d["sy"], d["init"])
        assert r.iterations == K == ref["iterations"]
        e = np.abs(np.array(r.pose) - np.array(ref["pose"]))
        assert e.max() < 1e-4


def test_determinism(Matcher):
    """Same input twice -> bitwise-identical grid, H, g, pose (fixed reduction trees, integer
    atomics only)."""
    d = synth.make_pair(2)
    out = []
    for _ in range(2):
        with Matcher() as m:
            m.set_target(d["tx"], d["ty"])
            grid = m.grid()
            r = m.align(d["sx"], d["sy"], d["init"])
            out.append((grid, r))
    (g0, r0), (g1, r1) = out
    for a, b in zip(g0, g1):
        np.testing.assert_array_equal(a, b)
    assert r0.pose == r1.pose and r0.iterations == r1.iterations
    np.testing.assert_array_equal(r0.H, r1.H)
    np.testing.assert_array_equal(r0.g, r1.g)


def test_permutation_invariance(Matcher):
    """Grid sums are exact integers: any target point order gives the same records bit for bit."""
    d = synth.make_pair(2)
    perm = np.random.default_rng(0).permutation(len(d["tx"]))
    with Matcher() as m:
        m.set_target(d["tx"], d["ty"])
        a = m.grid()
        m.set_target(d["tx"][perm], d["ty"][perm])
        b = m.grid()
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)


def test_identical_clouds_give_identity(Matcher):
    d = synth.make_pair(2)
    with Matcher() as m:
        m.set_target(d["tx"], d["ty"])
        r = m.align(d["tx"], d["ty"], (0.0, 0.0, 0.0))
    assert r.status == 0
    assert max(abs(v) for v in r.pose) < 1e-4


def test_incremental_target_equals_rebuild(Matcher):
    """adding a scan to the cached grid is exact - the result is
    bitwise the grid built from all points at once."""
    d = synth.make_pair(2)
    n = len(d["tx"])
    with Matcher() as m:
        m.set_target(d["tx"], d["ty"])
        full = m.grid()
        # first half must span the same extent: take every other point plus the extremes
        idx = np.arange(n)
        keep = (idx % 2 == 0)
        for a in (d["tx"], d["ty"]):
            keep[np.argmin(a)] = True
            keep[np.argmax(a)] = True
        m.set_target(d["tx"][keep], d["ty"][keep])
        outside = m.add_target_points(d["tx"][~keep], d["ty"][~keep])
        assert outside == 0
        inc = m.grid()
    for u, v in zip(full, inc):
        np.testing.assert_array_equal(u, v)


def test_edge_cases(Matcher, gpu_lib):
    from gtsam_ndt_amd import _lib as L
    d = synth.make_pair(1)
    with Matcher() as m:
        with pytest.raises(L.NdtError) as e:
            m.align(d["sx"], d["sy"])
        assert e.value.code == L.NDT_ERR_NO_TARGET
        # a target too sparse to have any valid cell
        m.set_target(np.array([0.0, 10.0], np.float32), np.array([0.0, 10.0], np.float32))
        r = m.align(d["sx"], d["sy"])
        assert r.status == L.NDT_TOO_FEW_CELLS
        # a source that misses every valid cell
        m.set_target(d["tx"], d["ty"])
        r = m.align(d["sx"] + 1000.0, d["sy"])
        assert r.status == L.NDT_TOO_FEW_HITS and r.iterations == 0
        # NaN / inf points are ignored, not propagated
        sx = d["sx"].copy(); sx[::7] = np.nan
        tx = d["tx"].copy(); tx[::11] = np.inf
        m.set_target(tx, d["ty"])
        r = m.align(sx, d["sy"], d["init"])
        assert np.isfinite(r.pose).all() and np.isfinite(r.H).all()
        # ragged sizes: 1 source point, non multiple of the block size
        m.set_target(d["tx"], d["ty"])
        r = m.align(d["sx"][:1], d["sy"][:1])
        assert r.status in (L.NDT_TOO_FEW_HITS, L.NDT_DEGENERATE_HESSIAN, L.NDT_OK, L.NDT_NOT_CONVERGED)
        r = m.align(d["sx"][:777], d["sy"][:777], d["init"])
        assert np.isfinite(r.pose).all()


def test_binned_build_equals_atomic_build(Matcher):
    """The three grid builds - chunk-sorted (default: ndt2d_build_sorted.hpp), the round-1 binned build
    (count / scan / scatter) and scattered global atomics - produce the same exact sums, hence bit-identical
    records and the same count of valid cells: for a one-shot build, an incremental update (also of a scan moved
    into the map frame on the way in), a second build on the same handle (the one-round-trip form) and overlapping
    grids, at the full 1M-point size."""
    d = synth.make_pair(3)
    half = len(d["tx"]) // 2
    ext = np.unique([np.argmin(d["tx"]), np.argmax(d["tx"]), np.argmin(d["ty"]), np.argmax(d["ty"])])
    first = np.union1d(np.arange(half), ext)
    rest = np.setdiff1d(np.arange(len(d["tx"])), first)
    out = {}
    for name, variant in (("sorted", 1), ("binned", 2), ("atomic", 0)):
        tune = {"binned_build": variant}
        with Matcher(tuning=tune) as m:
            info = m.set_target(d["tx"], d["ty"])
            full = m.grid() + (info.n_valid,)
            again = m.set_target(d["tx"], d["ty"])                       # the handle holds a grid now: one round trip
            assert again.n_valid == info.n_valid
            for u, v in zip(full[:3], m.grid()):
                np.testing.assert_array_equal(u, v)
            m.set_target(d["tx"][first], d["ty"][first])
            assert m.add_target_points(d["tx"][rest], d["ty"][rest]) == 0
            inc = m.grid() + (m.grid_info().n_valid,)
            assert m.add_target_points(d["tx"][:10] + 1e4, d["ty"][:10]) == 10      # outside: counted, ignored
            assert m.grid_info().n_valid == inc[3]
            # a scan merged with the pose an alignment returned (moved into the map frame inside the build)
            import torch
            m.add_target_points(torch.from_numpy(d["sx"]).cuda(), torch.from_numpy(d["sy"]).cuda(), pose=d["pose"])
            moved = m.grid() + (m.grid_info().n_valid,)
        with Matcher(overlap_grids=4, tuning=tune) as m:
            ov = m.set_target(d["tx"][:200000], d["ty"][:200000]).n_valid
            r = m.align(d["sx"], d["sy"], d["init"])
        out[name] = (full, inc, ov, r.pose, moved)
    fa, ia, oa, pa, ma = out["atomic"]
    assert ia[3] == fa[3]                                              # an update re-counts the valid cells correctly
    for name in ("sorted", "binned"):
        fb, ib, ob, pb, mb = out[name]
        for u, v in zip(fa[:3], fb[:3]):
            np.testing.assert_array_equal(u, v)
        assert fa[3] == fb[3] and oa == ob and pa == pb, name
        for u, v in zip(ia[:3], ib[:3]):
            np.testing.assert_array_equal(u, v)
        assert ia[3] == ib[3], name
        for u, v in zip(ma[:3], mb[:3]):
            np.testing.assert_array_equal(u, v)
        assert ma[3] == mb[3], name
    for u, v in zip(fa[:3], ia[:3]):
        np.testing.assert_array_equal(u, v)
