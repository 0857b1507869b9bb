"""Parity across parameter settings (cell sizes that are not powers of two, validity and clamp
thresholds, score constants, step limits) on both the single-pair and the batch path."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth

pytestmark = pytest.mark.gpu

CASES = [
    dict(cell_size=0.3),
    dict(cell_size=1.0, eig_ratio=0.01),
    dict(cell_size=0.7, min_points=6),
    dict(cell_size=0.5, d1=2.0, d2=0.5),
    dict(cell_size=0.5, step_max_trans=0.02, step_max_rot=0.002, max_iterations=200),
    dict(cell_size=0.5, eps_trans=1e-4, eps_rot=1e-4),
]


@pytest.fixture(scope="module")
def pair():
    return synth.make_pair(2, n_tgt=40000, n_src=40000)


@pytest.mark.parametrize("kw", CASES, ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()))
def test_single_and_batch_match_oracle(gpu_lib, pair, kw):
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
    from oracle import ndt2d as o
    d = pair
    prm = o.NdtParams(**kw)
    g = o.build_grid(d["tx"], d["ty"], prm)
    ref = o.align(g, d["sx"], d["sy"], d["init"], prm)
    with NdtMatcher2D(**kw) as m:
        info = m.set_target(d["tx"], d["ty"])
        assert (info.width, info.height, info.n_valid) == (g.W, g.H, g.n_valid)
        count, mean, icov = m.grid()
        np.testing.assert_array_equal(count.astype(np.int64), g.count)
        v = g.valid
        nrm = np.linalg.norm(g.icov[v], axis=1, keepdims=True)
        # fixed-point coordinates (c*2^-22) bound the error by ~2*c*2^-23/sigma_min of the thinnest
        # cell: 1e-5 on the standard configs, up to a few 1e-5 for 3-point slivers at other cell sizes
        assert np.max(np.abs(icov[v] - g.icov[v]) / nrm) < 1e-4
        r = m.align(d["sx"], d["sy"], d["init"])
    with NdtBatch2D(**kw) as b:
        rb = b.align([(d["tx"], d["ty"])], [(d["sx"], d["sy"])], [d["init"]])[0]
    assert ref["status"] == r.status == rb.status
    for got in (r, rb):
        e = np.abs(np.array(got.pose) - np.array(ref["pose"]))
        assert e[0] < 1e-4 and e[1] < 1e-4 and e[2] < 1e-4, (kw, got.pose, ref["pose"])
        assert abs(got.iterations - ref["iterations"]) <= 4
        assert abs(got.score - ref["score"]) / ref["score"] < 2e-3


def test_far_from_origin_and_tiny_inputs(gpu_lib, pair):
    """Coordinates hundreds of metres from the origin (float32 resolution ~3e-5 m) and the
    smallest legal inputs keep working; an empty source is a usage error."""
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    from oracle import ndt2d as o
    d = pair
    off = np.float32(512.0)
    prm = o.NdtParams()
    tx, ty = d["tx"] + off, d["ty"] - off
    init = (d["init"][0] + 512.0, d["init"][1] - 512.0, 0.0)
    ref = o.align(o.build_grid(tx, ty, prm), d["sx"], d["sy"], init, prm)
    with NdtMatcher2D() as m:
        m.set_target(tx, ty)
        r = m.align(d["sx"], d["sy"], init)
        assert r.status == 0 == ref["status"]
        assert np.abs(np.array(r.pose) - np.array(ref["pose"])).max() < 1e-4
        with pytest.raises(L.NdtError):
            m.align(d["sx"][:0], d["sy"][:0], init)
        m.set_target(d["tx"][:3], d["ty"][:3])
        r = m.align(d["sx"][:5], d["sy"][:5], d["init"])
        assert r.status in (L.NDT_TOO_FEW_CELLS, L.NDT_TOO_FEW_HITS, L.NDT_OK, L.NDT_NOT_CONVERGED, L.NDT_DEGENERATE_HESSIAN)


@pytest.mark.parametrize("mode", [0, 1])
def test_overlapping_grids_match_oracle(gpu_lib, pair, mode):
    """Biber's four half-cell-shifted grids (params.overlap_grids = 4) on the single-pair path."""
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
    from oracle import ndt2d as o
    d = pair
    prm = o.NdtParams(overlap=4, hessian_mode=mode)
    grids = o.build_grids(d["tx"], d["ty"], prm)
    ref = o.align(grids, d["sx"], d["sy"], d["init"], prm)
    with NdtMatcher2D(overlap_grids=4, hessian_mode=mode) as m:
        info = m.set_target(d["tx"], d["ty"])
        assert (info.width, info.height) == (grids[0].W, grids[0].H)
        assert info.n_valid == sum(g.n_valid for g in grids)
        count, mean, icov = m.grid()                       # grid 0, the unshifted one
        np.testing.assert_array_equal(count.astype(np.int64), grids[0].count)
        for pose in (d["init"], d["pose"]):
            H, g, s, nh = m.evaluate(d["sx"], d["sy"], pose)
            Hm, gm, sm, nm = o.evaluate(grids, d["sx"], d["sy"], pose, prm, mirror32=True)
            assert abs(nh - nm) <= 6 and abs(s - sm) / sm < 5e-5
            assert np.abs(H - Hm).max() / np.abs(Hm).max() < 5e-5
        r = m.align(d["sx"], d["sy"], d["init"])
    assert r.status == 0 == ref["status"]
    e = np.abs(np.array(r.pose) - np.array(ref["pose"]))
    assert e[0] < 1e-4 and e[1] < 1e-4 and e[2] < 1e-4
    with NdtBatch2D(overlap_grids=4, hessian_mode=mode) as b:      # the batch path with the same option (more in test_gpu_batch_overlap.py)
        rb = b.align([(d["tx"], d["ty"])], [(d["sx"], d["sy"])], [d["init"]])[0]
    assert rb.status == 0 and np.abs(np.array(rb.pose) - np.array(r.pose)).max() < 2e-5


CASES3 = [
    dict(cell_size=0.7),
    dict(cell_size=2.0, eig_ratio=0.01),
    dict(cell_size=1.3, min_points=8),
    dict(cell_size=1.0, d1=2.0, d2=0.5),
    dict(cell_size=1.0, step_max_trans=0.05, step_max_rot=0.004, max_iterations=200),
    dict(cell_size=1.0, eps_trans=1e-4, eps_rot=1e-4, min_hits=50),
]


@pytest.fixture(scope="module")
def pair3():
    from gtsam_ndt_amd import synth3d
    return synth3d.make_pair3d(n_elev=32, n_azim=512, pose=tuple(0.5 * np.array(synth3d.T_STAR_3D)))


@pytest.mark.parametrize("kw", CASES3, ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()))
def test_3d_single_batch_and_multi_match_oracle(gpu_lib, pair3, kw):
    """The 3D paths across parameter settings (voxel sizes that are not powers of two, validity and clamp thresholds, score
    constants, step limits): voxel grid, single-pair alignment, the batch kernel and the multi-scan chain against the
    oracle."""
    import torch
    from gtsam_ndt_amd.matcher import NdtBatch3D, NdtMatcher3D
    from oracle import ndt3d as o3
    d = pair3
    prm = o3.Ndt3Params(**kw)
    g = o3.build_grid3(d["tx"], d["ty"], d["tz"], prm)
    ref = o3.align3(g, d["sx"], d["sy"], d["sz"], d["init"], prm)
    with NdtMatcher3D(**kw) as m:
        info = m.set_target(d["tx"], d["ty"], d["tz"])
        assert (info.width, info.height, info.depth, info.n_valid) == (*g.dims, g.n_valid)
        count, mean, icov = m.grid()
        np.testing.assert_array_equal(count.astype(np.int64), g.count)
        v = g.valid
        nrm = np.linalg.norm(g.icov[v], axis=1, keepdims=True)
        assert np.max(np.abs(icov[v] - g.icov[v]) / nrm) < 1e-4
        r = m.align(d["sx"], d["sy"], d["sz"], d["init"])
        s = tuple(torch.from_numpy(d[k]).cuda() for k in ("sx", "sy", "sz"))
        rm = m.align_multi_scan([s, s], [d["init"], d["init"]])[1]
    with NdtBatch3D(**kw) as b:
        rb = b.align([(d["tx"], d["ty"], d["tz"])], [(d["sx"], d["sy"], d["sz"])], [d["init"]])[0]
    assert ref["status"] == r.status == rb.status == rm.status
    assert rm.pose == r.pose and rm.iterations == r.iterations
    for got in (r, rb):
        e = np.abs(np.array(got.pose) - np.array(ref["pose"]))
        assert e.max() < 1e-4, (kw, got.pose, ref["pose"])
        assert abs(got.iterations - ref["iterations"]) <= 4
        assert abs(got.score - ref["score"]) / ref["score"] < 2e-3
