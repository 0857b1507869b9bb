"""ndt2d_align_multi_start_dev: up to 8 alignments of one scan from different initial poses in one
launch chain.  Contract: start k's result is bit for bit what ndt2d_align_dev returns for that
initial pose on the launch-per-iteration path (and hence within 1e-4 of the oracle, as that path
is); a start that has finished is frozen while the others go on."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth

pytestmark = pytest.mark.gpu

OFFSETS = [(0.0, 0.0, 0.0), (0.03, -0.02, 0.002), (-0.04, 0.03, -0.003), (0.05, 0.05, 0.004),
           (-0.02, -0.05, 0.0), (0.06, -0.01, -0.005), (-0.06, 0.02, 0.006), (0.01, 0.07, -0.002)]


def _starts(init, m):
    return [(init[0] + o[0], init[1] + o[1], init[2] + o[2]) for o in OFFSETS[:m]]


def _same(a, b):
    return (a.pose == b.pose and a.iterations == b.iterations and a.status == b.status and a.n_hit == b.n_hit
            and a.score == b.score and np.array_equal(a.H, b.H) and np.array_equal(a.g, b.g))


def _dev(d):
    import torch
    return torch.from_numpy(d["sx"]).cuda(), torch.from_numpy(d["sy"]).cuda()


@pytest.fixture(scope="module")
def pair2():
    return synth.make_pair(2)


@pytest.mark.parametrize("m", [1, 2, 3, 5, 8])
def test_every_start_equals_its_single_start_alignment_bitwise(gpu_lib, pair2, m):
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = pair2
    sx, sy = _dev(d)
    starts = _starts(d["init"], m)
    with NdtMatcher2D() as mm:
        mm.set_target(d["tx"], d["ty"])
        multi = mm.align_multi_start(sx, sy, starts)
        single = [mm.align(sx, sy, s) for s in starts]
    assert len(multi) == m
    iters = {r.iterations for r in multi}
    if m >= 5:
        assert len(iters) > 1                      # the starts do finish at different launches
    for a, b in zip(multi, single):
        assert b.status == 0 and _same(a, b), (a, b)


@pytest.mark.parametrize("kw", [dict(fixed_iterations=30), dict(hessian_mode=1), dict(line_search=4),
                                dict(step_scale=3.0), dict(fixed_iterations=7, hessian_mode=1),
                                dict(overlap_grids=4), dict(overlap_grids=4, hessian_mode=1, fixed_iterations=9)])
def test_options_keep_the_bitwise_contract(gpu_lib, pair2, kw):
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = pair2
    sx, sy = _dev(d)
    starts = _starts(d["init"], 6)
    with NdtMatcher2D(**kw) as mm:
        mm.set_target(d["tx"], d["ty"])
        multi = mm.align_multi_start(sx, sy, starts)
        single = [mm.align(sx, sy, s) for s in starts]
    for a, b in zip(multi, single):
        assert _same(a, b), (kw, a, b)


def test_starts_agree_with_the_oracle(gpu_lib, pair2):
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    from oracle import ndt2d as o
    d = pair2
    sx, sy = _dev(d)
    starts = _starts(d["init"], 4)
    prm = o.NdtParams()
    g = o.build_grid(d["tx"], d["ty"], prm)
    with NdtMatcher2D() as mm:
        mm.set_target(d["tx"], d["ty"])
        multi = mm.align_multi_start(sx, sy, starts)
    for s, r in zip(starts, multi):
        ref = o.align(g, d["sx"], d["sy"], s, prm)
        assert r.status == ref["status"] == 0
        assert np.abs(np.array(r.pose) - np.array(ref["pose"])).max() < 1e-4          # BASELINE.json tolerance
        assert abs(r.iterations - ref["iterations"]) <= 3


def test_sixty_four_starts_in_one_chain(gpu_lib, pair2):
    """The largest call: 64 starts on a 4 x 4 x 4 lattice around the guess (16 subsets of 4 per launch);
    a sample of them against their single-start alignments, bit for bit."""
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = pair2
    sx, sy = _dev(d)
    starts = [(d["init"][0] + 0.02 * (i - 1.5), d["init"][1] + 0.02 * (j - 1.5), 0.002 * (k - 1.5))
              for i in range(4) for j in range(4) for k in range(4)]
    with NdtMatcher2D(fixed_iterations=12) as mm:
        mm.set_target(d["tx"], d["ty"])
        multi = mm.align_multi_start(sx, sy, starts)
        assert len(multi) == 64 and all(r.status == 0 and r.iterations == 12 for r in multi)
        for k in (0, 7, 21, 42, 63):
            assert _same(multi[k], mm.align(sx, sy, starts[k]))


def test_a_start_that_misses_the_map_does_not_disturb_the_others(gpu_lib, pair2):
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = pair2
    sx, sy = _dev(d)
    starts = _starts(d["init"], 3)
    starts.insert(1, (1000.0, 0.0, 0.0))
    with NdtMatcher2D() as mm:
        mm.set_target(d["tx"], d["ty"])
        multi = mm.align_multi_start(sx, sy, starts)
        assert multi[1].status == L.NDT_TOO_FEW_HITS and multi[1].iterations == 0
        for k in (0, 2, 3):
            assert _same(multi[k], mm.align(sx, sy, starts[k]))


def test_wide_workgroups_and_short_scans(gpu_lib):
    """A 320k-point scan runs on 1024-thread workgroups (chains of four starts); a 3000-point scan is
    compared with the launch-per-iteration path (the single-start entry would pick the one-workgroup
    kernel, whose summation order differs)."""
    import torch
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(3, n_tgt=400_000, n_src=320_000)
    sx, sy = _dev(d)
    starts = _starts(d["init"], 6)
    with NdtMatcher2D(fixed_iterations=12) as mm:
        mm.set_target(d["tx"], d["ty"])
        multi = mm.align_multi_start(sx, sy, starts)
        for a, s in zip(multi, starts):
            assert _same(a, mm.align(sx, sy, s))
    d = synth.make_pair(1, n_tgt=3000, n_src=3000)
    sx, sy = _dev(d)
    starts = _starts(d["init"], 8)
    with NdtMatcher2D() as mm:
        mm.set_target(d["tx"], d["ty"])
        multi = mm.align_multi_start(sx, sy, starts)
        for a, s in zip(multi, starts):
            b = mm.align(sx, sy, s)                       # k_align_small
            assert a.status == b.status and np.abs(np.array(a.pose) - np.array(b.pose)).max() < 2e-5


def test_argument_checks(gpu_lib, pair2):
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = pair2
    sx, sy = _dev(d)
    with NdtMatcher2D() as mm:
        with pytest.raises(L.NdtError) as e:
            mm.align_multi_start(sx, sy, _starts(d["init"], 2))
        assert e.value.code == L.NDT_ERR_NO_TARGET
        mm.set_target(d["tx"], d["ty"])
        with pytest.raises(L.NdtError) as e:
            mm.align_multi_start(sx, sy, [d["init"]] * 65)
        assert e.value.code == L.NDT_ERR_INVALID_ARG
    # overlapping grids: the same chains with four lookups per point - the same contract
    with NdtMatcher2D(overlap_grids=4) as mm:
        mm.set_target(d["tx"], d["ty"])
        starts = _starts(d["init"], 3)
        multi = mm.align_multi_start(sx, sy, starts)
        scans = mm.align_multi_scan([(sx, sy)] * 2, starts[:2])
        single = [mm.align(sx, sy, st) for st in starts]
        assert all(_same(a, b) for a, b in zip(multi, single)) and all(_same(a, b) for a, b in zip(scans, single))


def _scans_of_config3(n_scans, n_pts, ragged=False):
    """Distinct scans taken in room (2,1) of the config-3 submap: other sampling seeds, slightly other poses."""
    import torch
    L, S, tiles = 50.0, 3, 4
    half = 0.5 * tiles * L
    room = synth.room_scene(S + 1000 * (1 * tiles + 2), L, 2 * L - half, 1 * L - half)
    centre = (2 * L - half + 0.5 * L, 1 * L - half + 0.5 * L)
    scans, inits, truth = [], [], []
    for k in range(n_scans):
        pose = (centre[0] + 0.1 - 0.01 * k, centre[1] - 0.08 + 0.005 * k, 0.01 - 0.001 * k)
        n = n_pts - 997 * k if ragged else n_pts
        x, y = synth.sample_scene(room, n, seed=40_000 + k, sigma=synth.SIGMA)
        x, y = synth.to_source_frame(x, y, pose)
        scans.append((torch.from_numpy(x.astype(np.float32)).cuda(), torch.from_numpy(y.astype(np.float32)).cuda()))
        inits.append((centre[0], centre[1], 0.0))
        truth.append(pose)
    return scans, inits, truth


@pytest.mark.parametrize("m,kw", [(3, {}), (7, {}), (13, dict(fixed_iterations=9)), (5, dict(hessian_mode=1)),
                                  (6, dict(overlap_grids=4)), (14, dict(overlap_grids=4, fixed_iterations=8))])
def test_multi_scan_every_scan_equals_its_own_alignment_bitwise(gpu_lib, m, kw):
    """ndt2d_align_multi_scan_dev: m different scans (ragged sizes) against one submap in one chain; scan k's
    result is bit for bit its own ndt2d_align_dev result, and the pose is the one the scan was taken at."""
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(3, n_tgt=400_000, n_src=1000)
    scans, inits, truth = _scans_of_config3(m, 30_000, ragged=True)
    with NdtMatcher2D(**kw) as mm:
        mm.set_target(d["tx"], d["ty"])
        multi = mm.align_multi_scan(scans, inits)
        for k, (a, (sx, sy)) in enumerate(zip(multi, scans)):
            b = mm.align(sx, sy, inits[k])
            assert _same(a, b), (k, a, b)
            if not kw:
                assert a.status == 0 and np.abs(np.array(a.pose) - np.array(truth[k])).max() < 5e-3


@pytest.mark.parametrize("kw", [dict(), dict(fixed_iterations=11), dict(hessian_mode=1), dict(line_search=3), dict(overlap_grids=4)])
def test_split_chain_equals_fused_chain_bitwise(gpu_lib, pair2, kw):
    """From 12 starts on a call runs two kernels per iteration (one workgroup per start solves, then everybody
    evaluates) instead of the fused kernel; both chains, forced for every size, give the same bits - the single-
    start alignment's - for shared and for per-start scans."""
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = pair2
    sx, sy = _dev(d)
    scans = [(sx[: 100_000 - 1111 * k].contiguous(), sy[: 100_000 - 1111 * k].contiguous()) for k in range(13)]
    res = {}
    for name, split_from in (("split", 1), ("fused", 1000)):
        with NdtMatcher2D(tuning={"split_from": split_from}, **kw) as mm:
            mm.set_target(d["tx"], d["ty"])
            res[name] = [mm.align_multi_start(sx, sy, _starts(d["init"], m)) for m in (2, 5, 8)]
            res[name].append(mm.align_multi_start(sx, sy, [OFFSETS[k % 8] for k in range(40)]))
            res[name].append(mm.align_multi_scan(scans, [d["init"]] * 13))
            if name == "split":
                single = [mm.align(sx, sy, s) for s in _starts(d["init"], 8)]
    for a_list, b_list in zip(res["split"], res["fused"]):
        assert len(a_list) == len(b_list)
        for a, b in zip(a_list, b_list):
            assert _same(a, b)
    for a, b in zip(res["split"][2], single):
        assert _same(a, b)
