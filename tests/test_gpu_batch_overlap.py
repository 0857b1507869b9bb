"""Biber's four overlapping grids (params.overlap_grids = 4) on the loop-closure batch path: every pair runs on the
global-table variant of the batch kernel with its four grids back to back (csrc/ndt2d_batch.hpp process_pair, NG = 4).
Each pair must equal the single-pair path with the same option (same records bit for bit, same per-point arithmetic:
float32 summation order is the only difference) and the CPU oracle (BASELINE tolerance)."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pairs():
    return [synth.make_pair(4, pair_index=k, n_tgt=20000, n_src=20000) for k in range(6)]


@pytest.mark.parametrize("mode", [0, 1])
def test_batch_overlap_equals_single_pair_and_oracle(gpu_lib, pairs, mode):
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
    from oracle import ndt2d as o
    prm = o.NdtParams(overlap=4, hessian_mode=mode)
    with NdtBatch2D(overlap_grids=4, hessian_mode=mode) as b:
        res = b.align([(p["tx"], p["ty"]) for p in pairs], [(p["sx"], p["sy"]) for p in pairs], [p["init"] for p in pairs])
    with NdtMatcher2D(overlap_grids=4, hessian_mode=mode) as m:
        for k, (p, r) in enumerate(zip(pairs, res)):
            m.set_target(p["tx"], p["ty"])
            s = m.align(p["sx"], p["sy"], p["init"])
            assert r.status == 0 == s.status
            # Full Newton steps on four overlapping grids creep towards the optimum (55-58 steps here, the float64 oracle
            # 19) and stop on the step-size test while the float32 sums still differ: the stopping points of two
            # summation orders lie 5e-5 m apart (both within 1e-4 of the oracle); Gauss-Newton stops at the same point.
            assert np.abs(np.array(r.pose) - np.array(s.pose)).max() < (2e-5 if mode == 0 else 1e-4)
            if mode == 0:
                assert abs(r.iterations - s.iterations) <= 1
            assert abs(r.n_hit - s.n_hit) <= 3                         # hits over all four grids
            assert abs(r.score - s.score) / s.score < 1e-4
            assert np.abs(r.H - s.H).max() / np.abs(s.H).max() < (2e-4 if mode == 0 else 5e-3)
            if k < 2:                                                  # the float64 oracle is slow: two pairs
                ref = o.align(o.build_grids(p["tx"], p["ty"], prm), p["sx"], p["sy"], p["init"], prm)
                assert ref["status"] == 0
                assert np.abs(np.array(r.pose) - np.array(ref["pose"])).max() < 1e-4       # BASELINE.json tolerance
                if mode == 0:
                    assert abs(r.iterations - ref["iterations"]) <= 3


@pytest.mark.parametrize("mode,k_fixed", [(0, 5), (1, 2)])
def test_batch_overlap_fixed_iterations_follow_the_single_pair_path(gpu_lib, pairs, mode, k_fixed):
    """(Full Newton steps from these starts are not contractive - the Hessian is indefinite away from the optimum - so a
    float32 rounding difference grows by an order of magnitude per step: two steps there, five Gauss-Newton steps.)"""
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
    with NdtBatch2D(overlap_grids=4, hessian_mode=mode, fixed_iterations=k_fixed) as b:
        res = b.align([(p["tx"], p["ty"]) for p in pairs], [(p["sx"], p["sy"]) for p in pairs], [p["init"] for p in pairs])
    with NdtMatcher2D(overlap_grids=4, hessian_mode=mode, fixed_iterations=k_fixed) as m:
        for p, r in zip(pairs, res):
            m.set_target(p["tx"], p["ty"])
            s = m.align(p["sx"], p["sy"], p["init"])
            assert r.iterations == k_fixed == s.iterations
            assert np.abs(np.array(r.pose) - np.array(s.pose)).max() < (5e-6 if mode == 0 else 2e-5)
            # (the Newton Hessian's rotation term carries the lever arm: 1e-6 rad between the two poses moves a point at
            # 20 m by 3e-5 m, a relative change of 1e-3 in terms whose scale is the 2 cm of a wall's thickness)
            assert np.abs(r.H - s.H).max() / np.abs(s.H).max() < (2e-4 if mode == 0 else 5e-3)
            assert abs(r.n_hit - s.n_hit) <= 3


@pytest.mark.parametrize("mode", [0, 1])
def test_batch_overlap_after_one_evaluation_is_the_single_pair_evaluation(gpu_lib, pairs, mode):
    """One fixed iteration: H, g, score and hits of the evaluation at the initial pose - the grids themselves."""
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
    with NdtBatch2D(overlap_grids=4, fixed_iterations=1, hessian_mode=mode) as b:
        res = b.align([(p["tx"], p["ty"]) for p in pairs], [(p["sx"], p["sy"]) for p in pairs], [p["init"] for p in pairs])
    with NdtMatcher2D(overlap_grids=4, hessian_mode=mode) as m:
        for p, r in zip(pairs, res):
            m.set_target(p["tx"], p["ty"])
            H, g, sc, nh = m.evaluate(p["sx"], p["sy"], p["init"])
            assert r.n_hit == nh                                      # the same cells are valid in all four grids
            assert abs(r.score - sc) / sc < 2e-5
            assert np.abs(r.H - H).max() / np.abs(H).max() < 2e-5


def test_batch_overlap_device_entry_ragged_edge_pairs_and_determinism(gpu_lib, pairs):
    """Device-pointer entry point: ragged sizes, a pair with a sparse target, a source that misses the map, a target
    beyond the tables (status from the device entry point); two runs and the reversed order give the same bits."""
    import torch
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
    p0, p1 = pairs[0], pairs[1]
    big = synth.make_pair(3, n_tgt=200000, n_src=2000)                # 404 x 404 cells: four grids exceed 2^18 cells
    targets = [(p0["tx"], p0["ty"]), (np.array([0.0, 10.0], np.float32), np.array([0.0, 10.0], np.float32)),
               (p1["tx"][:5000], p1["ty"][:5000]), (p0["tx"], p0["ty"]), (big["tx"], big["ty"])]
    sources = [(p0["sx"][:777], p0["sy"][:777]), (p0["sx"], p0["sy"]), (p1["sx"], p1["sy"]), (p0["sx"] + 1000.0, p0["sy"]),
               (big["sx"], big["sy"])]
    inits = [p0["init"], p0["init"], p1["init"], p0["init"], big["init"]]

    def pack(clouds):
        off = np.zeros(len(clouds) + 1, np.uint64)
        off[1:] = np.cumsum([len(c[0]) for c in clouds])
        return (torch.from_numpy(np.concatenate([c[0] for c in clouds])).cuda(), torch.from_numpy(np.concatenate([c[1] for c in clouds])).cuda(),
                torch.from_numpy(off.astype(np.int64)).cuda())

    def run(order):
        tx, ty, toff = pack([targets[i] for i in order])
        sx, sy, soff = pack([sources[i] for i in order])
        init = torch.from_numpy(np.array([inits[i] for i in order], np.float64)).cuda()
        with NdtBatch2D(overlap_grids=4, fixed_iterations=6) as b:
            out = b.align_dev(tx, ty, toff, sx, sy, soff, init)
            torch.cuda.synchronize()
            rows = b.decode(out)
        back = [None] * len(order)
        for j, i in enumerate(order):
            back[i] = rows[j]
        return back

    a, b2, c = run([0, 1, 2, 3, 4]), run([0, 1, 2, 3, 4]), run([4, 3, 2, 1, 0])
    assert a[1].status == L.NDT_TOO_FEW_CELLS
    assert a[3].status == L.NDT_TOO_FEW_HITS and a[3].iterations == 0
    assert a[4].status == L.NDT_ERR_CAPACITY
    for x, y, z in zip(a, b2, c):
        assert x.status == y.status == z.status
        assert x.pose == y.pose == z.pose
        np.testing.assert_array_equal(x.H, y.H)
        np.testing.assert_array_equal(x.H, z.H)
    with NdtMatcher2D(overlap_grids=4, fixed_iterations=6) as m:
        for k in (0, 2):
            m.set_target(*targets[k])
            s = m.align(*sources[k], inits[k])
            assert a[k].iterations == 6 == s.iterations
            assert np.abs(np.array(a[k].pose) - np.array(s.pose)).max() < 2e-5
    # the host-pointer entry point re-runs the pair beyond the tables through the single-pair path
    with NdtBatch2D(overlap_grids=4) as b:
        r = b.align([targets[4]], [sources[4]], [inits[4]])[0]
    with NdtMatcher2D(overlap_grids=4) as m:
        m.set_target(*targets[4])
        s = m.align(*sources[4], inits[4])
    assert r.status == s.status == 0 and np.abs(np.array(r.pose) - np.array(s.pose)).max() < 1e-6


def test_batch_overlap_with_grids_too_large_to_take_turns_on_chip(gpu_lib, pairs):
    """0.2 m cells on a 50 m room: 253 x 253 cells per grid, four of them just fit the tables (2^18 cells) but one grid's
    index slice and records no longer fit the LDS, so the loop gathers through L2 instead (process_pair: `turns`)."""
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
    kw = dict(overlap_grids=4, cell_size=0.2, fixed_iterations=4)
    with NdtBatch2D(**kw) as b:
        res = b.align([(p["tx"], p["ty"]) for p in pairs[:2]], [(p["sx"], p["sy"]) for p in pairs[:2]], [p["init"] for p in pairs[:2]])
    with NdtMatcher2D(**kw) as m:
        for p, r in zip(pairs, res):
            info = m.set_target(p["tx"], p["ty"])
            assert 4 * info.width * info.height <= 1 << 18 and 2 * info.width * info.height > 120 * 1024
            s = m.align(p["sx"], p["sy"], p["init"])
            assert r.status == s.status == 0 and r.iterations == 4
            assert np.abs(np.array(r.pose) - np.array(s.pose)).max() < 5e-6
            assert abs(r.n_hit - s.n_hit) <= 3
            assert np.abs(r.H - s.H).max() / np.abs(s.H).max() < 2e-4


def test_pyramid_with_an_overlapping_fine_level(gpu_lib, pairs):
    """Coarse single-grid levels on chip, the fine level with four grids: the chain of levels on the device equals the
    same chain made of single-pair alignments."""
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D, pyramid_params
    levels = pyramid_params()
    levels[2].overlap_grids = 4
    off = np.array([0.6, -0.5, 0.05])
    inits = [tuple(np.array(p["init"]) + off) for p in pairs[:3]]
    with NdtBatch2D(levels=levels) as b:
        res = b.align([(p["tx"], p["ty"]) for p in pairs[:3]], [(p["sx"], p["sy"]) for p in pairs[:3]], inits)
    for p, r, init in zip(pairs[:3], res, inits):
        pose, total = init, 0
        for lv in levels:
            with NdtMatcher2D(params=lv) as m:
                m.set_target(p["tx"], p["ty"])
                s = m.align(p["sx"], p["sy"], pose)
            pose, total = s.pose, total + s.iterations
        assert r.status == s.status == 0
        assert np.abs(np.array(r.pose) - np.array(pose)).max() < 5e-5
        assert abs(r.iterations - total) <= 3
        assert np.abs(np.array(r.pose) - np.array(p["pose"])).max() < 0.02      # and it is the right basin


def test_multi_device_contexts_with_overlapping_grids(gpu_lib, pairs):
    """ndt2d_multi_* (one batch context per device entry; two on device 0 here) with the option: bit for bit one context's results."""
    from gtsam_ndt_amd import matcher as M
    T = [(p["tx"], p["ty"]) for p in pairs[:5]]
    S = [(p["sx"], p["sy"]) for p in pairs[:5]]
    I = [p["init"] for p in pairs[:5]]
    with M.NdtBatch2D(overlap_grids=4) as b:
        ref = b.align(T, S, I)
    with M.NdtMulti2D(devices=[0, 0], overlap_grids=4) as mm:
        out = mm.align(T, S, I)
    for a, r in zip(out, ref):
        assert a.status == r.status == 0 and a.iterations == r.iterations and a.n_hit == r.n_hit
        assert a.pose == r.pose and np.array_equal(a.H, r.H)
