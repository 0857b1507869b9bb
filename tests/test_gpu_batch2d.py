"""Loop-closure batch path (BASELINE config 4): every pair's result must equal the single-pair
path's and the CPU oracle's, independent of batching and of the queue order."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pairs():
    return [synth.make_pair(4, pair_index=k, n_tgt=20000, n_src=20000) for k in range(12)]


def _run_batch(gpu_lib, pairs, **kw):
    from gtsam_ndt_amd.matcher import NdtBatch2D
    with NdtBatch2D(**kw) as b:
        return b.align([(p["tx"], p["ty"]) for p in pairs], [(p["sx"], p["sy"]) for p in pairs],
                       [p["init"] for p in pairs])


def test_batch_matches_oracle_and_single_pair_path(gpu_lib, pairs):
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    from oracle import ndt2d as o
    res = _run_batch(gpu_lib, pairs)
    prm = o.NdtParams()
    with NdtMatcher2D() as m:
        for p, r in zip(pairs, res):
            ref = o.align(o.build_grid(p["tx"], p["ty"], prm), p["sx"], p["sy"], p["init"], prm)
            m.set_target(p["tx"], p["ty"])
            s = m.align(p["sx"], p["sy"], p["init"])
            assert r.status == 0 == ref["status"] == s.status
            e = np.abs(np.array(r.pose) - np.array(ref["pose"]))
            assert e[0] < 1e-4 and e[1] < 1e-4 and e[2] < 1e-4          # BASELINE.json tolerance
            assert np.abs(np.array(r.pose) - np.array(s.pose)).max() < 2e-5
            assert abs(r.iterations - ref["iterations"]) <= 3
            assert abs(r.n_hit - ref["n_hit"]) <= 3
            assert abs(r.score - ref["score"]) / ref["score"] < 1e-3
            assert np.abs(r.H - ref["H"]).max() / np.abs(ref["H"]).max() < 2e-3


def test_batch_fixed_iterations_and_determinism(gpu_lib, pairs):
    a = _run_batch(gpu_lib, pairs, fixed_iterations=7)
    b = _run_batch(gpu_lib, pairs[::-1], fixed_iterations=7)[::-1]     # other queue order
    for x, y in zip(a, b):
        assert x.iterations == 7 == y.iterations
        assert x.pose == y.pose                                         # bitwise: fixed trees, integer atomics
        np.testing.assert_array_equal(x.H, y.H)


def test_batch_ragged_and_edge_pairs(gpu_lib, pairs):
    """Pairs of different sizes in one batch; empty-ish, sparse and far-away pairs get their
    status without disturbing their neighbours."""
    from gtsam_ndt_amd import _lib as L
    p0, p1 = pairs[0], pairs[1]
    sparse_t = (np.array([0.0, 10.0], np.float32), np.array([0.0, 10.0], np.float32))
    targets = [(p0["tx"], p0["ty"]), sparse_t, (p1["tx"][:5000], p1["ty"][:5000]), (p0["tx"], p0["ty"])]
    sources = [(p0["sx"][:777], p0["sy"][:777]), (p0["sx"], p0["sy"]), (p1["sx"], p1["sy"]),
               (p0["sx"] + 1000.0, p0["sy"])]
    inits = [p0["init"], p0["init"], p1["init"], p0["init"]]
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
    with NdtBatch2D() as b:
        res = b.align(targets, sources, inits)
    assert res[1].status == L.NDT_TOO_FEW_CELLS
    assert res[3].status == L.NDT_TOO_FEW_HITS and res[3].iterations == 0
    with NdtMatcher2D() as m:
        for k in (0, 2):
            m.set_target(*targets[k])
            s = m.align(*sources[k], inits[k])
            assert res[k].status == s.status
            assert np.abs(np.array(res[k].pose) - np.array(s.pose)).max() < 5e-5


def test_batch_capacity_fallback(gpu_lib):
    """A target wider than the on-chip index table (here 200 m at 0.5 m cells) goes through the third
    variant of the batch kernel (tables in global memory), host-pointer entry point."""
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
    d = synth.make_pair(3, n_tgt=200000, n_src=20000)       # 200 m submap: 404 x 404 cells
    small = synth.make_pair(4, pair_index=3, n_tgt=20000, n_src=20000)
    with NdtBatch2D() as b:
        res = b.align([(d["tx"], d["ty"]), (small["tx"], small["ty"])],
                      [(d["sx"], d["sy"]), (small["sx"], small["sy"])], [d["init"], small["init"]])
    with NdtMatcher2D() as m:
        m.set_target(d["tx"], d["ty"])
        s = m.align(d["sx"], d["sy"], d["init"])
    assert res[0].status == s.status == 0 and np.abs(np.array(res[0].pose) - np.array(s.pose)).max() < 2e-5
    assert abs(res[0].iterations - s.iterations) <= 1 and abs(res[0].n_hit - s.n_hit) <= 2
    assert res[1].status == 0


def test_batch_device_pointer_entry_point(gpu_lib, pairs):
    """ndt2d_batch_align_dev on torch CUDA tensors (what bench.py and a multi-GPU host use):
    asynchronous on the caller's stream, results decoded from the device rows."""
    import torch
    from gtsam_ndt_amd import dist as nd
    from gtsam_ndt_amd.matcher import NdtBatch2D
    h = nd.pack_pairs(pairs)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(v).to(dev) for k, v in h.items()}
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    with NdtBatch2D() as b:
        with torch.cuda.stream(side):
            out = b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"],
                              stream=side.cuda_stream)
        side.synchronize()
        dev_rows = b.decode(out)
        host_rows = b.align([(p["tx"], p["ty"]) for p in pairs], [(p["sx"], p["sy"]) for p in pairs],
                            [p["init"] for p in pairs])
    for a_, b_ in zip(dev_rows, host_rows):
        assert a_.status == 0 and a_.pose == b_.pose and a_.iterations == b_.iterations


def test_batch_empty_clouds(gpu_lib, pairs):
    """Pairs with an empty source or an empty target (zero-length slices, first and last in the
    batch) report a status and touch no memory outside their slice."""
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtBatch2D
    p = pairs[0]
    e = np.zeros(0, np.float32)
    with NdtBatch2D() as b:
        res = b.align([(p["tx"], p["ty"]), (e, e), (p["tx"], p["ty"]), (p["tx"], p["ty"])],
                      [(e, e), (p["sx"], p["sy"]), (p["sx"], p["sy"]), (e, e)],
                      [p["init"]] * 4)
    assert res[0].status == L.NDT_TOO_FEW_HITS and res[3].status == L.NDT_TOO_FEW_HITS
    assert res[1].status == L.NDT_TOO_FEW_CELLS
    assert res[2].status == 0


def test_multi_context_equals_single_batch(gpu_lib):
    from gtsam_ndt_amd import matcher as M
    """ndt2d_multi_* with two contexts on device 0 (the box has one GPU): the sharded run returns
    exactly what one batch context returns, in the caller's pair order."""
    pairs = [synth.make_pair(4, pair_index=k, n_tgt=3000 + 500 * (k % 3), n_src=2500) for k in range(7)]
    T = [(p["tx"], p["ty"]) for p in pairs]
    S = [(p["sx"], p["sy"]) for p in pairs]
    I = [p["init"] for p in pairs]
    with M.NdtBatch2D() as b:
        ref = b.align(T, S, I)
    for devices in ([0], [0, 0], [0, 0, 0]):
        with M.NdtMulti2D(devices=devices) as mm:
            assert mm.device_count == len(devices)
            out = mm.align(T, S, I)
        for a, r in zip(out, ref):
            assert a.status == r.status and a.iterations == r.iterations and a.n_hit == r.n_hit
            assert a.pose == r.pose and np.array_equal(a.H, r.H)
    with M.NdtMulti2D() as mm:           # every visible device
        assert mm.device_count >= 1
        out = mm.align(T[:2], S[:2], I[:2])
        assert out[1].pose == ref[1].pose


def test_multi_context_rejects_bad_devices(gpu_lib):
    from gtsam_ndt_amd import matcher as M
    from gtsam_ndt_amd._lib import NdtError
    with pytest.raises(NdtError):
        M.NdtMulti2D(devices=[0, 99])


def _oracle_pyramid(p, init, levels):
    """The frozen composition: level k starts from level k-1's pose; a status other than OK /
    NOT_CONVERGED ends the pair; iterations add up (include/ndt_hip.h, ndt2d_batch_create_pyramid)."""
    from oracle import ndt2d as o
    pose, total, r = tuple(init), 0, None
    for lv in levels:
        prm = o.NdtParams(cell_size=lv.cell_size, min_points=lv.min_points, eig_ratio=lv.eig_ratio, d1=lv.d1, d2=lv.d2,
                          hessian_mode=lv.hessian_mode, max_iterations=lv.max_iterations,
                          fixed_iterations=lv.fixed_iterations, eps_trans=lv.eps_trans, eps_rot=lv.eps_rot,
                          step_max_trans=lv.step_max_trans, step_max_rot=lv.step_max_rot, min_hits=lv.min_hits)
        r = o.align(o.build_grid(p["tx"], p["ty"], prm), p["sx"], p["sy"], pose, prm)
        total += r["iterations"]
        if r["status"] not in (o.NDT_OK, o.NDT_NOT_CONVERGED):
            break
        pose = r["pose"]
    r["iterations"] = total
    return r


def test_batch_pyramid_matches_oracle_composition_and_widens_the_basin(gpu_lib, pairs):
    """Coarse-to-fine over the batch (one launch per level, poses chained on the device)."""
    from gtsam_ndt_amd import matcher as M
    levels = M.pyramid_params()
    # guesses 0.5 m / 0.04 rad off the generating pose: outside the single-level basin
    rng = np.random.default_rng(3)
    sub = pairs[:8]
    inits = [tuple(np.array(p["pose"]) + rng.uniform([-0.5, -0.5, -0.04], [0.5, 0.5, 0.04])) for p in sub]
    T = [(p["tx"], p["ty"]) for p in sub]
    S = [(p["sx"], p["sy"]) for p in sub]
    with M.NdtBatch2D(levels=levels) as b:
        res = b.align(T, S, inits)
    with M.NdtBatch2D() as b1:
        flat = b1.align(T, S, inits)
    near = lambda r, p: np.abs(np.array(r.pose) - np.array(p["pose"])).max() < 0.02
    n_pyr = sum(near(r, p) for r, p in zip(res, sub))
    n_flat = sum(near(r, p) for r, p in zip(flat, sub))
    assert n_pyr >= n_flat and n_pyr >= 6, (n_pyr, n_flat)
    for p, init, r in zip(sub, inits, res):
        ref = _oracle_pyramid(p, init, levels)
        assert r.status == ref["status"]
        if near(r, p):          # trajectories that reach the optimum are stable; compare those to 1e-4
            e = np.abs(np.array(r.pose) - np.array(ref["pose"]))
            assert e.max() < 1e-4, (r.pose, ref["pose"])
            assert abs(r.iterations - ref["iterations"]) <= 6
    # the multi-device context takes the same schedule
    with M.NdtMulti2D(devices=[0, 0], levels=levels) as mm:
        rm = mm.align(T, S, inits)
    assert [r.pose for r in rm] == [r.pose for r in res]
    # a level that fails ends the pair with that status: 3 target points give no valid cell at any level
    with M.NdtBatch2D(levels=levels) as b:
        bad = b.align([(sub[0]["tx"][:3], sub[0]["ty"][:3]), T[1]], [S[0], S[1]], inits[:2])
    assert bad[0].status == 4 and bad[0].iterations == 0
    assert bad[1].pose == res[1].pose


def test_small_pair_variant_equals_large_variant_and_oracle(gpu_lib):
    """Lidar-sized pairs run on the 256-thread variant of the batch kernel (three pairs per CU);
    pairs over its limits (points, cells, occupied cells) are left to the 1024-thread variant in
    the same call.  Same results as with the small variant switched off, and as the oracle."""
    from gtsam_ndt_amd import matcher as M
    from oracle import ndt2d as o
    sizes = [(1000, 1000), (360, 360), (4000, 3000), (8192, 8192), (8193, 500), (500, 8193), (20000, 20000), (2000, 2000),
             (3, 50), (1500, 0)]
    pairs = []
    for k, (nt, ns) in enumerate(sizes):
        p = synth.make_pair(4, pair_index=40 + k, n_tgt=max(nt, 1), n_src=max(ns, 1))
        if ns == 0:
            p["sx"], p["sy"] = p["sx"][:0], p["sy"][:0]
        pairs.append(p)
    T = [(p["tx"], p["ty"]) for p in pairs]
    S = [(p["sx"], p["sy"]) for p in pairs]
    I = [p["init"] for p in pairs]
    res = {}
    for off in ("0", "1"):
        with M.NdtBatch2D(small_variant=(off == "0")) as b:
            res[off] = b.align(T, S, I)
            # (8193, 500), (500, 8193), (20000, 20000) exceed the point limit; (8192, 8192) occupies more
            # than 767 cells; the others fit the small variant
            assert b.last_large_count == (len(T) if off == "1" else 4), b.last_large_count
            again = b.align(T, S, I)
            assert [r.pose for r in again] == [r.pose for r in res[off]]          # deterministic, marks reset
    prm = o.NdtParams()
    for k, (p, a, b) in enumerate(zip(pairs, res["0"], res["1"])):
        assert a.status == b.status, (k, a.status, b.status)
        if a.status in (0, 1):
            # float32 summation order differs between the variants; sparse scans stop a step apart
            assert abs(a.iterations - b.iterations) <= 2 and abs(a.n_hit - b.n_hit) <= 1, k
            assert np.abs(np.array(a.pose) - np.array(b.pose)).max() < 2e-5, k
            if len(p["tx"]) >= 1000:
                ref = o.align(o.build_grid(p["tx"], p["ty"], prm), p["sx"], p["sy"], p["init"], prm, mirror32=True)
                assert np.abs(np.array(a.pose) - np.array(ref["pose"])).max() < 1e-5, k
    # a cell size that needs more than 64 x 64 cells: the small variant hands the pair over
    for off in ("0", "1"):
        with M.NdtBatch2D(cell_size=0.2, small_variant=(off == "0")) as b:
            res[off] = b.align(T[:3], S[:3], I[:3])
    for a, b in zip(res["0"], res["1"]):
        assert a.status == b.status and np.abs(np.array(a.pose) - np.array(b.pose)).max() < 2e-5
    # coarse-to-fine over a mixed batch
    with M.NdtBatch2D(levels=M.pyramid_params()) as b:
        pyr = b.align(T[:8], S[:8], I[:8])
    with M.NdtBatch2D(levels=M.pyramid_params(), small_variant=False) as b:
        pyr_l = b.align(T[:8], S[:8], I[:8])
    for a, b in zip(pyr, pyr_l):
        assert a.status == b.status and abs(a.iterations - b.iterations) <= 3
        assert np.abs(np.array(a.pose) - np.array(b.pose)).max() < 2e-5


def test_multi_device_context_with_rccl_gather(gpu_lib, pairs):
    """ndt2d_multi_align_dev: shards resident on their devices, rows exchanged by ncclAllGather on the
    contexts' streams.  One GPU here, so one rank (the gather is a copy through RCCL); the sharding
    arithmetic for more ranks is covered by the gloo and plan tests."""
    import torch
    from gtsam_ndt_amd import _lib as L, dist as nd
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMulti2D
    t = {k: torch.from_numpy(v).cuda() for k, v in nd.pack_pairs(pairs).items()}
    with NdtBatch2D() as b:
        want = b.decode(b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"]))
    with NdtMulti2D(devices=[0]) as m:
        got = m.align_dev([t])
        assert m.last_shard_stride == len(pairs)
        again = m.align_dev([t])                      # communicator and buffers are reused
    for a, b_, c in zip(got, want, again):
        assert a.status == 0 and a.pose == b_.pose == c.pose and a.iterations == b_.iterations
        assert np.array_equal(a.H, b_.H)
    with NdtMulti2D(devices=[0, 0]) as m:             # a device cannot gather with itself
        with pytest.raises(L.NdtError) as e:
            m.align_dev([t, None])
        assert e.value.code == L.NDT_ERR_INVALID_ARG


def test_device_entry_point_handles_pairs_over_the_on_chip_capacity(gpu_lib, pairs):
    """ndt2d_batch_align_dev never fails a pair a front end may legally produce: a scan against a 200 m
    submap (404 x 404 cells, far over the 128 x 128-cell LDS table) is handed, inside the same call, to the
    third variant of the kernel (tables in global memory) and gets the single-pair path's result; a 100 m
    room at 0.5 m cells (over 16384 cells) likewise; the lidar-sized and config-4-sized neighbours in the
    batch are untouched.  Only a grid beyond 512 x 512 cells still reports NDT_ERR_CAPACITY."""
    import torch
    from gtsam_ndt_amd import _lib as L, dist as nd
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
    big = synth.make_pair(3, n_tgt=300_000, n_src=30_000)                 # 200 m submap
    sc = synth.room_scene(77, 100.0, -50.0, -50.0)                        # 100 m room: 202 x 202 cells
    x, y = synth.sample_scene(sc, 150_000, seed=5, sigma=0.03)
    xs, ys = synth.sample_scene(sc, 40_000, seed=6, sigma=0.03)
    xs, ys = synth.to_source_frame(xs, ys, (0.08, -0.05, 0.008))
    room = {"tx": x.astype(np.float32), "ty": y.astype(np.float32), "sx": xs.astype(np.float32), "sy": ys.astype(np.float32),
            "init": (0.0, 0.0, 0.0)}
    huge = {"tx": np.array([0.0, 0.1, 0.2, 400.0, 400.1, 400.2], np.float32), "ty": np.array([0.0, 0.1, 0.0, 300.0, 300.1, 300.0], np.float32),
            "sx": pairs[0]["sx"][:100], "sy": pairs[0]["sy"][:100], "init": (0.0, 0.0, 0.0)}     # 802 x 602 cells
    batch = [pairs[0], big, pairs[1], room, huge, pairs[2]]
    t = {k: torch.from_numpy(v).cuda() for k, v in nd.pack_pairs(batch).items()}
    for kw in (dict(), dict(fixed_iterations=9), dict(small_variant=False)):
        with NdtBatch2D(**kw) as b:
            rows = b.decode(b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"]))
            again = b.decode(b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"]))
        assert [r.pose for r in rows] == [r.pose for r in again]                 # deterministic, marks reset
        assert rows[4].status == L.NDT_ERR_CAPACITY
        mkw = {k: v for k, v in kw.items() if k != "small_variant"}
        with NdtMatcher2D(**mkw) as m:
            for k in (0, 1, 2, 3, 5):
                p = batch[k]
                m.set_target(p["tx"], p["ty"])
                s = m.align(p["sx"], p["sy"], p["init"])
                assert rows[k].status == s.status == 0 and abs(rows[k].iterations - s.iterations) <= 1, (k, rows[k], s)
                assert np.abs(np.array(rows[k].pose) - np.array(s.pose)).max() < 2e-5, (k, rows[k].pose, s.pose)
                assert abs(rows[k].n_hit - s.n_hit) <= 2
    # the variant's workgroup count (one table slab each) is a memory / rate knob: results do not depend on it
    with NdtBatch2D() as b:
        ref = b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"]).cpu()
        for wgs in (1, 64):
            b.set_tuning("batch_global_workgroups", wgs)
            assert torch.equal(b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"]).cpu(), ref), wgs
        for bad in (0, 257):
            with pytest.raises(L.NdtError) as e:
                b.set_tuning("batch_global_workgroups", bad)
            assert e.value.code == L.NDT_ERR_INVALID_ARG
    # coarse-to-fine over the same batch: every level hands the big pairs over again
    from gtsam_ndt_amd.matcher import pyramid_params
    with NdtBatch2D(levels=pyramid_params()) as b:
        rows = b.decode(b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"]))
    assert all(rows[k].status == 0 for k in (0, 1, 2, 3, 5))
    assert np.abs(np.array(rows[1].pose) - np.array(big["pose"])).max() < 3e-3
