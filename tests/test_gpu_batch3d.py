"""3D loop-closure batch (k_batch3, ndt3d_batch_*) vs the single-pair 3D path and the CPU oracle.
Parity unpinned: the oracle is this repo's own (the reference holds no code)."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth3d

pytestmark = pytest.mark.gpu

POSES = [(0.30, -0.20, 0.05, 0.01, -0.01, 0.03), (-0.25, 0.15, -0.04, -0.008, 0.012, -0.02),
         (0.10, 0.28, 0.02, 0.0, 0.015, 0.035), (-0.12, -0.22, 0.06, 0.012, 0.0, -0.03),
         (0.22, 0.05, -0.03, -0.01, -0.012, 0.015), (0.0, 0.0, 0.0, 0.0, 0.0, 0.0)]
SHAPES = [(16, 256), (32, 512), (16, 512), (24, 384), (32, 256), (16, 300)]     # ragged: 4096 ... 16384 points


def _pairs(poses=POSES, shapes=SHAPES):
    ds = [synth3d.make_pair3d(n_elev=e, n_azim=a, pose=p) for p, (e, a) in zip(poses, shapes)]
    T = [(d["tx"], d["ty"], d["tz"]) for d in ds]
    S = [(d["sx"], d["sy"], d["sz"]) for d in ds]
    return ds, T, S


def _single(T, S, inits, **kw):
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    out = []
    with NdtMatcher3D(**kw) as m:
        for t, s, i in zip(T, S, inits):
            m.set_target(*t)
            out.append(m.align(*s, i))
    return out


def _same(rb, rs, pose_tol=2e-6, h_tol=2e-4, iter_tol=1):
    """batch result vs single-pair result: same records, same sums in another form and summation order"""
    assert rb.status == rs.status and abs(rb.iterations - rs.iterations) <= iter_tol, (rb, rs)
    assert abs(rb.n_hit - rs.n_hit) <= 2
    assert np.abs(np.array(rb.pose) - np.array(rs.pose)).max() < pose_tol, (rb.pose, rs.pose)
    sc = np.sqrt(np.outer(np.abs(np.diag(rs.H)), np.abs(np.diag(rs.H)))) + 1e-30
    assert np.max(np.abs(rb.H - rs.H) / sc) < h_tol
    assert abs(rb.score - rs.score) <= 1e-4 * abs(rs.score)


@pytest.mark.parametrize("mode", [0, 1])
def test_batch3d_equals_single_pair_and_oracle(gpu_lib, mode):
    from gtsam_ndt_amd.matcher import NdtBatch3D
    from oracle import ndt3d as o
    ds, T, S = _pairs()
    inits = [d["init"] for d in ds]
    if mode == 1:          # Newton's basin is small (tests/test_gpu_ndt3d.py): start next to the Gauss-Newton optimum
        gn = _single(T, S, inits)
        inits = [tuple(np.array(r.pose) + 2e-3 * np.array([1, -1, 0.5, 0.1, -0.1, 0.2])) for r in gn]
    with NdtBatch3D(hessian_mode=mode) as b:
        rb = b.align(T, S, inits)
        rb2 = b.align(T, S, inits)
    rs = _single(T, S, inits, hessian_mode=mode)
    for k, (x, y) in enumerate(zip(rb, rs)):
        _same(x, y)
        assert x.pose == rb2[k].pose and np.array_equal(x.H, rb2[k].H)      # deterministic, whichever CU took the pair
    prm = o.Ndt3Params(hessian_mode=mode)
    for k in (0, 3):
        ref = o.align3(o.build_grid3(*T[k], prm), *S[k], inits[k], prm)
        e = np.abs(np.array(rb[k].pose) - np.array(ref["pose"]))
        assert rb[k].status == ref["status"] == 0 and e.max() < 1e-4        # BASELINE.json: 1e-4 m / 1e-4 rad
        assert abs(rb[k].iterations - ref["iterations"]) <= 3


def test_batch3d_full_config5_pairs(gpu_lib):
    """Full-size pairs (131 072 points each, 17 424 voxels, 2 706 occupied: the LDS carve of config 5), fixed 10
    iterations: oracle parity on one of them, the single-pair path on all."""
    from gtsam_ndt_amd.matcher import NdtBatch3D
    from oracle import ndt3d as o
    ds, T, S = _pairs(POSES[:3], [(64, 2048)] * 3)
    inits = [d["init"] for d in ds]
    with NdtBatch3D(fixed_iterations=10) as b:
        rb = b.align(T, S, inits)
    rs = _single(T, S, inits, fixed_iterations=10)
    for x, y in zip(rb, rs):
        assert x.iterations == 10 and x.status == 0
        # mid-descent, 128 points per thread summed in float32 (the single-pair kernel: 2): the poses drift 2e-6 m
        # apart, and with 9 mm thin voxel Gaussians that moves H by 2 q dp / sigma^2 = 5e-4
        _same(x, y, pose_tol=1e-5, h_tol=3e-3)
    prm = o.Ndt3Params(fixed_iterations=10)
    ref = o.align3(o.build_grid3(*T[1], prm), *S[1], inits[1], prm)
    e = np.abs(np.array(rb[1].pose) - np.array(ref["pose"]))
    es = np.abs(np.array(rs[1].pose) - np.array(ref["pose"]))
    print("full-size pair after 10 iterations, |pose - oracle|: batch", e.max(), "single-pair", es.max())
    assert e.max() < 1e-4
    with NdtBatch3D() as b:
        rc = b.align(T, S, inits)
    for r, d in zip(rc, ds):
        e = np.abs(np.array(r.pose) - np.array(d["pose"]))
        assert r.status == 0 and e[:3].max() < 5e-3 and e[3:].max() < 1e-3   # recovers the generating pose


def test_batch3d_options_and_edge_cases(gpu_lib):
    from gtsam_ndt_amd.matcher import NdtBatch3D
    ds, T, S = _pairs()
    inits = [d["init"] for d in ds]
    # line search and over-relaxation go through the same gn_update3
    kw = dict(line_search=4, step_scale=1.5)
    with NdtBatch3D(**kw) as b:
        rb = b.align(T, S, inits)
    for x, y in zip(rb, _single(T, S, inits, **kw)):
        _same(x, y)
    # edge cases: a source that misses the map, a target with too few points per voxel, an empty source
    far = tuple(np.asarray(c) + np.float32(500.0) for c in S[0])
    tiny = tuple(np.asarray(c)[:4] for c in T[1])            # fewer than min_points anywhere
    empty = tuple(np.zeros(0, np.float32) for _ in range(3))
    with NdtBatch3D() as b:
        r = b.align([T[0], tiny, T[2], T[3]], [far, S[1], empty, S[3]], [inits[0], inits[1], inits[2], inits[3]])
    assert r[0].status == 3 and r[0].n_hit == 0                    # NDT_TOO_FEW_HITS
    assert r[1].status == 4                                        # NDT_TOO_FEW_CELLS
    assert r[2].status == 3 and r[2].iterations == 0
    _same(r[3], _single([T[3]], [S[3]], [inits[3]])[0])            # the pair behind them is untouched by their exits


def _dev_args(T, S, inits):
    import torch
    dev = torch.device("cuda:0")
    cat = lambda cl, a: torch.from_numpy(np.concatenate([np.asarray(c[a], np.float32) for c in cl])).to(dev)
    off = lambda cl: torch.tensor(np.concatenate([[0], np.cumsum([len(c[0]) for c in cl])]), dtype=torch.int64, device=dev)
    return ([cat(T, a) for a in range(3)], off(T), [cat(S, a) for a in range(3)], off(S),
            torch.tensor(np.array(inits), dtype=torch.float64, device=dev))


@pytest.mark.parametrize("mode", [0, 1])
def test_batch3d_capacity_handover_on_the_device(gpu_lib, mode):
    """0.25 m voxels: 170 x 170 x 30 voxels do not fit the LDS carve.  k_batch3 hands such pairs to the variant
    with its tables in global memory inside the same call - through the device entry point too - while the
    pairs that fit stay on chip (a mixed batch: cell size is per context, so the small pairs here are cropped
    scans whose grid fits)."""
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtBatch3D
    ds, T, S = _pairs(POSES[:3], [(32, 512), (16, 256), (16, 256)])
    crop = lambda c, r: tuple(np.asarray(a)[(np.abs(c[0]) < r) & (np.abs(c[1]) < r)] for a in c)
    T[2], S[2] = crop(T[2], 6.0), crop(S[2], 6.5)               # 12 m x 12 m x 6 m at 0.25 m: 50 x 50 x 26 voxels, on chip
    inits = [d["init"] for d in ds]
    kw = dict(cell_size=0.25, min_points=3, step_max_trans=0.25, hessian_mode=mode)
    if mode == 1:          # Newton iterations on 0.25 m voxels are chaotic: one evaluation and update next to the GN optimum
        gn = _single(T, S, inits, cell_size=0.25, min_points=3, step_max_trans=0.25)
        inits = [tuple(np.array(r.pose) + 5e-4 * np.array([1, -1, 0.5, 0.1, -0.1, 0.2])) for r in gn]
        kw["fixed_iterations"] = 1
    with NdtBatch3D(**kw) as b:
        rd = b.decode(b.align_dev(*_dev_args(T, S, inits)))
        rh = b.align(T, S, inits)
        # the variant's workgroup count (one 7.9 MB table slab each) is a memory / rate knob: the same bits with 1
        b.set_tuning("batch_global_workgroups", 1)
        r1 = b.decode(b.align_dev(*_dev_args(T, S, inits)))
        for bad in (0, 257):
            with pytest.raises(L.NdtError):
                b.set_tuning("batch_global_workgroups", bad)
    rs = _single(T, S, inits, **kw)
    for x, h, y, z in zip(rd, rh, rs, r1):
        assert x.status == y.status and x.status in (0, 1)
        _same(x, y, pose_tol=5e-6)
        assert x.pose == h.pose and np.array_equal(x.H, h.H)       # host entry point: the same kernels
        assert x.pose == z.pose and np.array_equal(x.H, z.H)


def test_batch3d_beyond_the_global_tables(gpu_lib):
    """0.1 m voxels over the 40 m room: 425 x 425 x 75 voxels exceed the global-memory variant too (2^20 voxels).
    The device entry point says so per pair; the host entry point re-runs the pair through the single-pair path."""
    from gtsam_ndt_amd.matcher import NdtBatch3D
    ds, T, S = _pairs(POSES[:2], [(32, 512), (16, 256)])
    inits = [d["init"] for d in ds]
    kw = dict(cell_size=0.1, min_points=3, step_max_trans=0.1)
    with NdtBatch3D(**kw) as b:
        rd = b.decode(b.align_dev(*_dev_args(T, S, inits)))
        rb = b.align(T, S, inits)
    assert [r.status for r in rd] == [-5, -5]                      # NDT_ERR_CAPACITY
    for x, y in zip(rb, _single(T, S, inits, **kw)):
        assert x.status == y.status and x.pose == y.pose and np.array_equal(x.H, y.H)


def test_batch3d_pyramid_and_device_entry(gpu_lib):
    """Coarse to fine over the batch (4 m -> 2 m -> 1 m voxels, one launch per level) from 1.2 m / 0.08 rad off,
    through the device-pointer entry point on torch's stream."""
    import torch
    from gtsam_ndt_amd.matcher import NdtBatch3D, default_params3d, PYRAMID_LEVELS
    ds, T, S = _pairs(POSES[:3], [(32, 1024)] * 3)
    true = [np.array(d["pose"]) for d in ds]
    inits = [tuple(t + np.array([1.2, -0.9, 0.15, 0.0, 0.0, 0.08]) * s) for t, s in zip(true, (1.0, -1.0, 0.7))]
    fine = default_params3d()
    levels = []
    for mult, er in PYRAMID_LEVELS:
        levels.append(default_params3d(cell_size=fine.cell_size * mult, eig_ratio=er, eps_trans=1e-3, eps_rot=1e-4,
                                       max_iterations=30, step_max_trans=fine.step_max_trans * mult))
    levels.append(fine)
    with NdtBatch3D(levels=levels) as b, NdtBatch3D() as flat:
        args = _dev_args(T, S, inits)
        rp = b.decode(b.align_dev(*args))
        rf = flat.decode(flat.align_dev(*args))
        rh = b.align(T, S, inits)                                   # host entry point, same levels
    for r, h, f, t in zip(rp, rh, rf, true):
        e = np.abs(np.array(r.pose) - t)
        assert r.status == 0 and e[:3].max() < 0.03 and e[3:].max() < 5e-3, (r.pose, t)
        assert r.pose == h.pose and r.iterations == h.iterations
        assert np.abs(np.array(f.pose) - t)[:3].max() > e[:3].max()       # the 1 m grid alone ends elsewhere


@pytest.mark.parametrize("mode", [0, 1])
def test_batch3d_records_beyond_the_carve_overflow_to_the_slab(gpu_lib, mode):
    """0.9 m voxels on full-size scans: 48 x 48 x 10 voxels, 3 309 occupied - 127 more than the 3 182 records the LDS carve
    holds beside that table.  The pair stays on the on-chip kernel: the last records live in the slab and are gathered
    through L2.  Same results as the single-pair path; a second pair that fits rides along in the same launch."""
    from gtsam_ndt_amd.matcher import NdtBatch3D
    ds, T, S = _pairs(POSES[:2], [(64, 2048), (32, 512)])
    inits = [d["init"] for d in ds]
    kw = dict(cell_size=0.9, step_max_trans=0.9, hessian_mode=mode)
    if mode == 1:
        gn = _single(T, S, inits, cell_size=0.9, step_max_trans=0.9)
        inits = [tuple(np.array(r.pose) + 1e-3 * np.array([1, -1, 0.5, 0.1, -0.1, 0.2])) for r in gn]
        kw["fixed_iterations"] = 1
    with NdtBatch3D(**kw) as b:
        rb = b.align(T, S, inits)
        rb2 = b.align(T, S, inits)
    rs = _single(T, S, inits, **kw)
    for x, x2, y in zip(rb, rb2, rs):
        assert x.status == y.status and x.status in (0, 1)
        _same(x, y, pose_tol=1e-5, h_tol=3e-3)
        assert x.pose == x2.pose and np.array_equal(x.H, x2.H)


def test_multi_device_3d_context_host_shards_and_rccl_gather(gpu_lib):
    """ndt3d_multi_*: the host-pointer form with device 0 listed twice (two contexts, two host threads, work-balanced
    contiguous shards) and the device-resident form with its RCCL all-gather (one GPU here: one rank) equal the
    single-context batch bit for bit."""
    import torch
    from gtsam_ndt_amd import _lib as L, dist as nd
    from gtsam_ndt_amd.matcher import NdtBatch3D, NdtMulti3D
    ds, T, S = _pairs()
    inits = [d["init"] for d in ds]
    with NdtBatch3D() as b:
        want = b.align(T, S, inits)
    with NdtMulti3D(devices=[0, 0]) as m:
        assert m.device_count == 2
        got = m.align(T, S, inits)
    for a, w in zip(got, want):
        assert a.status == w.status and a.pose == w.pose and np.array_equal(a.H, w.H) and a.iterations == w.iterations
    b3 = nd.pack_pairs3d(ds)
    cu = lambda v: torch.from_numpy(v).cuda()
    shard = {"t": [cu(v) for v in b3["t"]], "toff": cu(b3["toff"]), "s": [cu(v) for v in b3["s"]], "soff": cu(b3["soff"]),
             "init": cu(b3["init"])}
    with NdtMulti3D(devices=[0]) as m:
        got = m.align_dev([shard])
        assert m.last_shard_stride == len(ds)
        again = m.align_dev([shard])                  # communicator and buffers are reused
    for a, w, c in zip(got, want, again):
        assert a.status == w.status and a.pose == w.pose == c.pose and np.array_equal(a.H, w.H)
    with NdtMulti3D(devices=[0, 0]) as m:             # a device cannot gather with itself
        with pytest.raises(L.NdtError) as e:
            m.align_dev([shard, None])
        assert e.value.code == L.NDT_ERR_INVALID_ARG


def test_batch3d_does_not_depend_on_the_order_of_the_points(gpu_lib):
    """The build walks the target in rows of 64 points and adds runs of one voxel up in registers before it touches LDS
    (fast on scans in firing order); its sums are exact integers, so any order of the same points gives the same grid,
    and the alignment differs by float32 summation order only: ring by ring, firing order, shuffled."""
    from gtsam_ndt_amd.matcher import NdtBatch3D
    ds, T, S = _pairs(POSES[:2], [(64, 512), (16, 300)])
    inits = [d["init"] for d in ds]
    rng = np.random.default_rng(3)
    def reorder(c, how, shape):
        n = len(c[0])
        if how == "firing":
            perm = np.arange(n).reshape(shape).T.reshape(-1)
        else:
            perm = rng.permutation(n)
        return tuple(np.ascontiguousarray(np.asarray(a)[perm]) for a in c)
    shapes = [(64, 512), (16, 300)]
    with NdtBatch3D() as b:
        ref = b.align(T, S, inits)
        for how in ("firing", "shuffled"):
            T2 = [reorder(c, how, sh) for c, sh in zip(T, shapes)]
            S2 = [reorder(c, how, sh) for c, sh in zip(S, shapes)]
            for x, y in zip(b.align(T2, S2, inits), ref):
                assert x.status == y.status == 0 and abs(x.iterations - y.iterations) <= 1 and abs(x.n_hit - y.n_hit) <= 2
                assert np.abs(np.array(x.pose) - np.array(y.pose)).max() < 5e-6, (how, x.pose, y.pose)
