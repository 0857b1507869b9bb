"""The device-side workload generator (include/ndt_synth.h) against the numpy generator it
mirrors: same scenes and same scans, bit for bit.  (Generator only - no matcher code involved.)"""
import re
import os

import numpy as np
import pytest

from gtsam_ndt_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def synth_dev():
    from gtsam_ndt_amd import build, synth_dev
    build.build_synth()
    synth_dev.load()
    return synth_dev


def test_library_exports_every_declared_symbol(synth_dev):
    hdr = open(os.path.join(ROOT, "include", "ndt_synth.h")).read()
    declared = set(re.findall(r"\b(ndt_synth_\w+)\s*\(", hdr))
    assert declared == set(synth_dev.SIGNATURES)
    lib = synth_dev.load()
    for name in declared:
        assert getattr(lib, name) is not None


def test_room_scene_twin_is_bit_identical(synth_dev):
    for seed, L, x0, y0 in ((1, 8.0, -4.0, -4.0), (2, 50.0, -25.0, -25.0), (9000, 50.0, -25.0, -25.0),
                            (13095, 50.0, 0.0, -100.0), (3 + 1000 * 7, 50.0, 25.0, -50.0)):
        a = synth.room_scene(seed, L, x0, y0)
        b = synth_dev.room_scene(seed, L, x0, y0)
        for u, v in ((a.ax, b.ax), (a.ay, b.ay), (a.bx, b.bx), (a.by, b.by)):
            assert np.array_equal(u, v)


@pytest.mark.gpu
def test_config4_candidates_are_bit_identical_to_numpy(synth_dev, gpu_lib):
    """Full-size pairs (100k/100k) from both ends of the 4096-candidate list and a ragged size."""
    for first, n, nt, ns in ((0, 3, 100_000, 100_000), (4093, 3, 100_000, 100_000), (500, 2, 777, 1234)):
        t = synth_dev.config4_batch(first, n, nt, ns)
        h = {k: v.cpu().numpy() for k, v in t.items()}
        assert h["toff"].tolist() == [k * nt for k in range(n + 1)] and h["soff"].tolist() == [k * ns for k in range(n + 1)]
        for j in range(n):
            p = synth.make_pair(4, pair_index=first + j, n_tgt=nt, n_src=ns)
            assert np.array_equal(h["tx"][j * nt:(j + 1) * nt], p["tx"]) and np.array_equal(h["ty"][j * nt:(j + 1) * nt], p["ty"])
            assert np.array_equal(h["sx"][j * ns:(j + 1) * ns], p["sx"]) and np.array_equal(h["sy"][j * ns:(j + 1) * ns], p["sy"])
            assert tuple(h["pose"][j]) == p["pose"] and tuple(h["init"][j]) == p["init"]


@pytest.mark.gpu
def test_config4_strided_shard_equals_the_contiguous_candidates(synth_dev, gpu_lib):
    """A strided shard (pair k -> rank k mod world: bench.py --converged-batch) holds the same candidates, bit for bit,
    as the contiguous generator call produces for those indices."""
    nt, ns = 3000, 2500
    full = {k: v.cpu().numpy() for k, v in synth_dev.config4_batch(0, 12, nt, ns).items()}
    idx = list(range(1, 12, 4))
    t = {k: v.cpu().numpy() for k, v in synth_dev.config4_batch(0, 0, nt, ns, indices=idx).items()}
    assert t["toff"].tolist() == [j * nt for j in range(len(idx) + 1)] and t["soff"].tolist() == [j * ns for j in range(len(idx) + 1)]
    for j, k in enumerate(idx):
        for a, n in (("tx", nt), ("ty", nt), ("sx", ns), ("sy", ns)):
            assert np.array_equal(t[a][j * n:(j + 1) * n], full[a][k * n:(k + 1) * n])
        assert np.array_equal(t["pose"][j], full["pose"][k]) and np.array_equal(t["init"][j], full["init"][k])


@pytest.mark.gpu
def test_sample_scene_on_the_config3_submap_is_bit_identical(synth_dev, gpu_lib):
    """The 16-room submap scene (1216 segments) and the scan taken inside it, as make_pair(3) builds them."""
    d = synth.make_pair(3, n_tgt=300_000, n_src=50_000)
    L, S, tiles = 50.0, 3, 4
    half = 0.5 * tiles * L
    scene = None
    for j in range(tiles):
        for i in range(tiles):
            r = synth.room_scene(S + 1000 * (j * tiles + i), L, i * L - half, j * L - half)
            scene = r if scene is None else scene.concat(r)
    x, y = synth_dev.sample_scene(scene, 300_000, seed=S * 7919 + 11, sigma=synth.SIGMA)
    assert np.array_equal(x.cpu().numpy(), d["tx"]) and np.array_equal(y.cpu().numpy(), d["ty"])
    src_scene = synth.room_scene(S + 1000 * (1 * tiles + 2), L, 2 * L - half, 1 * L - half)
    sx, sy = synth_dev.sample_scene(src_scene, 50_000, seed=S * 7919 + 12, sigma=synth.SIGMA, pose=d["pose"])
    assert np.array_equal(sx.cpu().numpy(), d["sx"]) and np.array_equal(sy.cpu().numpy(), d["sy"])
    # a later window of the same stream: `first`
    x2, _ = synth_dev.sample_scene(scene, 1000, seed=S * 7919 + 11, sigma=synth.SIGMA, first=299_000)
    assert np.array_equal(x2.cpu().numpy(), d["tx"][299_000:])


@pytest.mark.gpu
def test_device_lidar3d_matches_numpy_generator(gpu_lib):
    """ndt_synth_lidar3d_dev vs synth3d.lidar_scan: same scene, beams and noise counters; float64 ray casting on both
    sides, the beam directions' cos / sin from two different libms - equal to float32 rounding, except that a ray
    grazing a box edge may hit on one side and miss on the other (a handful of beams at most)."""
    from gtsam_ndt_amd import synth3d, synth_dev
    for seed, pose, shape, scene in ((101, (0.0,) * 6, (64, 2048), 5), (7, (0.30, -0.20, 0.05, 0.01, -0.01, 0.03), (32, 512), 5),
                                     (9, (-1.5, 2.0, 0.1, -0.02, 0.015, 1.1), (16, 300), 12)):
        ref = synth3d.lidar_scan(seed, pose, shape[0], shape[1], 0.02, scene_seed=scene).astype(np.float32)
        x, y, z = (t.cpu().numpy() for t in synth_dev.lidar_scan3d(seed, pose, shape[0], shape[1], 0.02, scene_seed=scene))
        got = np.stack([x, y, z], axis=1)
        err = np.abs(got - ref).max(axis=1)
        assert (err > 1e-5).sum() <= 4, ((err > 1e-5).sum(), err.max())
        assert np.median(err) < 2e-6
        # firing order (all beams of one bearing, then the next bearing): the same points, permuted, bit for bit
        fx, fy, fz = (t.cpu().numpy() for t in synth_dev.lidar_scan3d(seed, pose, shape[0], shape[1], 0.02, scene_seed=scene,
                                                                       firing_order=True))
        for ring_major, firing in ((x, fx), (y, fy), (z, fz)):
            assert np.array_equal(ring_major.reshape(shape).T.reshape(-1), firing)
