"""The grid builds' cross-workgroup hand-offs (2D: lists of a shared tile's cells, 3D: slabs of a shared tile's sums; both
write-through stores + a ticket, no fences - MI355X_MICROARCH.md "Valid forms") under UNEVEN load: the same cloud is built
again and again on one handle while another stream keeps the chip busy with bursts of copies and alignments of varying
length; every build's map buffer (the exact integer sums of every cell) must equal the first one's bit for bit.  A lost or
stale word of a hand-off shows up as a different sum."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _noise(torch, rng, side, scratch):
    # a burst of device work of random length on another stream: copies (stream the L2s) and reductions
    with torch.cuda.stream(side):
        for _ in range(int(rng.integers(0, 4))):
            k = int(rng.integers(1 << 16, scratch.numel()))
            scratch[:k].copy_(scratch.flip(0)[:k])
            scratch[:k].sum()


def test_2d_builds_and_updates_repeat_bit_for_bit_under_load(gpu_lib):
    import torch
    from gtsam_ndt_amd import synth
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    rng = np.random.default_rng(5)
    d = synth.make_pair(2, n_tgt=300_000, n_src=60_000)               # 50 m room: 16 tiles, every one shared
    tx, ty, sx, sy = (torch.from_numpy(d[k]).cuda() for k in ("tx", "ty", "sx", "sy"))
    side, scratch = torch.cuda.Stream(), torch.rand(1 << 22, device="cuda")
    with NdtMatcher2D() as m:
        ref_build = ref_update = None
        for rep in range(120):
            _noise(torch, rng, side, scratch)
            m.set_target(tx, ty)
            a = m.save_map()
            _noise(torch, rng, side, scratch)
            m.add_target_points(sx, sy, pose=d["pose"])
            b = m.save_map()
            if ref_build is None:
                ref_build, ref_update = a, b
            else:
                assert np.array_equal(a, ref_build), rep
                assert np.array_equal(b, ref_update), rep
    torch.cuda.synchronize()


def test_3d_builds_and_updates_repeat_bit_for_bit_under_load(gpu_lib):
    import torch
    from gtsam_ndt_amd import synth_dev
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    rng = np.random.default_rng(6)
    t = synth_dev.lidar_scan3d(5, (0.0,) * 6, firing_order=True)      # config 5: dense tiles around the sensor, shared eightfold
    s = synth_dev.lidar_scan3d(6, (0.2, -0.1, 0.0, 0.0, 0.0, 0.01), firing_order=True)
    side, scratch = torch.cuda.Stream(), torch.rand(1 << 22, device="cuda")
    torch.cuda.synchronize()
    with NdtMatcher3D() as m, NdtMatcher3D(tuning={"single_sync_build": 0}) as m2:
        ref_build = ref_update = None
        for rep in range(120):
            _noise(torch, rng, side, scratch)
            m.set_target(*t)
            a = m.save_map()
            _noise(torch, rng, side, scratch)
            m.add_target_points(*s, pose=(0.2, -0.1, 0.0, 0.0, 0.0, 0.01))
            b = m.save_map()
            if rep % 10 == 0:                                           # the two-round-trip build, same bits
                m2.set_target(*t)
                assert np.array_equal(m2.save_map(), a), rep
            if ref_build is None:
                ref_build, ref_update = a, b
            else:
                assert np.array_equal(a, ref_build), rep
                assert np.array_equal(b, ref_update), rep
    torch.cuda.synchronize()
