"""Stream ordering of the device-pointer entry points: a handle enqueues on its own non-blocking
stream, so device arrays still being written on the caller's stream must be waited for
(ndt2d_wait_stream / ndt2d_batch_wait_stream / ndt3d_wait_stream; the Python wrappers call them
with torch's current stream).  The tests delay the producer on purpose and never synchronise the
host between producing the scan and aligning it."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth

pytestmark = pytest.mark.gpu

POSE = (9.5, 10.8, 0.1)
SLEEP_CYCLES = 200_000_000          # ~0.1 s of torch.cuda._sleep on the producer stream


def _poison_allocator(n):
    """Leave NaN in the cached blocks the next torch.empty(n) calls will be handed."""
    import torch
    junk = [torch.full((n,), float("nan"), dtype=torch.float32, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    del junk


def test_align_of_a_scan_still_being_converted(gpu_lib):
    import torch
    from gtsam_ndt_amd.matcher import NdtMatcher2D, polar_to_points
    sc = synth.room_scene(4242, 30.0)
    r0, a0, da = synth.lidar_scan2d(sc, (8.0, 10.0, 0.0), n_beams=7200, seed=100)
    x0, y0 = synth.scan_points(r0, a0, da)
    ok = ~np.isnan(x0)
    tx, ty = (x0[ok] + np.float32(8.0)).astype(np.float32), (y0[ok] + np.float32(10.0)).astype(np.float32)
    r1, a1, da1 = synth.lidar_scan2d(sc, POSE, n_beams=7200, seed=101)
    ranges = torch.from_numpy(r1).cuda()
    guess = (POSE[0] + 0.05, POSE[1] - 0.04, POSE[2] + 0.01)
    with NdtMatcher2D() as m:
        m.set_target(tx, ty)
        dx, dy = polar_to_points(ranges, a1, da1, 0.05, 30.0)
        torch.cuda.synchronize()
        want = m.align(dx, dy, guess)                         # everything complete: the reference run
        assert want.status == 0
        del dx, dy
        for use_async in (False, True):
            _poison_allocator(ranges.numel())
            torch.cuda._sleep(SLEEP_CYCLES)                   # the producer stream is busy ...
            dx, dy = polar_to_points(ranges, a1, da1, 0.05, 30.0)   # ... so the conversion has not run yet
            if use_async:
                m.align_async(dx, dy, guess)
                got = m.finish()
            else:
                got = m.align(dx, dy, guess)                  # no host synchronisation in between
            assert got.status == 0 and got.pose == want.pose and got.n_hit == want.n_hit
            del dx, dy


def test_batch_on_its_own_stream_waits_for_the_producer(gpu_lib):
    import torch
    from gtsam_ndt_amd import dist as nd
    from gtsam_ndt_amd.matcher import NdtBatch2D
    pairs = [synth.make_pair(4, pair_index=k, n_tgt=20000, n_src=20000) for k in range(3)]
    h = nd.pack_pairs(pairs)
    t = {k: torch.from_numpy(v).cuda() for k, v in h.items()}
    torch.cuda.synchronize()
    with NdtBatch2D() as b:
        want = b.decode(b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"]))
        _poison_allocator(t["sx"].numel())
        torch.cuda._sleep(SLEEP_CYCLES)
        sx2 = t["sx"] + 0.0                                   # produced late on torch's current stream
        sy2 = t["sy"] + 0.0
        out = b.align_dev(t["tx"], t["ty"], t["toff"], sx2, sy2, t["soff"], t["init"])      # stream=None: own stream
        torch.cuda.ExternalStream(b.stream).synchronize()
        got = b.decode(out)
    for g, w in zip(got, want):
        assert g.status == 0 and g.pose == w.pose


def test_3d_align_waits_for_the_producer(gpu_lib):
    import torch
    from gtsam_ndt_amd import synth3d
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    d = synth3d.make_pair3d(n_azim=512)
    with NdtMatcher3D(fixed_iterations=5) as m:
        m.set_target(d["tx"], d["ty"], d["tz"])
        s = [torch.from_numpy(d[k]).cuda() for k in ("sx", "sy", "sz")]
        torch.cuda.synchronize()
        want = m.align(*s, d["init"])
        _poison_allocator(s[0].numel())
        torch.cuda._sleep(SLEEP_CYCLES)
        late = [v + 0.0 for v in s]
        got = m.align(*late, d["init"])
    assert got.pose == want.pose


def test_evaluate_takes_device_arrays(gpu_lib):
    """ndt2d_evaluate_dev / ndt3d_evaluate_dev: the evaluation of a scan that is already on the device equals the one of
    its host copy bit for bit (same kernels, no upload)."""
    import torch
    from gtsam_ndt_amd import synth, synth3d
    from gtsam_ndt_amd.matcher import NdtMatcher2D, NdtMatcher3D
    d = synth.make_pair(2, n_tgt=30000, n_src=20000)
    with NdtMatcher2D() as m:
        m.set_target(d["tx"], d["ty"])
        a = m.evaluate(d["sx"], d["sy"], d["pose"])
        b = m.evaluate(torch.from_numpy(d["sx"]).cuda(), torch.from_numpy(d["sy"]).cuda(), d["pose"])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
    d3 = synth3d.make_pair3d(n_elev=16, n_azim=256)
    with NdtMatcher3D() as m:
        m.set_target(d3["tx"], d3["ty"], d3["tz"])
        a = m.evaluate(d3["sx"], d3["sy"], d3["sz"], d3["pose"])
        b = m.evaluate(*(torch.from_numpy(d3[k]).cuda() for k in ("sx", "sy", "sz")), d3["pose"])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
