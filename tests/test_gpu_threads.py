"""Threading contract of the C ABI: distinct handles may be used concurrently from distinct
threads (one stream each, no global mutable state)."""
import threading

import numpy as np
import pytest

from gtsam_ndt_amd import synth

pytestmark = pytest.mark.gpu


def test_two_handles_two_threads(gpu_lib):
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    pairs = [synth.make_pair(4, pair_index=k, n_tgt=30000, n_src=30000) for k in range(2)]
    serial = []
    for p in pairs:
        with NdtMatcher2D() as m:
            m.set_target(p["tx"], p["ty"])
            serial.append(m.align(p["sx"], p["sy"], p["init"]))
    out = [None, None]
    errs = []

    def work(k):
        try:
            with NdtMatcher2D() as m:
                for _ in range(5):
                    m.set_target(pairs[k]["tx"], pairs[k]["ty"])
                    out[k] = m.align(pairs[k]["sx"], pairs[k]["sy"], pairs[k]["init"])
        except Exception as e:      # surfaced below
            errs.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errs, repr(errs)
    assert all(o is not None for o in out)
    for a, b in zip(out, serial):
        assert a.status == b.status and a.iterations == b.iterations, (a, b)
        assert a.pose == b.pose and np.array_equal(a.H, b.H), (a.pose, b.pose)


def test_eight_busy_threads_give_the_single_thread_results(gpu_lib):
    """Eight handles driven flat out from eight host threads (long scans through the chunked
    launch chain, short scans through the single-workgroup kernel): every alignment must be the
    one a single thread gets.  Round 1 found a wait loop here that trusted hipStreamQuery to say
    "drained" and now and then returned a state a few chunks short of convergence."""
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    pairs = [synth.make_pair(4, pair_index=k, n_tgt=30000, n_src=30000) for k in range(4)]
    pairs += [synth.make_pair(4, pair_index=10 + k, n_tgt=1500, n_src=1500) for k in range(4)]
    serial = []
    for p in pairs:
        with NdtMatcher2D() as m:
            m.set_target(p["tx"], p["ty"])
            serial.append(m.align(p["sx"], p["sy"], p["init"]))
    bad, errs = [], []

    def work(k):
        try:
            p = pairs[k]
            with NdtMatcher2D() as m:
                for it in range(3):
                    m.set_target(p["tx"], p["ty"])
                    r = m.align(p["sx"], p["sy"], p["init"])
                    b = serial[k]
                    if not (r.status == b.status and r.iterations == b.iterations and r.pose == b.pose and np.array_equal(r.H, b.H)):
                        bad.append((k, it, r.iterations, b.iterations, r.pose, b.pose))
        except Exception as e:
            errs.append(e)

    for _ in range(10):
        th = [threading.Thread(target=work, args=(k,)) for k in range(len(pairs))]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=120)
    assert not errs, repr(errs)
    assert not bad, bad[:3]


def test_3d_contexts_on_four_threads(gpu_lib):
    """The same contract for the 3D entry points: four host threads, each with its own NdtMatcher3D (single calls and a
    multi-scan chain) and NdtBatch3D, all at once - every result must be the one a single thread gets."""
    import torch
    from gtsam_ndt_amd import synth3d
    from gtsam_ndt_amd.matcher import NdtBatch3D, NdtMatcher3D
    poses = [(0.15, -0.1, 0.02, 0.005, -0.005, 0.015), (-0.12, 0.08, -0.02, -0.004, 0.006, -0.01),
             (0.05, 0.14, 0.01, 0.0, 0.007, 0.017), (-0.06, -0.11, 0.03, 0.006, 0.0, -0.015)]
    ds = [synth3d.make_pair3d(n_elev=16, n_azim=256 + 64 * k, pose=p) for k, p in enumerate(poses)]
    zero = (0.0,) * 6
    serial = []
    for d in ds:
        with NdtMatcher3D() as m:
            m.set_target(d["tx"], d["ty"], d["tz"])
            serial.append(m.align(d["sx"], d["sy"], d["sz"], zero))
    with NdtBatch3D() as b:
        serial_b = b.align([(d["tx"], d["ty"], d["tz"]) for d in ds], [(d["sx"], d["sy"], d["sz"]) for d in ds], [zero] * 4)
    bad, errs = [], []

    def work(k):
        try:
            d = ds[k]
            s = tuple(torch.from_numpy(d[c]).cuda() for c in ("sx", "sy", "sz"))
            with NdtMatcher3D() as m, NdtBatch3D() as b:
                for rep in range(6):
                    m.set_target(d["tx"], d["ty"], d["tz"])
                    a = m.align(d["sx"], d["sy"], d["sz"], zero)
                    mm = m.align_multi_scan([s, s, s], [zero] * 3)
                    rb = b.align([(x["tx"], x["ty"], x["tz"]) for x in ds], [(x["sx"], x["sy"], x["sz"]) for x in ds], [zero] * 4)
                    if not (a.pose == serial[k].pose and all(q.pose == serial[k].pose for q in mm)
                            and all(p.pose == q.pose for p, q in zip(rb, serial_b))):
                        bad.append((k, rep))
        except Exception as e:      # surfaced below
            errs.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, repr(errs)
    assert not bad, bad


def test_yielding_host_wait_gives_the_same_results(gpu_lib):
    """ndt_set_host_wait(1): waiting host threads give their core away between polls instead of spinning (a SLAM process
    with more matcher threads than cores).  Eight threads on eight handles, converged alignments: every result equals the
    spinning default's, bit for bit."""
    import threading
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    lib = L.load()
    assert lib.ndt_set_host_wait(2) == L.NDT_ERR_INVALID_ARG
    d = synth.make_pair(2, n_tgt=40_000, n_src=20_000)
    with NdtMatcher2D() as m:
        m.set_target(d["tx"], d["ty"])
        want = m.align(d["sx"], d["sy"], d["init"])
    assert want.status == 0
    bad, errs = [], []

    def work(k):
        try:
            with NdtMatcher2D() as m:
                m.set_target(d["tx"], d["ty"])
                for rep in range(40):
                    r = m.align(d["sx"], d["sy"], d["init"])
                    if not (r.pose == want.pose and r.iterations == want.iterations and r.status == 0):
                        bad.append((k, rep))
        except Exception as e:
            errs.append(e)

    assert lib.ndt_set_host_wait(1) == 0
    try:
        th = [threading.Thread(target=work, args=(k,)) for k in range(8)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
    finally:
        assert lib.ndt_set_host_wait(0) == 0
    assert not errs, repr(errs)
    assert not bad, bad
