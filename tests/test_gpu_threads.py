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
