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
