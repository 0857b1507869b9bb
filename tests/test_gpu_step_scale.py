"""params.step_scale on the device paths against the oracle (single pair, batch, 3D)."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w", [2.0, 3.0])
def test_single_and_batch(gpu_lib, w):
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
    from oracle import ndt2d as o
    d = synth.make_pair(2, n_tgt=40000, n_src=40000)
    prm = o.NdtParams(step_scale=w)
    g = o.build_grid(d["tx"], d["ty"], prm)
    ref = o.align(g, d["sx"], d["sy"], d["init"], prm)
    plain = o.align(g, d["sx"], d["sy"], d["init"], o.NdtParams())
    with NdtMatcher2D(step_scale=w) as m:
        m.set_target(d["tx"], d["ty"])
        r = m.align(d["sx"], d["sy"], d["init"])
    with NdtBatch2D(step_scale=w) as b:
        rb = b.align([(d["tx"], d["ty"])], [(d["sx"], d["sy"])], [d["init"]])[0]
    for got in (r, rb):
        assert got.status == 0 == ref["status"]
        assert np.abs(np.array(got.pose) - np.array(ref["pose"])).max() < 1e-4     # 1e-4 m / 1e-4 rad
        assert abs(got.iterations - ref["iterations"]) <= 3
        assert got.iterations < 0.7 * plain["iterations"]
    with pytest.raises(L.NdtError):
        NdtMatcher2D(step_scale=9.0)
    with pytest.raises(L.NdtError):
        NdtMatcher2D(step_scale=-1.0)
    with NdtMatcher2D(step_scale=0.0) as m:                                         # 0 means 1
        m.set_target(d["tx"], d["ty"])
        r0 = m.align(d["sx"], d["sy"], d["init"])
    assert abs(r0.iterations - plain["iterations"]) <= 3


def test_3d(gpu_lib):
    from gtsam_ndt_amd import synth3d
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    from oracle import ndt3d as o3
    d = synth3d.make_pair3d(16, 256)
    prm = o3.Ndt3Params(step_scale=2.5)
    ref = o3.align3(o3.build_grid3(d["tx"], d["ty"], d["tz"], prm), d["sx"], d["sy"], d["sz"], d["init"], prm)
    with NdtMatcher3D(step_scale=2.5) as m:
        m.set_target(d["tx"], d["ty"], d["tz"])
        r = m.align(d["sx"], d["sy"], d["sz"], d["init"])
    assert r.status == 0 == ref["status"]
    assert np.abs(np.array(r.pose) - np.array(ref["pose"])).max() < 1e-4
    assert abs(r.iterations - ref["iterations"]) <= 3
