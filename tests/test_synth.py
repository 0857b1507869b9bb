"""The seeded scene generator: reproducible, documented shapes, sane geometry."""
import numpy as np

from gtsam_ndt_amd import synth


def test_splitmix_known_answers():
    # splitmix64 reference values for seed 0 (first outputs of the canonical generator)
    z = synth.splitmix64(0, np.arange(3, dtype=np.uint64))
    assert [int(v) for v in z] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]
    u = synth.uniform01(7, np.arange(1000, dtype=np.uint64))
    assert (u >= 0).all() and (u < 1).all() and 0.45 < u.mean() < 0.55


def test_pairs_are_reproducible_and_shaped():
    for cfg, nt, ns in ((1, 1000, 1000), (2, 100_000, 100_000)):
        a, b = synth.make_pair(cfg), synth.make_pair(cfg)
        for k in ("tx", "ty", "sx", "sy"):
            assert a[k].dtype == np.float32 and np.array_equal(a[k], b[k])
        assert len(a["tx"]) == nt and len(a["sx"]) == ns
    p0, p1 = synth.make_pair(4, pair_index=0, n_tgt=2000, n_src=2000), synth.make_pair(4, pair_index=1, n_tgt=2000, n_src=2000)
    assert not np.array_equal(p0["tx"], p1["tx"]) and p0["pose"] != p1["pose"]
    assert max(abs(v) for v in p0["pose"][:2]) <= 0.1 and abs(p0["pose"][2]) <= 0.01


def test_source_maps_onto_target_surfaces():
    """Applying the generating pose to the source puts it back on the target's surfaces."""
    d = synth.make_pair(2, n_tgt=20000, n_src=2000)
    tx, ty, th = d["pose"]
    c, s = np.cos(th), np.sin(th)
    wx = c * d["sx"] - s * d["sy"] + tx
    wy = s * d["sx"] + c * d["sy"] + ty
    from scipy.spatial import cKDTree
    dist, _ = cKDTree(np.c_[d["tx"], d["ty"]]).query(np.c_[wx, wy])
    assert np.median(dist) < 0.1
    assert np.abs(d["tx"]).max() <= 25.5 and np.abs(d["ty"]).max() <= 25.5


def test_config3_scan_sits_inside_the_submap():
    d = synth.make_pair(3, n_tgt=50000, n_src=5000)
    assert d["init"] == (25.0, -25.0, 0.0)
    assert np.abs(d["tx"]).max() <= 100.5
    # the scan is expressed in the sensor frame: centred near the origin, 50 m across
    assert np.abs(d["sx"]).max() < 26.0 and np.abs(d["sy"]).max() < 26.0
