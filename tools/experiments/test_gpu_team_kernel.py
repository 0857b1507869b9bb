"""The team kernel (ndt2d_xcd.hpp; opt-in, NDT_TUNE_TEAM_KERNEL = 1): scans between 4097 points and the wide
threshold run the whole Gauss-Newton loop in one launch, 32 workgroups per start synchronising through L2.  Against the
launch-per-iteration path (same per-point code, another summation order), against the oracle, and
with the GPU kept busy by other work so that a team cannot assemble (bounded wait, fallback)."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth

pytestmark = pytest.mark.gpu


def _dev(d):
    import torch
    return torch.from_numpy(d["sx"]).cuda(), torch.from_numpy(d["sy"]).cuda()


@pytest.mark.parametrize("cfg,n_src,kw", [(2, 100_000, {}), (3, 100_000, {}), (2, 5000, {}), (2, 33_000, dict(hessian_mode=1)),
                                          (2, 100_000, dict(fixed_iterations=30)), (2, 60_001, dict(line_search=4)),
                                          (2, 250_000, dict(step_scale=2.5))])
def test_team_kernel_equals_launch_per_iteration_path_and_oracle(gpu_lib, cfg, n_src, kw):
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    from oracle import ndt2d as o
    d = synth.make_pair(cfg, n_tgt=300_000 if cfg == 3 else None, n_src=n_src)
    sx, sy = _dev(d)
    res = {}
    for team in (1, 0):
        with NdtMatcher2D(tuning={"team_kernel": team}, **kw) as m:
            m.set_target(d["tx"], d["ty"])
            a = m.align(sx, sy, d["init"])
            b = m.align(d["sx"], d["sy"], d["init"])               # host arrays
            m.align_async(sx, sy, d["init"]); m.align_async(sx, sy, d["init"])      # back to back, last one fetched
            c = m.finish()
            e = m.evaluate(d["sx"], d["sy"], d["pose"])
            assert a.pose == b.pose == c.pose and np.array_equal(a.H, c.H)          # deterministic, every entry point
            assert m.team_fallbacks == 0
            res[team] = (a, e)
    (a, ea), (b, eb) = res[1], res[0]
    assert a.status == b.status == 0
    if kw.get("hessian_mode", 0) == 0:
        assert abs(a.iterations - b.iterations) <= 1 and abs(a.n_hit - b.n_hit) <= 1
        assert np.abs(np.array(a.pose) - np.array(b.pose)).max() < 3e-6
    assert ea[3] == eb[3] and abs(ea[2] - eb[2]) <= 3e-6 * eb[2] and np.abs(ea[0] - eb[0]).max() <= 3e-6 * np.abs(eb[0]).max()
    prm = o.NdtParams(**kw)
    ref = o.align(o.build_grid(d["tx"], d["ty"], prm), d["sx"], d["sy"], d["init"], prm)
    assert ref["status"] == 0
    if kw.get("hessian_mode", 0) == 0:
        # sparse scans settle a few cell flips apart in float32 and float64 (as on every path): looser there
        assert np.abs(np.array(a.pose) - np.array(ref["pose"])).max() < (1e-4 if n_src >= 30_000 else 5e-4)   # BASELINE.json: 1e-4
        if n_src >= 30_000:
            assert abs(a.iterations - ref["iterations"]) <= 3


def test_edge_statuses_through_the_team_kernel(gpu_lib):
    from gtsam_ndt_amd import _lib as L
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(2, n_tgt=50_000, n_src=20_000)
    with NdtMatcher2D(tuning={"team_kernel": 1}) as m:
        m.set_target(d["tx"], d["ty"])
        r = m.align(d["sx"] + 1000.0, d["sy"], d["init"])
        assert r.status == L.NDT_TOO_FEW_HITS and r.iterations == 0
        sx = d["sx"].copy(); sx[::7] = np.nan
        r = m.align(sx, d["sy"], d["init"])
        assert r.status == 0 and np.isfinite(r.pose).all() and np.isfinite(r.H).all()
    with NdtMatcher2D(max_iterations=5, tuning={"team_kernel": 1}) as m:
        m.set_target(d["tx"], d["ty"])
        r = m.align(d["sx"], d["sy"], d["init"])
        assert r.status == L.NDT_NOT_CONVERGED and r.iterations == 5


def test_team_kernel_gives_way_when_the_gpu_is_busy(gpu_lib):
    """Other work holds the CUs: the teams cannot assemble, the bounded wait ends, the alignment runs on
    the launch-per-iteration path - same answer, a counted fallback, no hang."""
    import torch
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(2)
    sx, sy = _dev(d)
    with NdtMatcher2D(tuning={"team_kernel": 1}) as m:
        m.set_target(d["tx"], d["ty"])
        want = m.align(sx, sy, d["init"])
        side = torch.cuda.Stream()
        big = torch.randn(8192, 8192, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(40):                       # ~0.2 s of a kernel that fills every CU
                big = torch.sin(big) * 1.0001
        got = [m.align(sx, sy, d["init"]) for _ in range(3)]
        side.synchronize()
        after = m.align(sx, sy, d["init"])
    for g in got:
        assert g.status == 0 and np.abs(np.array(g.pose) - np.array(want.pose)).max() < 3e-6
    assert after.pose == want.pose


def test_multi_start_through_teams_is_bitwise_the_single_start_team_result(gpu_lib):
    """With the team kernel on, ndt2d_align_multi_start_dev gives a start to each team; every start is bit for
    bit what the single-start call (one team) returns."""
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    d = synth.make_pair(2)
    sx, sy = _dev(d)
    starts = [(d["init"][0] + 0.01 * k, d["init"][1] - 0.008 * k, 0.0006 * k) for k in range(11)]
    with NdtMatcher2D(tuning={"team_kernel": 1}) as m:
        m.set_target(d["tx"], d["ty"])
        multi = m.align_multi_start(sx, sy, starts)
        for a, s in zip(multi, starts):
            b = m.align(sx, sy, s)
            assert a.pose == b.pose and a.iterations == b.iterations and a.status == b.status and np.array_equal(a.H, b.H)
        assert m.team_fallbacks == 0
