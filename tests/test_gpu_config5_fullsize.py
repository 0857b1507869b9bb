"""BASELINE config 5 at full size (64 x 2048 = 131 072-point scans) in CONVERGED mode: north_star's criterion is the
converged pose within 1e-4 m / 1e-4 rad of the CPU reference, asserted here for each of the three 3D drivers -
k_iterate3 (single pair), k_batch3 (loop-closure batch) and the multi-scan chain (k_multi_solve3 / k_multi_body3) -
against the float64 oracle (oracle/ndt3d.py through its C twin orc3d_*, which tests/test_oracle_c.py pins to the numpy
form at 1e-9; parity unpinned: the reference holds no code).  VERDICT r2 weak #2.

The second test settles VERDICT r2 weak #3: the bench's 3D batch leg reports a few pairs that end more than 1 cm from
their generating pose after the 30 fixed iterations.  Those same scenes (deterministic seeds) are aligned by the oracle
with the same 30 fixed iterations and in converged mode: the oracle is as far from truth after 30 iterations as the kernel
(both still moving: the two trajectories agree to 1e-3 in flight), and where the iteration ENDS the kernel is within 1e-4
of the oracle - the strays are where this score's iteration goes from those starts (a slow basin), not a kernel defect."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth3d

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cport():
    from gtsam_ndt_amd import build
    build.build_oracle()
    from oracle import cport as c
    return c


def _close(pose, ref, tol=1e-4):
    e = np.abs(np.array(pose) - np.array(ref))
    return e[:3].max() < tol and e[3:].max() < tol, e


def test_converged_pose_of_the_full_config5_pair_through_every_3d_driver(gpu_lib, cport):
    import torch
    from gtsam_ndt_amd.matcher import NdtBatch3D, NdtMatcher3D
    from oracle import ndt3d as o3
    d = synth3d.make_pair3d()                                    # the config-5 pair: 131 072 + 131 072 points
    assert len(d["sx"]) == 131072
    prm = o3.Ndt3Params()
    cg = cport.CGrid3(d["tx"], d["ty"], d["tz"], prm)
    ref = cg.align(d["sx"], d["sy"], d["sz"], d["init"], threads=8)
    assert ref["status"] == 0
    # a second scan for the multi-scan chain and the batch: same scene, another pose and noise
    d2 = synth3d.make_pair3d(pose=(-0.2, 0.25, 0.03, -0.008, 0.012, -0.025))
    cg2 = cport.CGrid3(d2["tx"], d2["ty"], d2["tz"], prm)
    ref2 = cg2.align(d2["sx"], d2["sy"], d2["sz"], d2["init"], threads=8)
    ref2_on_1 = cg.align(d2["sx"], d2["sy"], d2["sz"], d2["init"], threads=8)       # scan 2 against target 1 (multi-scan)
    assert ref2["status"] == 0 and ref2_on_1["status"] == 0

    # k_iterate3, converged mode
    with NdtMatcher3D() as m:
        m.set_target(d["tx"], d["ty"], d["tz"])
        r = m.align(d["sx"], d["sy"], d["sz"], d["init"])
        ok, e = _close(r.pose, ref["pose"])
        assert r.status == 0 and ok, e
        assert abs(r.iterations - ref["iterations"]) <= 3
        # multi-scan chain: both scans against the one cached voxel grid
        cu = lambda a: torch.from_numpy(a).cuda()
        scans = [tuple(cu(d[k]) for k in ("sx", "sy", "sz")), tuple(cu(d2[k]) for k in ("sx", "sy", "sz"))]
        rs = m.align_multi_scan(scans, [d["init"], d2["init"]])
        for got, want in zip(rs, (ref, ref2_on_1)):
            ok, e = _close(got.pose, want["pose"])
            assert got.status == 0 and ok, e
            assert abs(got.iterations - want["iterations"]) <= 3
        assert rs[0].pose == r.pose                                  # the chain is bit-identical to the single call

    # k_batch3, converged mode: both pairs in one call through the device entry point
    with NdtBatch3D() as b:
        res = b.align([(d["tx"], d["ty"], d["tz"]), (d2["tx"], d2["ty"], d2["tz"])],
                      [(d["sx"], d["sy"], d["sz"]), (d2["sx"], d2["sy"], d2["sz"])], [d["init"], d2["init"]])
    for got, want in zip(res, (ref, ref2)):
        ok, e = _close(got.pose, want["pose"])
        assert got.status == 0 and ok, e
        assert abs(got.iterations - want["iterations"]) <= 3
    cg.close(); cg2.close()


def test_the_bench_legs_stray_scenes_are_the_scores_basin_not_a_kernel_defect(gpu_lib, cport):
    """bench.py's batch_3d leg: 256 distinct scenes (clutter seed 5 + k, noise seeds 1000 + 2k / 1001 + 2k, relative pose
    from default_rng(5)), fixed 30 iterations.  Find the pairs that end > 1 cm from truth, align exactly those (and a few
    that do not stray) with the oracle, under the same 30 fixed iterations and to convergence."""
    import torch
    from gtsam_ndt_amd import synth_dev
    from gtsam_ndt_amd.matcher import NdtBatch3D, NdtMatcher3D
    from oracle import ndt3d as o3
    n_pairs, n_elev, n_azim = 256, 64, 2048
    npts = n_elev * n_azim
    rng = np.random.default_rng(5)
    poses = [tuple(np.array(synth3d.T_STAR_3D) * rng.uniform(-0.5, 0.5, 6)) for _ in range(n_pairs)]
    dev = torch.device("cuda:0")
    t = [torch.empty(n_pairs * npts, dtype=torch.float32, device=dev) for _ in range(3)]
    s = [torch.empty(n_pairs * npts, dtype=torch.float32, device=dev) for _ in range(3)]
    for k, p in enumerate(poses):
        sl = slice(k * npts, (k + 1) * npts)
        synth_dev.lidar_scan3d(1000 + 2 * k, (0.0,) * 6, n_elev, n_azim, 0.02, scene_seed=5 + k, out=tuple(c[sl] for c in t), firing_order=True)
        synth_dev.lidar_scan3d(1001 + 2 * k, p, n_elev, n_azim, 0.02, scene_seed=5 + k, out=tuple(c[sl] for c in s), firing_order=True)
    off = torch.arange(n_pairs + 1, dtype=torch.int64, device=dev) * npts
    init = torch.zeros((n_pairs, 6), dtype=torch.float64, device=dev)
    with NdtBatch3D(fixed_iterations=30) as b:
        res = b.decode(b.align_dev(t, off, s, off, init))
    assert all(r.status == 0 and r.iterations == 30 for r in res)
    errs = np.array([np.abs(np.array(res[k].pose) - np.array(poses[k])).max() for k in range(n_pairs)])
    strays = [int(k) for k in np.nonzero(errs > 1e-2)[0]]
    sample = strays + [k for k in (0, 64, 128, 192) if k not in strays]
    prm = o3.Ndt3Params(fixed_iterations=30)
    worst_fixed = worst_conv = 0.0
    # converged mode on the same scenes: where each pair's iteration ENDS is the contract (1e-4), and the strays end
    # where the oracle's ends
    with NdtBatch3D() as b:
        conv = b.decode(b.align_dev(t, off, s, off, init))
    report = []
    with NdtMatcher3D(fixed_iterations=30) as m:
        for k in sample:
            sl = slice(k * npts, (k + 1) * npts)
            th = [c[sl].cpu().numpy() for c in t]
            sh = [c[sl].cpu().numpy() for c in s]
            cg = cport.CGrid3(*th, prm)
            ref = cg.align(*sh, (0.0,) * 6, threads=8)
            ref_conv = cg.align(*sh, (0.0,) * 6, threads=8, fixed_iterations=0)
            cg.close()
            assert ref["iterations"] == 30 and ref_conv["status"] == 0
            m.set_target(*th)
            r1 = m.align(*(c[sl].contiguous() for c in s), (0.0,) * 6)
            e = np.abs(np.array(res[k].pose) - np.array(ref["pose"]))
            e1 = np.abs(np.array(r1.pose) - np.array(ref["pose"]))
            ec = np.abs(np.array(conv[k].pose) - np.array(ref_conv["pose"]))
            report.append((k, float(errs[k]), float(e.max()), float(e1.max()), float(ec.max()), conv[k].iterations, ref_conv["iterations"]))
            # fixed 30 iterations: a pair that is not a stray has arrived (1e-4); a stray is still moving centimetres per
            # ten iterations, and two float32 / float64 trajectories in flight are compared at 1e-3
            tol = 1e-3 if k in strays else 1e-4
            assert e.max() < tol and e1.max() < tol, (k, e, e1, errs[k])
            assert conv[k].status == 0 and ec[:3].max() < 1e-4 and ec[3:].max() < 1e-4, (k, ec)     # where the iteration ends
            assert abs(conv[k].iterations - ref_conv["iterations"]) <= max(3, ref_conv["iterations"] // 10), (k, conv[k].iterations, ref_conv["iterations"])
            worst_fixed = max(worst_fixed, float(e.max()), float(e1.max()))
            worst_conv = max(worst_conv, float(ec.max()))
            if k in strays:                                        # the oracle is as far from truth after 30 iterations: the basin
                assert np.abs(np.array(ref["pose"]) - np.array(poses[k])).max() > 0.5e-2
    for row in report:
        print("pair %3d: err vs truth after 30 its %.4f m | batch vs oracle (30 its) %.2e | single vs oracle (30 its) %.2e | "
              "converged batch vs converged oracle %.2e (%d / %d iterations)" % row)
    print(f"strays {strays}; kernel vs oracle on {len(sample)} scenes: fixed-30 max {worst_fixed:.2e}, converged max {worst_conv:.2e}")
