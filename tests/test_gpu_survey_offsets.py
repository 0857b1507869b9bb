"""The offsets SURVEY.md section 8d proposed, at full size, through the coarse-to-fine paths.

The workloads of synth.py start 0.10 m / 0.01 rad (configs 1-3) or up to 0.1 m / 0.01 rad (config 4) from the
generating pose, because the survey's T* = (0.30 m, -0.20 m, 0.05 rad) and its +-0.4 m / +-0.08 rad loop-closure
offsets lie outside the basin of a single 0.5 m grid.  They are inside the basin of the
library's 3-level schedule (2 m -> 1 m -> 0.5 m cells): this file runs them at BASELINE.json's sizes (100k / 100k,
1M / 100k, 100k-point candidate pairs) and pins the result on the oracle's composition of the same levels.
Parity is against this repo's oracle (reference implementation unavailable, /root/reference/README.md:1)."""
import numpy as np
import pytest

from gtsam_ndt_amd import synth

pytestmark = pytest.mark.gpu

T_SURVEY = np.array([0.30, -0.20, 0.05])


def _oracle_pyramid(p, init, levels):
    from oracle import ndt2d as o
    pose, total, r = tuple(init), 0, None
    for lv in levels:
        prm = o.NdtParams(cell_size=lv.cell_size, min_points=lv.min_points, eig_ratio=lv.eig_ratio, d1=lv.d1, d2=lv.d2,
                          hessian_mode=lv.hessian_mode, max_iterations=lv.max_iterations,
                          fixed_iterations=lv.fixed_iterations, eps_trans=lv.eps_trans, eps_rot=lv.eps_rot,
                          step_max_trans=lv.step_max_trans, step_max_rot=lv.step_max_rot, min_hits=lv.min_hits)
        r = o.align(o.build_grid(p["tx"], p["ty"], prm), p["sx"], p["sy"], pose, prm)
        total += r["iterations"]
        if r["status"] not in (o.NDT_OK, o.NDT_NOT_CONVERGED):
            break
        pose = r["pose"]
    r["iterations"] = total
    return r


@pytest.mark.parametrize("config", [2, 3])
def test_survey_offset_single_pair_full_size(gpu_lib, config):
    """Configs 2 (100k / 100k) and 3 (1M-point submap / 100k-point scan) started T_SURVEY away from the generating
    pose: the single 0.5 m grid does not get there, the pyramid does, on the oracle's trajectory."""
    from gtsam_ndt_amd import matcher as M
    d = synth.make_pair(config)
    init = tuple(np.array(d["pose"]) - T_SURVEY)
    with M.NdtPyramid2D() as p:
        p.set_target(d["tx"], d["ty"])
        r = p.align(d["sx"], d["sy"], init)
    with M.NdtMatcher2D() as m:
        m.set_target(d["tx"], d["ty"])
        flat = m.align(d["sx"], d["sy"], init)
    e = np.abs(np.array(r.pose) - np.array(d["pose"]))
    assert r.status == 0 and e[:2].max() < 5e-3 and e[2] < 5e-4, (r.pose, d["pose"])     # sampling noise of the scans
    assert np.abs(np.array(flat.pose) - np.array(d["pose"]))[:2].max() > 5e-3             # the single level stalls elsewhere (config 2: 1.9 cm off)
    ref = _oracle_pyramid(d, init, M.pyramid_params())
    assert ref["status"] == 0
    assert np.abs(np.array(r.pose) - np.array(ref["pose"])).max() < 1e-4                  # BASELINE.json: 1e-4 m / 1e-4 rad
    assert abs(r.iterations - ref["iterations"]) <= 6


def test_survey_loop_closure_offsets_full_size_batch(gpu_lib):
    """Config-4 candidates (100k / 100k points) with the survey's offsets, U(+-0.4 m, +-0.4 m, +-0.08 rad), through the
    coarse-to-fine batch on the device entry point (one k_batch launch per level)."""
    import torch
    from gtsam_ndt_amd import dist as nd
    from gtsam_ndt_amd import matcher as M
    idx = (0, 137, 2048, 4095)
    pairs = [synth.make_pair(4, pair_index=k) for k in idx]
    for k, p in zip(idx, pairs):
        u = synth.uniform01(7000 + k, np.arange(3, 6, dtype=np.uint64))
        off = (u - 0.5) * np.array([0.8, 0.8, 0.16])
        p["init"] = tuple(np.array(p["pose"]) - off)
    levels = M.pyramid_params()
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(v).to(dev) for k, v in nd.pack_pairs(pairs).items()}
    with M.NdtBatch2D(levels=levels) as b:
        rows = b.decode(b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"]))
    n_near = 0
    for p, r in zip(pairs, rows):
        ref = _oracle_pyramid(p, p["init"], levels)
        assert r.status == ref["status"]
        e = np.abs(np.array(r.pose) - np.array(p["pose"]))
        if e[:2].max() < 5e-3 and e[2] < 5e-4:          # reached the generating pose: then on the oracle's trajectory
            n_near += 1
            assert np.abs(np.array(r.pose) - np.array(ref["pose"])).max() < 1e-4, (r.pose, ref["pose"])
            assert abs(r.iterations - ref["iterations"]) <= 6
        else:                                           # a candidate the schedule does not recover: the oracle agrees
            assert np.abs(np.array(ref["pose"]) - np.array(p["pose"]))[:2].max() > 5e-3
    assert n_near >= 3, [r.pose for r in rows]
