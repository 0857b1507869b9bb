"""Committed golden vectors (tests/golden/ndt2d_config1.npz, made by tests/golden/make_golden.py
from this repo's oracle - the reference holds no fixtures; parity unpinned).

CPU: the oracle and the generator still reproduce them (guards against silent drift).
GPU: the HIP path matches them without importing the oracle."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ndt2d_config1.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)            # allow_pickle stays False


def test_generator_reproduces_golden_inputs(gold):
    from gtsam_ndt_amd import synth
    d = synth.make_pair(1)
    for k in ("tx", "ty", "sx", "sy"):
        np.testing.assert_array_equal(d[k], gold[k])
    assert tuple(gold["init"]) == d["init"] and tuple(gold["true_pose"]) == d["pose"]


def test_oracle_reproduces_golden_outputs(gold):
    from oracle import ndt2d as o
    prm = o.NdtParams()
    g = o.build_grid(gold["tx"], gold["ty"], prm)
    assert [float(g.ox), float(g.oy), float(g.inv_c), g.W, g.H, g.n_valid] == list(gold["grid_geom"])
    np.testing.assert_array_equal(g.count, gold["grid_count"])
    np.testing.assert_array_equal(g.valid, gold["grid_valid"])
    np.testing.assert_allclose(g.mean, gold["grid_mean"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(g.icov, gold["grid_icov"], rtol=1e-12)
    trace = []
    r = o.align(g, gold["sx"], gold["sy"], tuple(gold["init"]), prm, trace=trace)
    assert r["iterations"] == int(gold["final_iterations"]) and r["status"] == int(gold["final_status"])
    np.testing.assert_allclose(r["pose"], gold["final_pose"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.array([t["pose"] for t in trace]), gold["trace_pose"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(np.array([t["H"] for t in trace]), gold["trace_H"], rtol=1e-9)
    np.testing.assert_array_equal(np.array([t["n_hit"] for t in trace]), gold["trace_n_hit"])
    for p, H, Hn in zip(gold["eval_pose"], gold["eval_H"], gold["eval_H_newton"]):
        np.testing.assert_allclose(o.evaluate(g, gold["sx"], gold["sy"], p, prm)[0], H, rtol=1e-11)
        np.testing.assert_allclose(
            o.evaluate(g, gold["sx"], gold["sy"], p, o.NdtParams(hessian_mode=1))[0], Hn, rtol=1e-9,
            atol=1e-9 * np.abs(Hn).max())


@pytest.mark.gpu
def test_hip_path_matches_golden(gold, gpu_lib):
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    with NdtMatcher2D() as m:
        info = m.set_target(gold["tx"], gold["ty"])
        assert [info.ox, info.oy, info.inv_cell, info.width, info.height, info.n_valid] == list(gold["grid_geom"])
        count, mean, icov = m.grid()
        np.testing.assert_array_equal(count, gold["grid_count"])          # bit-exact integer work
        v = gold["grid_valid"]
        np.testing.assert_array_equal(icov[:, 0] != 0, v)
        np.testing.assert_allclose(mean[v], gold["grid_mean"][v], rtol=0, atol=2e-6)
        nrm = np.linalg.norm(gold["grid_icov"][v], axis=1, keepdims=True)
        assert np.max(np.abs(icov[v] - gold["grid_icov"][v]) / nrm) < 1e-5
        for p, H, g, s, nh in zip(gold["eval_pose"], gold["eval_H"], gold["eval_g"], gold["eval_score"],
                                  gold["eval_n_hit"]):
            Hd, gd, sd, nd = m.evaluate(gold["sx"], gold["sy"], p)
            assert abs(nd - nh) <= 2
            assert np.abs(Hd - H).max() / np.abs(H).max() < 5e-3       # a boundary point may change cell
            assert abs(sd - s) / s < 5e-3
        r = m.align(gold["sx"], gold["sy"], tuple(gold["init"]))
        assert r.status == int(gold["final_status"])
        e = np.abs(np.array(r.pose) - gold["final_pose"])
        assert e[0] < 1e-4 and e[1] < 1e-4 and e[2] < 1e-4          # BASELINE.json: 1e-4 m / 1e-4 rad
    with NdtMatcher2D(fixed_iterations=5) as m:
        m.set_target(gold["tx"], gold["ty"])
        r = m.align(gold["sx"], gold["sy"], tuple(gold["init"]))
        assert r.iterations == 5 and np.abs(np.array(r.pose) - gold["fixed5_pose"]).max() < 1e-4


GOLD3 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ndt3d_small.npz")


def test_oracle3d_reproduces_golden():
    from oracle import ndt3d as o3
    g3 = np.load(GOLD3)
    prm = o3.Ndt3Params()
    g = o3.build_grid3(g3["tx"], g3["ty"], g3["tz"], prm)
    assert [*map(float, g.o), float(g.inv_c), *g.dims, g.n_valid] == list(g3["grid_geom"])
    np.testing.assert_array_equal(g.count, g3["grid_count"])
    np.testing.assert_array_equal(g.valid, g3["grid_valid"])
    np.testing.assert_allclose(g.icov, g3["grid_icov"], rtol=1e-9, atol=1e-9 * np.abs(g3["grid_icov"]).max())
    r = o3.align3(g, g3["sx"], g3["sy"], g3["sz"], tuple(g3["init"]), prm)
    assert r["iterations"] == int(g3["final_iterations"]) and r["status"] == int(g3["final_status"])
    np.testing.assert_allclose(r["pose"], g3["final_pose"], rtol=0, atol=1e-10)


@pytest.mark.gpu
def test_hip_3d_matches_golden(gpu_lib):
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    g3 = np.load(GOLD3)
    with NdtMatcher3D() as m:
        info = m.set_target(g3["tx"], g3["ty"], g3["tz"])
        assert [info.ox, info.oy, info.oz, info.inv_cell, info.width, info.height, info.depth, info.n_valid] == list(g3["grid_geom"])
        count, mean, icov = m.grid()
        np.testing.assert_array_equal(count, g3["grid_count"])
        H, g, s, nh = m.evaluate(g3["sx"], g3["sy"], g3["sz"], tuple(g3["true_pose"]))
        assert abs(nh - int(g3["eval_n_hit"])) <= 3 and abs(s - float(g3["eval_score"])) / float(g3["eval_score"]) < 5e-3
        r = m.align(g3["sx"], g3["sy"], g3["sz"], tuple(g3["init"]))
    e = np.abs(np.array(r.pose) - g3["final_pose"])
    assert r.status == int(g3["final_status"]) and e[:3].max() < 1e-4 and e[3:].max() < 1e-4
    with NdtMatcher3D(fixed_iterations=5) as m:
        m.set_target(g3["tx"], g3["ty"], g3["tz"])
        r = m.align(g3["sx"], g3["sy"], g3["sz"], tuple(g3["init"]))
    assert r.iterations == 5 and np.abs(np.array(r.pose) - g3["fixed5_pose"]).max() < 1e-4


GOLD_OPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ndt2d_options.npz")


def _option_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(GOLD), "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.OPTION_CASES, mod.magnusson


def test_oracle_reproduces_option_golden(gold):
    """The optional parts of the contract (over-relaxation, line search, overlapping grids, other
    cell sizes, mixture constants) still give the committed results on the config-1 pair."""
    from oracle import ndt2d as o
    cases, magnusson = _option_cases()
    opt = np.load(GOLD_OPT)
    for name, (kw, off) in cases.items():
        kw = dict(kw)
        if name.startswith("magnusson"):
            kw["d1"], kw["d2"] = magnusson(0.3, 0.5, 2)
            np.testing.assert_allclose(opt[name + "_d1d2"], [kw["d1"], kw["d2"]], rtol=1e-15)
        prm = o.NdtParams(**kw)
        grid = o.build_grids(gold["tx"], gold["ty"], prm) if prm.overlap == 4 else o.build_grid(gold["tx"], gold["ty"], prm)
        init = tuple(a + b for a, b in zip(gold["init"], off))
        r = o.align(grid, gold["sx"], gold["sy"], init, prm)
        assert [r["iterations"], r["status"], r["n_hit"]] == list(opt[name + "_meta"]), name
        np.testing.assert_allclose(np.array(r["pose"]), opt[name + "_pose"], rtol=0, atol=1e-12, err_msg=name)
        assert abs(r["score"] - float(opt[name + "_score"])) <= 1e-9 * float(opt[name + "_score"]), name


@pytest.mark.gpu
def test_hip_path_matches_option_golden(gold, gpu_lib):
    from gtsam_ndt_amd.matcher import NdtMatcher2D, magnusson_constants
    cases, _ = _option_cases()
    opt = np.load(GOLD_OPT)
    for name, (kw, off) in cases.items():
        kw = dict(kw)
        if "overlap" in kw:
            kw["overlap_grids"] = kw.pop("overlap")
        if name.startswith("magnusson"):
            kw["d1"], kw["d2"] = magnusson_constants(0.3, 0.5, 2)          # the library's own constants
            np.testing.assert_allclose([kw["d1"], kw["d2"]], opt[name + "_d1d2"], rtol=1e-12)
        init = tuple(a + b for a, b in zip(gold["init"], off))
        with NdtMatcher2D(**kw) as m:
            m.set_target(gold["tx"], gold["ty"])
            r = m.align(gold["sx"], gold["sy"], init)
        its, status, n_hit = (int(v) for v in opt[name + "_meta"])
        assert r.status == status, name
        e = np.abs(np.array(r.pose) - opt[name + "_pose"])
        assert e.max() < 1e-4, (name, r.pose, opt[name + "_pose"])              # 1e-4 m / 1e-4 rad
        assert abs(r.iterations - its) <= 3, (name, r.iterations, its)


@pytest.mark.gpu
def test_hip_per_iteration_trace_follows_the_golden_trace(gold, gpu_lib):
    """ndt2d_align_trace against the committed oracle trace of the config-1 pair, iteration by iteration:
    hits, score, H and g of every evaluation, and the pose after every update."""
    from gtsam_ndt_amd.matcher import NdtMatcher2D
    with NdtMatcher2D() as m:
        m.set_target(gold["tx"], gold["ty"])
        rows = m.align_trace(gold["sx"], gold["sy"], tuple(gold["init"]))
        final = m.align(gold["sx"], gold["sy"], tuple(gold["init"]))
    tp, tH, tg, ts, tn = gold["trace_pose"], gold["trace_H"], gold["trace_g"], gold["trace_score"], gold["trace_n_hit"]
    assert abs(len(rows) - len(tp)) <= 3 and rows[-1].status == int(gold["final_status"])
    # the device path (k_iterate) and the one-workgroup kernel the default call uses end at the same pose
    assert np.abs(np.array(rows[-1].pose) - np.array(final.pose)).max() < 2e-6
    k = min(len(rows), len(tp)) - 1
    for j in range(k):
        r = rows[j]
        assert r.iterations == j + 1
        assert abs(r.n_hit - int(tn[j])) <= 2, j                       # a boundary point may change cell in float32
        assert abs(r.score - ts[j]) / ts[j] < 5e-3, j
        assert np.abs(r.H - tH[j]).max() / np.abs(tH[j]).max() < 5e-3, j
        gs = np.sqrt(np.abs(np.diag(tH[j])) * max(ts[j], 1.0))
        assert np.max(np.abs(r.g - tg[j]) / gs) < 5e-3, j
        assert np.abs(np.array(r.pose) - tp[j + 1]).max() < 1e-4, j    # pose after update j+1 = pose of evaluation j+1


def test_oracle3d_reproduces_golden_trace():
    from oracle import ndt3d as o3
    g3 = np.load(GOLD3)
    prm = o3.Ndt3Params()
    trace = []
    o3.align3(o3.build_grid3(g3["tx"], g3["ty"], g3["tz"], prm), g3["sx"], g3["sy"], g3["sz"], tuple(g3["init"]), prm, trace=trace)
    assert len(trace) == len(g3["trace_pose"])
    np.testing.assert_allclose(np.array([t["pose"] for t in trace]), g3["trace_pose"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(np.array([t["H"] for t in trace]), g3["trace_H"], rtol=1e-9, atol=1e-9 * np.abs(g3["trace_H"]).max())
    np.testing.assert_array_equal(np.array([t["n_hit"] for t in trace]), g3["trace_n_hit"])


@pytest.mark.gpu
def test_hip_3d_per_iteration_trace_follows_the_golden_trace(gpu_lib):
    """ndt3d_align_trace against the committed oracle trace of the 4096-point pair, iteration by iteration: hits,
    score, the 6 x 6 H and g of every evaluation, and the pose after every update."""
    from gtsam_ndt_amd.matcher import NdtMatcher3D
    g3 = np.load(GOLD3)
    with NdtMatcher3D() as m:
        m.set_target(g3["tx"], g3["ty"], g3["tz"])
        rows = m.align_trace(g3["sx"], g3["sy"], g3["sz"], tuple(g3["init"]))
        final = m.align(g3["sx"], g3["sy"], g3["sz"], tuple(g3["init"]))
    tp, tH, tg, ts, tn = g3["trace_pose"], g3["trace_H"], g3["trace_g"], g3["trace_score"], g3["trace_n_hit"]
    assert abs(len(rows) - len(tp)) <= 3 and rows[-1].status == int(g3["final_status"])
    assert rows[-1].pose == final.pose and rows[-1].iterations == final.iterations      # the same kernels, launched one by one
    k = min(len(rows), len(tp)) - 1
    for j in range(k):
        r = rows[j]
        assert r.iterations == j + 1
        assert abs(r.n_hit - int(tn[j])) <= 3, j                       # a boundary point may change voxel in float32
        assert abs(r.score - ts[j]) / ts[j] < 5e-3, j
        sc = np.sqrt(np.outer(np.abs(np.diag(tH[j])), np.abs(np.diag(tH[j]))))
        assert np.max(np.abs(r.H - tH[j]) / sc) < 5e-3, j
        gs = np.sqrt(np.abs(np.diag(tH[j])) * max(ts[j], 1.0))
        assert np.max(np.abs(r.g - tg[j]) / gs) < 5e-3, j
        assert np.abs(np.array(r.pose) - tp[j + 1]).max() < 1e-4, j    # pose after update j+1 = pose of evaluation j+1


@pytest.mark.gpu
def test_hip_3d_batch_and_multi_scan_match_golden(gpu_lib):
    """The committed 3D vectors through the other two 3D drivers: the on-chip batch kernel (k_batch3) and the multi-scan
    launch chain must land on the golden final pose as well (1e-4 m / rad), the fixed-5 pose included."""
    import torch
    from gtsam_ndt_amd.matcher import NdtBatch3D, NdtMatcher3D
    g3 = np.load(GOLD3)
    T, S = (g3["tx"], g3["ty"], g3["tz"]), (g3["sx"], g3["sy"], g3["sz"])
    init = tuple(g3["init"])
    with NdtBatch3D() as b:
        rb = b.align([T, T], [S, S], [init, init])
    with NdtBatch3D(fixed_iterations=5) as b:
        r5 = b.align([T], [S], [init])[0]
    with NdtMatcher3D() as m:
        m.set_target(*T)
        s = tuple(torch.from_numpy(np.ascontiguousarray(c)).cuda() for c in S)
        rm = m.align_multi_scan([s, s, s], [init] * 3)
    for r in (*rb, *rm):
        e = np.abs(np.array(r.pose) - g3["final_pose"])
        assert r.status == int(g3["final_status"]) and e.max() < 1e-4, (r.pose, g3["final_pose"])
        assert abs(r.iterations - int(g3["final_iterations"])) <= 3
    assert r5.iterations == 5 and np.abs(np.array(r5.pose) - g3["fixed5_pose"]).max() < 1e-4
