#!/usr/bin/env python3
"""Generates tests/golden/ndt2d_config1.npz: inputs and expected outputs of the 2D NDT path
on BASELINE config 1 (two 1k-point scans, 0.5 m cells).

The vectors come from THIS REPO'S float64 oracle (oracle/ndt2d.py), because the reference
checkout contains no implementation, test or fixture to generate them from
(/root/reference/README.md:1 is its only line) - parity unpinned, see DESIGN.md section 3.
They pin the oracle against silent change and give the GPU tests a fixture that does not
depend on importing the oracle.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gtsam_ndt_amd import synth          # noqa: E402
from oracle import ndt2d as o            # noqa: E402


def main():
    d = synth.make_pair(1)
    prm = o.NdtParams()
    g = o.build_grid(d["tx"], d["ty"], prm)
    trace = []
    r = o.align(g, d["sx"], d["sy"], d["init"], prm, trace=trace)
    k5 = o.align(g, d["sx"], d["sy"], d["init"], o.NdtParams(fixed_iterations=5))
    ev_pose = np.array([d["init"], d["pose"], (0.12, -0.07, 0.012)], dtype=np.float64)
    ev = [o.evaluate(g, d["sx"], d["sy"], p, prm) for p in ev_pose]
    evn = [o.evaluate(g, d["sx"], d["sy"], p, o.NdtParams(hessian_mode=o.HESSIAN_NEWTON)) for p in ev_pose]
    out = {
        "tx": d["tx"], "ty": d["ty"], "sx": d["sx"], "sy": d["sy"],
        "init": np.array(d["init"]), "true_pose": np.array(d["pose"]),
        "grid_geom": np.array([float(g.ox), float(g.oy), float(g.inv_c), g.W, g.H, g.n_valid], dtype=np.float64),
        "grid_count": g.count.astype(np.int32), "grid_mean": g.mean, "grid_icov": g.icov,
        "grid_valid": g.valid,
        "trace_pose": np.array([t["pose"] for t in trace]), "trace_H": np.array([t["H"] for t in trace]),
        "trace_g": np.array([t["g"] for t in trace]), "trace_score": np.array([t["score"] for t in trace]),
        "trace_n_hit": np.array([t["n_hit"] for t in trace], dtype=np.int32),
        "final_pose": np.array(r["pose"]), "final_iterations": np.int32(r["iterations"]),
        "final_status": np.int32(r["status"]), "fixed5_pose": np.array(k5["pose"]),
        "eval_pose": ev_pose,
        "eval_H": np.array([e[0] for e in ev]), "eval_g": np.array([e[1] for e in ev]),
        "eval_score": np.array([e[2] for e in ev]), "eval_n_hit": np.array([e[3] for e in ev], dtype=np.int32),
        "eval_H_newton": np.array([e[0] for e in evn]),
    }
    path = os.path.join(ROOT, "tests", "golden", "ndt2d_config1.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", r["iterations"], "iterations; pose", r["pose"])


def main3d():
    """tests/golden/ndt3d_small.npz: a 16 x 256-beam (4096-point) pair of the config-5 scene."""
    from gtsam_ndt_amd import synth3d
    from oracle import ndt3d as o3
    d = synth3d.make_pair3d(n_elev=16, n_azim=256)
    prm = o3.Ndt3Params()
    g = o3.build_grid3(d["tx"], d["ty"], d["tz"], prm)
    trace = []
    r = o3.align3(g, d["sx"], d["sy"], d["sz"], d["init"], prm, trace=trace)
    k5 = o3.align3(g, d["sx"], d["sy"], d["sz"], d["init"], o3.Ndt3Params(fixed_iterations=5))
    ev = o3.evaluate3(g, d["sx"], d["sy"], d["sz"], d["pose"], prm)
    out = {k: d[k] for k in ("tx", "ty", "tz", "sx", "sy", "sz")}
    out.update({"init": np.array(d["init"]), "true_pose": np.array(d["pose"]),
                "grid_geom": np.array([*map(float, g.o), float(g.inv_c), *g.dims, g.n_valid], dtype=np.float64),
                "grid_count": g.count.astype(np.int32), "grid_mean": g.mean, "grid_icov": g.icov,
                "grid_valid": g.valid, "final_pose": np.array(r["pose"]),
                "final_iterations": np.int32(r["iterations"]), "final_status": np.int32(r["status"]),
                "fixed5_pose": np.array(k5["pose"]), "eval_H": ev[0], "eval_g": ev[1],
                "eval_score": np.float64(ev[2]), "eval_n_hit": np.int32(ev[3]),
                # per-iteration trace: entry j = the evaluation at the pose before update j + 1
                "trace_pose": np.array([t["pose"] for t in trace]), "trace_H": np.array([t["H"] for t in trace]),
                "trace_g": np.array([t["g"] for t in trace]), "trace_score": np.array([t["score"] for t in trace]),
                "trace_n_hit": np.array([t["n_hit"] for t in trace], dtype=np.int32)})
    path = os.path.join(ROOT, "tests", "golden", "ndt3d_small.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", r["iterations"], "iterations; pose", r["pose"])


OPTION_CASES = {
    # name: (NdtParams keyword arguments, offset added to the config-1 initial guess)
    "step_scale_3": (dict(step_scale=3.0), (0.0, 0.0, 0.0)),
    "line_search_4": (dict(line_search=4), (0.05, -0.04, 0.01)),
    "overlap_4": (dict(overlap=4), (0.0, 0.0, 0.0)),
    "cell_1m_eig_0p03": (dict(cell_size=1.0, eig_ratio=0.03), (0.1, 0.1, -0.02)),
    "magnusson_0p3": (dict(), (0.0, 0.0, 0.0)),          # d1, d2 filled in from the mixture constants below
    "fixed_12_scale_2": (dict(fixed_iterations=12, step_scale=2.0), (0.0, 0.0, 0.0)),
}


def magnusson(outlier_ratio: float, cell: float, dim: int):
    """Magnusson 2009 eq. 6.8-6.10 as include/ndt_hip.h states them (d1 returned positive)."""
    import math
    c1 = 10.0 * (1.0 - outlier_ratio)
    c2 = outlier_ratio / cell ** dim
    d3 = -math.log(c2)
    d1 = -math.log(c1 + c2) - d3
    d2 = -2.0 * math.log((-math.log(c1 * math.exp(-0.5) + c2) - d3) / d1)
    return -d1, d2


def main_options():
    """tests/golden/ndt2d_options.npz: what the optional parts of the contract (over-relaxation,
    line search, overlapping grids, other cell sizes, mixture score constants) return on the
    config-1 pair.  Inputs are those of ndt2d_config1.npz; only results are stored."""
    d = synth.make_pair(1)
    out = {}
    for name, (kw, off) in OPTION_CASES.items():
        kw = dict(kw)
        if name.startswith("magnusson"):
            kw["d1"], kw["d2"] = magnusson(0.3, 0.5, 2)
            out[name + "_d1d2"] = np.array([kw["d1"], kw["d2"]])
        prm = o.NdtParams(**kw)
        grid = o.build_grids(d["tx"], d["ty"], prm) if prm.overlap == 4 else o.build_grid(d["tx"], d["ty"], prm)
        init = tuple(a + b for a, b in zip(d["init"], off))
        r = o.align(grid, d["sx"], d["sy"], init, prm)
        out[name + "_pose"] = np.array(r["pose"])
        out[name + "_meta"] = np.array([r["iterations"], r["status"], r["n_hit"]], dtype=np.int64)
        out[name + "_score"] = np.float64(r["score"])
        print(name, r["iterations"], r["status"], r["pose"])
    path = os.path.join(ROOT, "tests", "golden", "ndt2d_options.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
    main3d()
    main_options()
