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
    r = o3.align3(g, d["sx"], d["sy"], d["sz"], d["init"], prm)
    k5 = o3.align3(g, d["sx"], d["sy"], d["sz"], d["init"], o3.Ndt3Params(fixed_iterations=5))
    ev = o3.evaluate3(g, d["sx"], d["sy"], d["sz"], d["pose"], prm)
    out = {k: d[k] for k in ("tx", "ty", "tz", "sx", "sy", "sz")}
    out.update({"init": np.array(d["init"]), "true_pose": np.array(d["pose"]),
                "grid_geom": np.array([*map(float, g.o), float(g.inv_c), *g.dims, g.n_valid], dtype=np.float64),
                "grid_count": g.count.astype(np.int32), "grid_mean": g.mean, "grid_icov": g.icov,
                "grid_valid": g.valid, "final_pose": np.array(r["pose"]),
                "final_iterations": np.int32(r["iterations"]), "final_status": np.int32(r["status"]),
                "fixed5_pose": np.array(k5["pose"]), "eval_H": ev[0], "eval_g": ev[1],
                "eval_score": np.float64(ev[2]), "eval_n_hit": np.int32(ev[3])})
    path = os.path.join(ROOT, "tests", "golden", "ndt3d_small.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", r["iterations"], "iterations; pose", r["pose"])


if __name__ == "__main__":
    main()
    main3d()
