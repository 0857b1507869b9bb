"""Scratch: does the order of the source scan matter to k_iterate (records gathered through L2)?  Config 3, fixed 30
iterations, the scan in the generator's random order, sorted by bearing around its centroid, and sorted by 0.5 m cell."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtMatcher2D
d = synth.make_pair(3)
tx, ty = torch.from_numpy(d["tx"]).cuda(), torch.from_numpy(d["ty"]).cuda()
sx, sy = d["sx"], d["sy"]
orders = {"random (generator)": np.arange(len(sx)),
          "by bearing": np.argsort(np.arctan2(sy - sy.mean(), sx - sx.mean()), kind="stable"),
          "by cell (row-major)": np.argsort(np.floor(sy / 0.5).astype(np.int64) * 100000 + np.floor(sx / 0.5).astype(np.int64), kind="stable")}
with NdtMatcher2D(fixed_iterations=30) as m:
    m.set_target(tx, ty)
    for name, o in orders.items():
        a, b = torch.from_numpy(np.ascontiguousarray(sx[o])).cuda(), torch.from_numpy(np.ascontiguousarray(sy[o])).cuda()
        torch.cuda.synchronize()
        for _ in range(5):
            m.align_async(a, b, d["init"], producer_complete=True)
        m.finish()
        t0 = time.perf_counter()
        for _ in range(100):
            m.align_async(a, b, d["init"], producer_complete=True)
        r = m.finish()
        el = time.perf_counter() - t0
        print(f"{name:22s}: {100 * 30 / el / 1e3:7.1f}k iterations/s, {1e6 * el / (100 * 31):.3f} us per launch, pose {r.pose}")
