"""Scratch: ndt2d_set_target_dev time for 1M points (config 3 target)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtMatcher2D
d = synth.make_pair(3)
tx, ty = torch.from_numpy(d["tx"]).cuda(), torch.from_numpy(d["ty"]).cuda()
torch.cuda.synchronize()
with NdtMatcher2D() as m:
    ts = []
    for _ in range(12):
        t0 = time.perf_counter(); m.set_target(tx, ty); ts.append(time.perf_counter() - t0)
    print(os.environ.get("NDT_HIP_LIB", "default"), "set_target 1M points: median %.1f us" % (1e6 * np.median(ts[2:])))

# incremental update: a 7200-point scan merged into the 1M-point submap
from gtsam_ndt_amd.synth import make_pair
with NdtMatcher2D() as m:
    m.set_target(tx, ty)
    sx = tx[:7200].contiguous(); sy = ty[:7200].contiguous()
    ts = []
    for _ in range(12):
        t0 = time.perf_counter(); m.add_target_points(sx, sy, pose=(0.01, 0.0, 0.0)); ts.append(time.perf_counter() - t0)
    print("add_target_points_dev 7200 points into the 1M-point grid: median %.1f us" % (1e6 * np.median(ts[2:])))
