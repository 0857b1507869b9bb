"""Scratch: ndt2d_align (host arrays, converged mode) latency on the config-3 pair."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtMatcher2D
d = synth.make_pair(3)
with NdtMatcher2D() as m:
    m.set_target(d["tx"], d["ty"])
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); r = m.align(d["sx"], d["sy"], d["init"]); ts.append(time.perf_counter() - t0)
    print("ndt2d_align host arrays, converged: median %.1f us, %d iterations, pose %s" % (1e6 * np.median(ts[5:]), r.iterations, r.pose))
