"""Scratch: 3D set_target / align timing (config 5)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gtsam_ndt_amd import synth3d
from gtsam_ndt_amd.matcher import NdtMatcher3D
d = synth3d.make_pair3d()
with NdtMatcher3D(fixed_iterations=30) as m:
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); m.set_target(d["tx"], d["ty"], d["tz"]); ts.append(time.perf_counter() - t0)
    print("set_target ms (host arrays, incl. upload):", [round(1e3 * t, 3) for t in ts])
    s = [torch.from_numpy(d[k]).cuda() for k in ("sx", "sy", "sz")]
    for _ in range(3):
        t0 = time.perf_counter(); r = m.align(*s, d["init"]); t1 = time.perf_counter()
    print("align fixed-30 ms:", round(1e3 * (t1 - t0), 3), r.iterations)
with NdtMatcher3D() as m:
    m.set_target(d["tx"], d["ty"], d["tz"])
    for _ in range(3):
        t0 = time.perf_counter(); r = m.align(*s, d["init"]); t1 = time.perf_counter()
    print("align converged ms:", round(1e3 * (t1 - t0), 3), r.iterations, r.status)
