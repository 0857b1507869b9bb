"""ndt3d_set_target of a device-resident config-5 scan: ms per call (median of 20)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtsam_ndt_amd import synth_dev
from gtsam_ndt_amd.matcher import NdtMatcher3D
for firing in (False, True):
    t = synth_dev.lidar_scan3d(5, (0.0,) * 6, firing_order=firing)
    torch.cuda.synchronize()
    with NdtMatcher3D() as m:
        ts = []
        for _ in range(24):
            t0 = time.perf_counter(); m.set_target(*t); ts.append(time.perf_counter() - t0)
    print(os.environ.get("NDT_HIP_LIB", "product").split("/")[-1], "firing" if firing else "ring", f"{1e3 * float(np.median(ts[4:])):.4f} ms")
