"""ndt3d_set_target of a device-resident config-5 scan and ndt3d_add_target_points_dev of a second scan into it: ms per call
(medians of 20).  NDT_HIP_LIB selects another build of the library to compare with."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtsam_ndt_amd import synth_dev
from gtsam_ndt_amd.matcher import NdtMatcher3D
lib = os.environ.get("NDT_HIP_LIB", "product").split("/")[-1]
for firing in (False, True):
    t = synth_dev.lidar_scan3d(5, (0.0,) * 6, firing_order=firing)
    s = synth_dev.lidar_scan3d(6, (0.2, -0.1, 0.0, 0.0, 0.0, 0.01), firing_order=firing)
    torch.cuda.synchronize()
    with NdtMatcher3D() as m:
        ts = []
        for _ in range(24):
            t0 = time.perf_counter(); m.set_target(*t); ts.append(time.perf_counter() - t0)
        m.reserve_target((-24.0, -24.0, -3.0), (24.0, 24.0, 7.0))
        m.add_target_points(*t, pose=(0.0,) * 6)
        tu = []
        for _ in range(24):
            t0 = time.perf_counter(); m.add_target_points(*s, pose=(0.2, -0.1, 0.0, 0.0, 0.0, 0.01)); tu.append(time.perf_counter() - t0)
    print(lib, "firing" if firing else "ring", f"set_target {1e3 * float(np.median(ts[4:])):.4f} ms, update into a 48 x 48 x 10 m submap {1e3 * float(np.median(tu[4:])):.4f} ms",
          flush=True)
