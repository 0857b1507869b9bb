"""Scratch: ndt2d_set_target_dev of the 1M-point config-3 target and a 100k-point submap update, per build variant
(NDT_TUNE_BINNED_BUILD 1 = chunk-sorted, 2 = round-1 binned, 0 = atomics).  Under rocprofv3 --kernel-trace --stats the
kernel populations are these calls'."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtMatcher2D
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2").split(",")]
d = synth.make_pair(3)
tx, ty = torch.from_numpy(d["tx"]).cuda(), torch.from_numpy(d["ty"]).cuda()
sx, sy = torch.from_numpy(d["sx"]).cuda(), torch.from_numpy(d["sy"]).cuda()
torch.cuda.synchronize()
for v in variants:
    with NdtMatcher2D(tuning={"binned_build": v}) as m:
        ts = []
        for _ in range(22):
            t0 = time.perf_counter(); info = m.set_target(tx, ty); ts.append(time.perf_counter() - t0)
        print(f"variant {v}: set_target 1M points: median {1e6 * np.median(ts[2:]):.1f} us (n_valid {info.n_valid})", flush=True)
        ts = []
        for _ in range(22):
            t0 = time.perf_counter(); m.add_target_points(sx, sy, pose=d["pose"]); ts.append(time.perf_counter() - t0)
        print(f"variant {v}: add_target_points_dev 100k points (moved by a pose) into the 1M-point grid: median {1e6 * np.median(ts[2:]):.1f} us "
              f"(n_valid {m.grid_info().n_valid})", flush=True)
