"""Scratch: as quick_build.py, set_target only, with the two-round-trip build (grid = exactly the tiles)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtMatcher2D
d = synth.make_pair(3)
tx, ty = torch.from_numpy(d["tx"]).cuda(), torch.from_numpy(d["ty"]).cuda()
torch.cuda.synchronize()
for ss in (0, 1):
    with NdtMatcher2D(tuning={"single_sync_build": ss}) as m:
        ts = []
        for _ in range(22):
            t0 = time.perf_counter(); info = m.set_target(tx, ty); ts.append(time.perf_counter() - t0)
        print(f"single_sync {ss}: set_target 1M points: median {1e6 * np.median(ts[2:]):.1f} us", flush=True)
