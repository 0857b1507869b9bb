"""Scratch timing of the single-pair path (not the bench contract; see bench.py)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtMatcher2D

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
d = synth.make_pair(cfg)
dev = torch.device("cuda:0")
tx, ty, sx, sy = (torch.from_numpy(d[k]).to(dev) for k in ("tx", "ty", "sx", "sy"))
torch.cuda.synchronize()
with NdtMatcher2D(fixed_iterations=K) as m:
    for _ in range(3):
        t0 = time.perf_counter(); m.set_target(tx, ty); t1 = time.perf_counter()
    print(f"config {cfg}: set_target {1e3*(t1-t0):.3f} ms  n_valid {m.grid_info().n_valid}")
    for rep in range(5):
        t0 = time.perf_counter()
        m.align_async(sx, sy, d["init"]); r = m.finish()
        t1 = time.perf_counter()
        print(f"  align K={K}: {1e6*(t1-t0):.1f} us -> {K/(t1-t0):.0f} iters/s  pose {r.pose} it {r.iterations}")
with NdtMatcher2D() as m:
    m.set_target(tx, ty)
    t0 = time.perf_counter(); r = m.align(sx, sy, d["init"]); t1 = time.perf_counter()
    print(f"  converged align: {1e3*(t1-t0):.3f} ms iters {r.iterations} status {r.status} pose {r.pose} true {d['pose']}")
