"""Multi-start chains on the config-3 pair: time per fixed-30 call and aggregate iterations/s for m = 1..8."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtMatcher2D
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
d = synth.make_pair(cfg)
tx, ty, sx, sy = (torch.from_numpy(d[k]).cuda() for k in ("tx", "ty", "sx", "sy"))
K = 30
with NdtMatcher2D(fixed_iterations=K) as m:
    m.set_target(tx, ty)
    for _ in range(5):
        m.align(sx, sy, d["init"])
    t0 = time.perf_counter()
    for _ in range(50):
        m.align(sx, sy, d["init"])
    t1 = (time.perf_counter() - t0) / 50
    print(f"single: {1e6 * t1:.1f} us/call, {K / t1:.0f} it/s, {1e6 * t1 / (K + 1):.2f} us/launch")
    n = sx.numel()
    for M in (1, 2, 4, 6, 8, 16, 32, 64):
        starts = [(d["init"][0] + 0.01 * k, d["init"][1] - 0.01 * k, 0.001 * k) for k in range(M)]
        for _ in range(5):
            m.align_multi_start(sx, sy, starts)
        t0 = time.perf_counter()
        for _ in range(50):
            r = m.align_multi_start(sx, sy, starts)
        t = (time.perf_counter() - t0) / 50
        us = 1e6 * t / (K + 1)
        alg = n * (8 + 24 * M)
        print(f"M={M}: {1e6 * t:.1f} us/call, {M * K / t:.0f} it/s aggregate, {us:.2f} us/launch, "
              f"algorithmic {alg / 1e6:.1f} MB/launch -> {alg / us / 1e6:.2f} TB/s = {alg / us / 1e6 / 8:.3f} of 8 TB/s")
