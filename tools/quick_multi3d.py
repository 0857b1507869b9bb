"""Timing of the 3D multi-scan chain (ndt3d_align_multi_scan_dev) on config-5-sized scans: m scans (4 distinct ones
replicated into separate buffers) against one cached voxel grid, fixed 30 iterations."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from gtsam_ndt_amd import synth3d
from gtsam_ndt_amd.matcher import NdtMatcher3D

K = 30
f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
tgt = synth3d.lidar_scan(101, (0.0,) * 6)
rng = np.random.default_rng(5)
poses = [tuple(np.array(synth3d.T_STAR_3D) * rng.uniform(-1.0, 1.0, 6)) for _ in range(4)]
base = [synth3d.lidar_scan(300 + k, p) for k, p in enumerate(poses)]
with NdtMatcher3D(fixed_iterations=K) as m:
    m.set_target(f(tgt[:, 0]), f(tgt[:, 1]), f(tgt[:, 2]))
    for count in (1, 8, 16, 32, 64):
        scans = [tuple(torch.from_numpy(f(base[k % 4][:, c])).cuda() for c in range(3)) for k in range(count)]
        inits = [(0.0,) * 6] * count
        torch.cuda.synchronize()
        r = m.align_multi_scan(scans, inits)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); r = m.align_multi_scan(scans, inits); ts.append(time.perf_counter() - t0)
        ms = 1e3 * float(np.median(ts))
        n = scans[0][0].numel()
        alg = count * n * 52 * K
        assert all(x.iterations == K and x.status == 0 for x in r)
        e = max(np.abs(np.array(x.pose) - np.array(poses[k % 4])).max() for k, x in enumerate(r))
        print(f"m = {count:2d}: {ms:.3f} ms per call, {count * K / ms * 1e3 / 1e3:.1f}k aggregate iterations/s, {ms * 1e3 / K:.1f} us per iteration, "
              f"algorithmic {alg / ms / 1e6:.0f} GB/s = {alg / ms / 1e6 / 8000:.3f} of 8 TB/s; max pose err vs truth {e:.4f}")
