"""Fused (k_iterate_multi) vs split (k_multi_solve + k_multi_body) chains: per-call time vs number of starts."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtMatcher2D
d = synth.make_pair(3)
tx, ty, sx, sy = (torch.from_numpy(d[k]).cuda() for k in ("tx", "ty", "sx", "sy"))
K = 30
n = sx.numel()
for name, sf in (("fused", 1000), ("split", 1)):
    with NdtMatcher2D(fixed_iterations=K, tuning={"split_from": sf}) as m:
        m.set_target(tx, ty)
        for M in (2, 4, 6, 8, 12, 16, 32, 64):
            starts = [(d["init"][0] + 0.002 * k, d["init"][1] - 0.002 * k, 0.0002 * k) for k in range(M)]
            for _ in range(4):
                m.align_multi_start(sx, sy, starts)
            ts = []
            for _ in range(25):
                t0 = time.perf_counter(); m.align_multi_start(sx, sy, starts); ts.append(time.perf_counter() - t0)
            t = float(np.median(ts))
            alg = n * (8 + 24 * M)
            print(f"{name} M={M:2d}: {1e6*t:7.1f} us/call  {M*K/t/1e3:7.0f}k it/s  {1e6*t/(K+1):6.2f} us/step  {alg*(K+1)/t/8e12:.3f} of 8 TB/s", flush=True)
