"""Scratch: k_iterate launch time vs source size with 256- and 1024-thread workgroups (threshold choice)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtMatcher2D
base = synth.make_pair(3, n_src=1_000_000)
tx, ty = torch.from_numpy(base["tx"]).cuda(), torch.from_numpy(base["ty"]).cuda()
for n in (100_000, 150_000, 200_000, 300_000, 500_000, 1_000_000):
    sx, sy = torch.from_numpy(base["sx"][:n].copy()).cuda(), torch.from_numpy(base["sy"][:n].copy()).cuda()
    row = []
    for no_wide, thr in (("1", None), ("0", None)):
        # wide_threshold 0: never wide; 1: the wide variant at every size
        with NdtMatcher2D(fixed_iterations=30, tuning={"wide_threshold": 0 if no_wide == "1" else 1}) as m:
            m.set_target(tx, ty)
            for _ in range(3):
                m.align_async(sx, sy, base["init"]); m.finish()
            t0 = time.perf_counter()
            for _ in range(20):
                m.align_async(sx, sy, base["init"])
            m.finish()
            row.append(1e6 * (time.perf_counter() - t0) / (20 * 31))
    print(f"n_src {n:8d}: 256 threads {row[0]:.2f} us/launch   1024 threads {row[1]:.2f} us/launch")
