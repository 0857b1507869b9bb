"""Timing of the 3D loop-closure batch (k_batch3) on replicated config-5 pairs.
usage: python tools/quick_batch3d.py [n_pairs=256] [distinct=4] [n_azim=2048] [mode=0] [cell=1.0] [iterations=30] [order=ring|firing]"""
import sys, time
import numpy as np
import torch

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gtsam_ndt_amd import synth3d
from gtsam_ndt_amd.matcher import NdtBatch3D, NdtMatcher3D

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
distinct = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_azim = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cell = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
K = int(sys.argv[6]) if len(sys.argv) > 6 else 30
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
poses = [tuple(np.array(synth3d.T_STAR_3D) * rng.uniform(-1.0, 1.0, 6)) for _ in range(distinct)]
ds = [synth3d.make_pair3d(n_azim=n_azim, pose=p) for p in poses]
if len(sys.argv) > 7 and sys.argv[7] == "firing":      # all 64 beams of one bearing, then the next bearing (a driver's order)
    for d_ in ds:
        for c in ("tx", "ty", "tz", "sx", "sy", "sz"):
            d_[c] = np.ascontiguousarray(d_[c].reshape(64, n_azim).T).reshape(-1)
npts = ds[0]["tx"].size
rep = [k % distinct for k in range(n_pairs)]
t = [torch.from_numpy(np.concatenate([ds[r][c] for r in rep])).to(dev) for c in ("tx", "ty", "tz")]
s = [torch.from_numpy(np.concatenate([ds[r][c] for r in rep])).to(dev) for c in ("sx", "sy", "sz")]
off = torch.arange(n_pairs + 1, dtype=torch.int64, device=dev) * npts
init = torch.zeros((n_pairs, 6), dtype=torch.float64, device=dev)
with NdtBatch3D(fixed_iterations=K, hessian_mode=mode, cell_size=cell, step_max_trans=cell) as b:
    out = b.align_dev(t, off, s, off, init)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        out = b.align_dev(t, off, s, off, init)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    res = b.decode(out)
ms = 1e3 * float(np.median(ts))
assert all(r.status == 0 and r.iterations == K for r in res)
with NdtMatcher3D(fixed_iterations=K, hessian_mode=mode, cell_size=cell, step_max_trans=cell) as m:
    for k in range(distinct):
        m.set_target(ds[k]["tx"], ds[k]["ty"], ds[k]["tz"])
        r = m.align(ds[k]["sx"], ds[k]["sy"], ds[k]["sz"], (0.0,) * 6)
        e = np.abs(np.array(r.pose) - np.array(res[k].pose)).max()
        print("pair", k, "batch vs single pose diff", e)
        assert e < 1e-4 or mode == 1, e          # Newton from a far start is chaotic: compared in the tests near the optimum
alg = n_pairs * (npts * 12 + K * npts * 12)
print(f"pairs {n_pairs} x {npts} pts, mode {mode}, cell {cell}: {ms:.3f} ms per batch, {n_pairs * K / ms * 1e3 / 1e6:.3f} M pair-iterations/s, "
      f"{n_pairs / ms * 1e3:.0f} pairs/s, algorithmic {alg / 1e9:.2f} GB -> {alg / ms / 1e6:.0f} GB/s = {alg / ms / 1e6 / 8000:.3f} of 8 TB/s")
