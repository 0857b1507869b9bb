"""Where a pair's time goes in k_batch3: 100 MHz clock per phase, from a tools build of the library
(-DNDT_B3_PHASE_CLOCKS, which parks the ticks in the unused lower triangle of the result's H):
  hipcc ... -DNDT_B3_PHASE_CLOCKS -o gtsam_ndt_amd/lib/exp_phases.so ... ; NDT_HIP_LIB=.../exp_phases.so python tools/quick_batch3d_phases.py [firing]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtsam_ndt_amd import synth3d
from gtsam_ndt_amd.matcher import NdtBatch3D

n_pairs, K = 256, 30
d = synth3d.make_pair3d(n_azim=2048, pose=tuple(0.5 * np.array(synth3d.T_STAR_3D)))
if len(sys.argv) > 1 and sys.argv[1] == "firing":     # firing order (all 64 beams of one bearing, then the next bearing) instead of ring by ring
    for c in ("tx", "ty", "tz", "sx", "sy", "sz"):
        d[c] = np.ascontiguousarray(d[c].reshape(64, 2048).T).reshape(-1)
dev = torch.device("cuda:0")
t = [torch.from_numpy(np.tile(d[c], n_pairs)).to(dev) for c in ("tx", "ty", "tz")]
s = [torch.from_numpy(np.tile(d[c], n_pairs)).to(dev) for c in ("sx", "sy", "sz")]
off = torch.arange(n_pairs + 1, dtype=torch.int64, device=dev) * d["tx"].size
init = torch.zeros((n_pairs, 6), dtype=torch.float64, device=dev)
with NdtBatch3D(fixed_iterations=K) as b:
    b.align_dev(t, off, s, off, init)
    out = b.align_dev(t, off, s, off, init).cpu().numpy()
H = out[:, 6:42].reshape(n_pairs, 36)
ticks = H[:, 30:36]
names = ["bounds + geometry", "counts", "compaction", "sums", "finalise", f"{K} iterations"]
tot = ticks.sum(1).mean()
for j, nm in enumerate(names):
    print(f"{nm:20s} {ticks[:, j].mean() / 100:9.1f} us  ({100 * ticks[:, j].mean() / tot:4.1f} %)")
print(f"{'pair':20s} {tot / 100:9.1f} us")
