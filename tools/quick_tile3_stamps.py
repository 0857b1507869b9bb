"""Scratch: per-phase clocks of k_tile_accumulate3 (needs tools/bin/libndt_phase_clocks.so: the library built with
-DNDT_BUILD_PHASE_CLOCKS).  One row of the stamp table per workgroup (a (tile, share) pair of tile3_scan_block's list)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NDT_HIP_LIB"] = os.path.join(ROOT, "tools", "bin", "libndt_phase_clocks.so")
import numpy as np, torch
from gtsam_ndt_amd import synth_dev, _lib
from gtsam_ndt_amd.matcher import NdtMatcher3D
lib = _lib.load()
t = synth_dev.lidar_scan3d(5, (0.0,) * 6, firing_order=True)
torch.cuda.synchronize()
with NdtMatcher3D() as m:
    for _ in range(5):
        m.set_target(*t)
    buf = np.zeros((4096, 8), dtype=np.uint64)
    assert lib.ndt_exp_read_tile3_stamps(C.c_void_p(buf.ctypes.data)) == 0
s = buf.astype(np.int64)
live = np.nonzero(s[:, 5] > s[:, 0])[0]       # (stale rows of earlier calls have older clocks: keep the last call's window)
t_end = s[live, 5].max()
live = live[s[live, 0] > t_end - 10000]       # within 100 us of the end
t0 = s[live, 0].min()
print("workgroups:", len(live), " first start -> last end:", (t_end - t0) * 10, "ns")
names = ["init", "point loop", "hand-off", "finalise", "count add"]
fin = live[s[live, 4] > s[live, 3]]
for k in range(5):
    rows = live if k < 2 else fin
    dt = (s[rows, k + 1] - s[rows, k]) * 10.0
    print(f"  {names[k]:12s} median {np.median(dt):8.0f} ns   max {dt.max():8.0f} ns   ({len(rows)} workgroups)")
print("start skew (max):", (s[live, 0] - t0).max() * 10, "ns")
order = live[np.argsort(-(s[live, 5] - t0))][:10]
print("the last to end:  row  start  init  loop  handoff  finalise  add  end")
for b in order:
    v = s[b]
    print(f"  {b:5d} {(v[0]-t0)*10:7d} {(v[1]-v[0])*10:6d} {(v[2]-v[1])*10:6d} {(v[3]-v[2])*10:7d} {(v[4]-v[3])*10:7d} {(v[5]-v[4])*10:6d} {(v[5]-t0)*10:7d}")
