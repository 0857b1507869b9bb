"""Scratch: per-phase clocks of k_tile_gather (needs tools/bin/libndt_phase_clocks.so: -DNDT_BUILD_PHASE_CLOCKS)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NDT_HIP_LIB"] = os.path.join(ROOT, "tools", "bin", "libndt_phase_clocks.so")
import numpy as np, torch
from gtsam_ndt_amd import synth, _lib
from gtsam_ndt_amd.matcher import NdtMatcher2D
d = synth.make_pair(3)
tx, ty = torch.from_numpy(d["tx"]).cuda(), torch.from_numpy(d["ty"]).cuda()
sx, sy = torch.from_numpy(d["sx"]).cuda(), torch.from_numpy(d["sy"]).cuda()
lib = _lib.load()
def stamps():
    buf = np.zeros((1024, 8), dtype=np.uint64)
    assert lib.ndt_exp_read_stamps(C.c_void_p(buf.ctypes.data)) == 0
    return buf
with NdtMatcher2D() as m:
    for _ in range(5):
        m.set_target(tx, ty)
    s = stamps().astype(np.int64)
    live = s[:, 5] > s[:, 0]
    t0 = s[live, 0].min()
    print("set_target 1M: workgroups that ran:", int(live.sum()))
    names = ["row+any", "init", "point loop", "barrier", "finalise math", "stores"]
    for k in range(5):
        dt = (s[live, k + 1] - s[live, k]) * 10.0
        print(f"  {names[k]:14s} median {np.median(dt):8.0f} ns   max {dt.max():8.0f} ns")
    print(f"  start skew: max {(s[live,0]-t0).max()*10} ns; whole kernel (first start to last end) {(s[live,5].max()-t0)*10} ns")
    for _ in range(3):
        m.add_target_points(sx, sy, pose=d["pose"])
    s = stamps().astype(np.int64)

# per-tile view of the last 1M build: loop time against the tile's point count
with NdtMatcher2D() as m:
    for _ in range(3):
        info = m.set_target(tx, ty)
    s = stamps().astype(np.int64)
fx = np.floor((d["tx"] - info.ox) * info.inv_cell).astype(int); fy = np.floor((d["ty"] - info.oy) * info.inv_cell).astype(int)
ntx = (info.width + 31) // 32
h = np.bincount((fy >> 5) * ntx + (fx >> 5), minlength=ntx * ((info.height + 31) // 32))
live = np.nonzero(s[:, 5] > s[:, 0])[0]
t0 = s[live, 0].min()
order = live[np.argsort(-(s[live, 2] - s[live, 1]))]
print("tile  points  start_ns  row  init  loop  barrier  fin  stores_end_ns")
for b in list(order[:12]) + list(order[-4:]):
    v = s[b]
    print(f"{b:4d} {h[b] if b < len(h) else -1:7d} {(v[0]-t0)*10:8d} {(v[1]-v[0])*10:5d} {(v[2]-v[1])*10:6d} {(v[3]-v[2])*10:6d} {(v[4]-v[3])*10:6d} {(v[5]-v[4])*10:6d} {(v[5]-t0)*10:8d}")

ss = np.zeros((4096, 2), dtype=np.uint64)
assert lib.ndt_exp_read_sort_stamps(C.c_void_p(ss.ctypes.data)) == 0
ss = ss.astype(np.int64)[:244]
print(f"k_chunk_sort: first start 0, last start {(ss[:,0].max()-ss[:,0].min())*10} ns, last end {(ss[:,1].max()-ss[:,0].min())*10} ns, "
      f"median workgroup {np.median(ss[:,1]-ss[:,0])*10:.0f} ns")
print(f"gather first workgroup start - sort last workgroup end: {(t0 - ss[:,1].max())*10} ns; gather last end - sort first start: {(s[live,5].max()-ss[:,0].min())*10} ns")

we = np.zeros((1024, 16), dtype=np.uint64)
assert lib.ndt_exp_read_wave_ends(C.c_void_p(we.ctypes.data)) == 0
we = we.astype(np.int64)
print(f"last wave end of any workgroup - first workgroup start: {(we[live].max() - t0) * 10} ns; per-workgroup (last wave end - thread 0's last stamp): "
      f"median {np.median(we[live].max(axis=1) - s[live, 5]) * 10:.0f} ns, max {(we[live].max(axis=1) - s[live, 5]).max() * 10} ns")

# the submap update: 100k points (moved by a pose) into the 1M-point grid
with NdtMatcher2D() as m:
    m.set_target(tx, ty)
    for _ in range(4):
        m.add_target_points(sx, sy, pose=d["pose"])
    s = stamps().astype(np.int64)
    assert lib.ndt_exp_read_wave_ends(C.c_void_p(we.ctypes.data if False else np.zeros(1).ctypes.data)) if False else True
live = np.nonzero(s[:, 5] > s[:, 0])[0]
# (stamps of tiles that returned at once are stale: keep those whose start is within 100 us of the latest start)
live = live[s[live, 0] > s[live, 0].max() - 10000]
t0 = s[live, 0].min()
print("submap update: touched tiles", len(live))
print("tile  start_ns  row+init  loop(wave0)  wait  finalise  stores  end_ns")
for b in live[np.argsort(-(s[live, 5] - t0))][:10]:
    v = s[b]
    print(f"{b:4d} {(v[0]-t0)*10:8d} {(v[1]-v[0])*10:8d} {(v[2]-v[1])*10:8d} {(v[3]-v[2])*10:8d} {(v[4]-v[3])*10:8d} {(v[5]-v[4])*10:8d} {(v[5]-t0)*10:8d}")
