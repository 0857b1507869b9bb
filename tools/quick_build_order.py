"""Scratch: the 2D grid build (1M-point submap, 100k-point update) on points in the generator's random order against the
orders a front end delivers: the scan sorted by bearing around the sensor (a spinning lidar: neighbours in the array are
neighbours along the visible surface) and the submap room by room, each room's points by bearing around the room's centre
(a submap that is a sequence of such scans).  argv[1]: NDT_TUNE_BINNED_BUILD values to compare (1 = chunk-sorted, 2 = round-1
binned, 0 = atomics)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtMatcher2D
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1").split(",")]
d = synth.make_pair(3)


def by_bearing(x, y, cx, cy):
    o = np.argsort(np.arctan2(y - cy, x - cx), kind="stable")
    return x[o].copy(), y[o].copy()


def room_by_room(x, y, L=50.0):
    i, j = np.floor((x + 100.0) / L).clip(0, 3).astype(int), np.floor((y + 100.0) / L).clip(0, 3).astype(int)
    xs, ys = [], []
    for r in range(16):
        m = (j * 4 + i) == r
        a, b = by_bearing(x[m], y[m], (r % 4) * L - 100.0 + 0.5 * L, (r // 4) * L - 100.0 + 0.5 * L)
        xs.append(a); ys.append(b)
    return np.concatenate(xs), np.concatenate(ys)


cases = (("random order            ", (d["tx"], d["ty"]), (d["sx"], d["sy"])),
         ("rooms / scan by bearing ", room_by_room(d["tx"], d["ty"]), by_bearing(d["sx"], d["sy"], 0.0, 0.0)))
for v in variants:
    for label, (tx, ty), (sx, sy) in cases:
        tx, ty, sx, sy = (torch.from_numpy(a).cuda() for a in (tx, ty, sx, sy))
        torch.cuda.synchronize()
        with NdtMatcher2D(tuning={"binned_build": v}) as m:
            ts = []
            for _ in range(22):
                t0 = time.perf_counter(); info = m.set_target(tx, ty); ts.append(time.perf_counter() - t0)
            a = 1e6 * np.median(ts[2:])
            ts = []
            for _ in range(22):
                t0 = time.perf_counter(); m.add_target_points(sx, sy, pose=d["pose"]); ts.append(time.perf_counter() - t0)
            print(f"variant {v} {label}: set_target 1M {a:.1f} us (n_valid {info.n_valid}), update 100k {1e6 * np.median(ts[2:]):.1f} us", flush=True)
