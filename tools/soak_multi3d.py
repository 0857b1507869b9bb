"""Soak of the 3D multi-scan / multi-start chains' host protocol (converged mode: the flags are raised one launch
pair after the last start finishes; launches past the end must stay silent) and of the 3D batch hand-over: many
calls with varying numbers of starts, every result checked against its single-call reference."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtsam_ndt_amd import synth3d
from gtsam_ndt_amd.matcher import NdtBatch3D, NdtMatcher3D

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = np.random.default_rng(0)
f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
tgt = synth3d.lidar_scan(101, (0.0,) * 6, 32, 512, 0.02)
scans, t0 = [], time.time()
for k in range(8):
    p = tuple(np.array(synth3d.T_STAR_3D) * rng.uniform(-1, 1, 6))
    s = synth3d.lidar_scan(300 + k, p, int(rng.choice([16, 24, 32])), int(rng.choice([256, 300, 512])), 0.02)
    scans.append(tuple(torch.from_numpy(f(s[:, c])).cuda() for c in range(3)))
pool = [tuple(0.03 * rng.uniform(-1, 1, 6) * np.array([1, 1, 1, 0.1, 0.1, 0.1])) for _ in range(16)]
pool[3] = (400.0, 0.0, 0.0, 0.0, 0.0, 0.0)          # a start that ends at once (no hits)
with NdtMatcher3D() as m:
    m.set_target(f(tgt[:, 0]), f(tgt[:, 1]), f(tgt[:, 2]))
    ref = {(a, b): m.align(*scans[a], pool[b]) for a in range(8) for b in range(16)}
    bad = 0
    for it in range(N):
        mm = int(rng.integers(1, 65 if it % 10 == 0 else 13))
        pick = [(int(rng.integers(0, 8)), int(rng.integers(0, 16))) for _ in range(mm)]
        try:
            got = m.align_multi_scan([scans[a] for a, _ in pick], [pool[b] for _, b in pick])
        except Exception as e:
            print("FAILED at call", it, "m =", mm, str(e)[:160], flush=True)
            raise
        for key, g in zip(pick, got):
            r = ref[key]
            if not (g.pose == r.pose and g.iterations == r.iterations and g.status == r.status):
                bad += 1
        if it % 3 == 0:                               # single calls in between: the two chains share the stream and the flags
            a, b = int(rng.integers(0, 8)), int(rng.integers(0, 16))
            g = m.align(*scans[a], pool[b])
            bad += not (g.pose == ref[(a, b)].pose)
    print(f"{N} multi-scan calls, mismatches {bad}, {time.time() - t0:.1f}s", flush=True)
    assert bad == 0
