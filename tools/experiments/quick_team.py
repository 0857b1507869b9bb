"""Team kernel vs launch-per-iteration path: fixed-30 alignments enqueued back to back (as bench.py times the
headline), and per-call latency; multi-start through teams."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtMatcher2D
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
d = synth.make_pair(cfg)
tx, ty, sx, sy = (torch.from_numpy(d[k]).cuda() for k in ("tx", "ty", "sx", "sy"))
K = 30
for team in (0, 1):
    with NdtMatcher2D(fixed_iterations=K, tuning={"team_kernel": team}) as m:
        m.set_target(tx, ty)
        for _ in range(5):
            m.align_async(sx, sy, d["init"], producer_complete=True)
        m.finish()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            m.align_async(sx, sy, d["init"], producer_complete=True)
        r = m.finish()
        t = (time.perf_counter() - t0) / 100
        t0 = time.perf_counter()
        for _ in range(50):
            r = m.align(sx, sy, d["init"])
        tc = (time.perf_counter() - t0) / 50
        print(f"team={team}: back to back {1e6*t:.1f} us/alignment = {K/t:.0f} it/s ({1e6*t/K:.2f} us/iteration); per call {1e6*tc:.1f} us; iters {r.iterations} fallbacks {m.team_fallbacks}")
    with NdtMatcher2D(tuning={"team_kernel": team}) as m:
        m.set_target(tx, ty)
        m.align(sx, sy, d["init"])
        t0 = time.perf_counter()
        for _ in range(30):
            r = m.align(sx, sy, d["init"])
        tc = (time.perf_counter() - t0) / 30
        print(f"   converged: {1e6*tc:.1f} us per call, {r.iterations} iterations")
with NdtMatcher2D(fixed_iterations=K) as m:
    m.set_target(tx, ty)
    n = sx.numel()
    for M in (1, 2, 4, 8, 16):
        starts = [(d["init"][0] + 0.01 * k, d["init"][1] - 0.01 * k, 0.001 * k) for k in range(M)]
        for _ in range(5):
            m.align_multi_start(sx, sy, starts)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); m.align_multi_start(sx, sy, starts); ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        alg = n * (8 + 24 * M)
        print(f"multi M={M}: {1e6*t:.1f} us/call, {M*K/t:.0f} it/s aggregate, {alg*K/t/1e12:.2f} TB/s = {alg*K/t/8e12:.3f} of 8 TB/s, fallbacks {m.team_fallbacks}")
