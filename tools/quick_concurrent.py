"""Scratch: aggregate fixed-30 iteration rate of N handles driven from N host threads (config 3 pair)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtMatcher2D

d = synth.make_pair(3)
tx, ty, sx, sy = (torch.from_numpy(d[k]).cuda() for k in ("tx", "ty", "sx", "sy"))
torch.cuda.synchronize()
K, STEPS = 30, 200
for n in (1, 2, 4, 8):
    ms = [NdtMatcher2D(fixed_iterations=K) for _ in range(n)]
    for m in ms:
        m.set_target(tx, ty)
        m.align_async(sx, sy, d["init"]); m.finish()
    go = threading.Barrier(n + 1)
    def work(m):
        go.wait()
        for _ in range(STEPS):
            m.align_async(sx, sy, d["init"])
        m.finish()
    th = [threading.Thread(target=work, args=(m,)) for m in ms]
    for t in th: t.start()
    go.wait(); t0 = time.perf_counter()
    for t in th: t.join()
    el = time.perf_counter() - t0
    print(f"{n} handles: {n * STEPS * K / el:,.0f} iterations/s aggregate ({1e6 * el / (STEPS * (K + 1)):.2f} us per launch per handle)")
    for m in ms: m.close()
