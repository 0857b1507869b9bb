"""Converged-mode call latency (host call -> result on the host) on config 3 against the chunk size of the launch
chain (NDT_TUNE_CHUNK_LAUNCHES) and against a fixed-K chain of the same length."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtMatcher2D

d = synth.make_pair(3)
tx, ty, sx, sy = (torch.from_numpy(d[k]).cuda() for k in ("tx", "ty", "sx", "sy"))
torch.cuda.synchronize()


def lat(m, reps=40):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = m.align(sx, sy, d["init"]); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts[5:])), r.iterations


for chunk in (4, 8, 12, 16, 24, 48):
    with NdtMatcher2D(tuning={"chunk_launches": chunk}) as m:
        m.set_target(tx, ty)
        ms, it = lat(m)
        print(f"converged, chunk {chunk:3d}: {ms:.4f} ms per call, {it} iterations")
with NdtMatcher2D(fixed_iterations=40) as m:
    m.set_target(tx, ty)
    ms, it = lat(m)
    print(f"fixed 40 iterations:  {ms:.4f} ms per call")
with NdtMatcher2D(step_scale=3.0) as m:
    m.set_target(tx, ty)
    for chunk in (4, 8, 16):
        m.set_tuning("chunk_launches", chunk)
        ms, it = lat(m)
        print(f"step_scale 3, chunk {chunk:3d}: {ms:.4f} ms per call, {it} iterations")
