"""Scratch: GPU-side time of ndt2d_set_target_dev (1M points) and of a 100k-point submap update by HIP events on the
handle's stream (no profiler attached): the span from the first kernel's start to the read-back's end."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtMatcher2D
d = synth.make_pair(3)
tx, ty = torch.from_numpy(d["tx"]).cuda(), torch.from_numpy(d["ty"]).cuda()
sx, sy = torch.from_numpy(d["sx"]).cuda(), torch.from_numpy(d["sy"]).cuda()
torch.cuda.synchronize()
for v in [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "1,2").split(",")]:
    with NdtMatcher2D(tuning={"binned_build": v}) as m:
        st = torch.cuda.ExternalStream(m.stream)
        def timed(fn, reps=30):
            out = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); fn(); e1.record(st); e1.synchronize()
                out.append(1e3 * e0.elapsed_time(e1))
            return np.median(out[3:]), min(out[3:])
        m.set_target(tx, ty)
        print(f"variant {v}: set_target 1M: events median {timed(lambda: m.set_target(tx, ty))[0]:.1f} us", flush=True)
        print(f"variant {v}: add 100k (pose): events median {timed(lambda: m.add_target_points(sx, sy, pose=d['pose']))[0]:.1f} us", flush=True)
