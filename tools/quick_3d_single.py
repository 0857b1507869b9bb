"""k_iterate3 chain timing on config 5 (fixed 30 iterations, async back-to-back, HIP events): us per launch."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtsam_ndt_amd import synth3d
from gtsam_ndt_amd.matcher import NdtMatcher3D
d = synth3d.make_pair3d()
order = sys.argv[1] if len(sys.argv) > 1 else "ring"
if order == "firing":      # all 64 beams of one bearing, then the next bearing
    for c in ("tx", "ty", "tz", "sx", "sy", "sz"):
        d[c] = np.ascontiguousarray(d[c].reshape(64, 2048).T).reshape(-1)
s = [torch.from_numpy(d[k]).cuda() for k in ("sx", "sy", "sz")]
for mode in (0, 1):
    with NdtMatcher3D(fixed_iterations=30, hessian_mode=mode) as m:
        m.set_target(d["tx"], d["ty"], d["tz"])
        for _ in range(5):
            m.align_async(*s, d["init"], producer_complete=True)
        m.finish(); torch.cuda.synchronize()
        st = torch.cuda.ExternalStream(m.stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(40):
            m.align_async(*s, d["init"], producer_complete=True)
        e1.record(st); e1.synchronize()
        r = m.finish()
        print(f"mode {mode}: {1e3 * e0.elapsed_time(e1) / (40 * 31):.3f} us per launch; {os.environ.get('NDT_HIP_LIB', 'product library')}; {order} order")
