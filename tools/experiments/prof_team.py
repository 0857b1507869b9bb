"""In-kernel section times of k_align_xcd (needs tools/bin/libndt_xcdprof.so: a -DNDT_XCD_PROFILE build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["NDT_HIP_LIB"] = os.path.join(ROOT, "tools", "bin", "libndt_xcdprof.so")
sys.path.insert(0, ROOT)
import torch
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtMatcher2D
d = synth.make_pair(3)
tx, ty, sx, sy = (torch.from_numpy(d[k]).cuda() for k in ("tx", "ty", "sx", "sy"))
with NdtMatcher2D(fixed_iterations=30) as m:
    m.set_target(tx, ty)
    for _ in range(3):
        m.align(sx, sy, d["init"])
torch.cuda.synchronize()
