"""Scratch: section times of k_align_small (needs tools/bin/libndt_prof.so built with -DNDT_SMALL_PROFILE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NDT_HIP_LIB"] = os.path.join(ROOT, "tools", "bin", "libndt_prof.so")
import torch
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtMatcher2D
for n_src in (1000, 2048, 4096):
    d = synth.make_pair(1) if n_src == 1000 else synth.make_pair(2, n_tgt=100000, n_src=n_src)
    sx, sy = torch.from_numpy(d["sx"]).cuda(), torch.from_numpy(d["sy"]).cuda()
    with NdtMatcher2D(fixed_iterations=30) as m:
        m.set_target(d["tx"], d["ty"])
        for _ in range(3):
            m.align(sx, sy, d["init"])
        torch.cuda.synchronize()
