"""Scratch: short scans - single-workgroup loop (k_align_small) vs one launch per iteration."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gtsam_ndt_amd import synth, _lib as L
from gtsam_ndt_amd.matcher import NdtMatcher2D

lib = L.load()
for n_src in (360, 1000, 2048, 4096):
    d = synth.make_pair(2, n_tgt=100000, n_src=n_src) if n_src != 1000 else synth.make_pair(1)
    sx, sy = torch.from_numpy(d["sx"]).cuda(), torch.from_numpy(d["sy"]).cuda()
    torch.cuda.synchronize()
    for small in ("1", "0"):
        for kw in (dict(), dict(fixed_iterations=30)):
            with NdtMatcher2D(tuning={"short_scan_kernel": int(small)}, **kw) as m:
                m.set_target(d["tx"], d["ty"])
                init = (C.c_double * 3)(*d["init"])
                out = L.Result2D()
                lat = []
                for _ in range(40):
                    t0 = time.perf_counter()
                    lib.ndt2d_align_dev(m._h, sx.data_ptr(), sy.data_ptr(), sx.numel(), init, C.byref(out))
                    lat.append(time.perf_counter() - t0)
                med = float(np.median(lat[5:]))
                print(f"n_src {n_src:5d} small={small} {'fixed30' if kw else 'converged'}: {1e6*med:7.1f} us/call  iters {out.iterations:3d} "
                      f"-> {1e6*med/max(out.iterations,1):.2f} us/iter  pose {out.pose[0]:.6f} {out.pose[1]:.6f} {out.pose[2]:.6f} st {out.status}")
