"""Scratch: converged-mode call latency of the single-pair path vs chunk length."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gtsam_ndt_amd import synth, _lib as L
from gtsam_ndt_amd.matcher import NdtMatcher2D

d = synth.make_pair(3)
dev = torch.device("cuda:0")
tx, ty, sx, sy = (torch.from_numpy(d[k]).to(dev) for k in ("tx", "ty", "sx", "sy"))
torch.cuda.synchronize()
lib = L.load()
for chunk in (4, 8, 12, 16, 24, 32, 64):
    with NdtMatcher2D(tuning={"chunk_launches": chunk}) as m:
        m.set_target(tx, ty)
        init = (C.c_double * 3)(*d["init"])
        out = L.Result2D()
        lat = []
        for _ in range(30):
            t0 = time.perf_counter()
            lib.ndt2d_align_dev(m._h, sx.data_ptr(), sy.data_ptr(), sx.numel(), init, C.byref(out))
            lat.append(time.perf_counter() - t0)
        print(f"chunk {chunk:3d}: median {1e6*np.median(lat[5:]):.1f} us  min {1e6*min(lat):.1f} us  iters {out.iterations}")

# alternating evaluate / align on one handle: the chain graphs of both lengths stay cached
with NdtMatcher2D() as m:
    m.set_target(tx, ty)
    lat = []
    for _ in range(30):
        t0 = time.perf_counter()
        m.evaluate(d["sx"], d["sy"], d["init"])
        m.align(sx, sy, d["init"])
        lat.append(time.perf_counter() - t0)
    print(f"evaluate + align alternating: median {1e6*np.median(lat[5:]):.1f} us")
