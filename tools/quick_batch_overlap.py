"""Scratch: the loop-closure batch with Biber's four overlapping grids (global-table variant, process_pair NG = 4) against
the single-grid batch on the same config-4 pairs."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from gtsam_ndt_amd import synth_dev
from gtsam_ndt_amd.matcher import NdtBatch2D

n_pairs, npts, K = 256, 100_000, 30
t = synth_dev.config4_batch(0, n_pairs, npts, npts)
for label, kw in (("one grid (on chip)      ", {}), ("four overlapping grids ", {"overlap_grids": 4})):
    with NdtBatch2D(fixed_iterations=K, **kw) as b:
        out = b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"])
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            out = b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"], out=out)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        rows = b.decode(out)
    ms = 1e3 * float(np.median(ts))
    err = max(np.abs(np.array(r.pose) - t["pose"][k].cpu().numpy()).max() for k, r in enumerate(rows)) if "pose" in t else float("nan")
    print(f"{label}: {ms:.3f} ms per {n_pairs} pairs, {n_pairs * K / ms * 1e3 / 1e6:.2f} M pair-iterations/s, statuses {sorted(set(r.status for r in rows))}, "
          f"max pose error vs truth {err:.2e}", flush=True)
