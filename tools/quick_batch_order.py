"""k_batch on config-4 pairs whose points are in random order (the synthetic generator's) against the same pairs with
every cloud sorted by bearing (the order a spinning lidar delivers): the LDS record gathers of neighbouring lanes then
hit the same or adjacent cells."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from gtsam_ndt_amd import synth_dev
from gtsam_ndt_amd.matcher import NdtBatch2D

n_pairs, npts, K = 512, 100_000, 30
t = synth_dev.config4_batch(0, n_pairs, npts, npts)


def sort_by_bearing(x, y):
    xs, ys = x.view(n_pairs, npts), y.view(n_pairs, npts)
    order = torch.argsort(torch.atan2(ys, xs), dim=1)
    return torch.gather(xs, 1, order).reshape(-1).contiguous(), torch.gather(ys, 1, order).reshape(-1).contiguous()


def run(tx, ty, sx, sy, label):
    with NdtBatch2D(fixed_iterations=K) as b:
        out = b.align_dev(tx, ty, t["toff"], sx, sy, t["soff"], t["init"])
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            out = b.align_dev(tx, ty, t["toff"], sx, sy, t["soff"], t["init"], out=out)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        rows = b.decode(out)
    ms = 1e3 * float(np.median(ts))
    print(f"{label}: {ms:.3f} ms per 512 pairs, {n_pairs * K / ms * 1e3 / 1e6:.2f} M pair-iterations/s", flush=True)
    return rows


r0 = run(t["tx"], t["ty"], t["sx"], t["sy"], "random point order  ")
stx, sty = sort_by_bearing(t["tx"], t["ty"])
ssx, ssy = sort_by_bearing(t["sx"], t["sy"])
r1 = run(stx, sty, ssx, ssy, "sorted by bearing   ")
r2 = run(t["tx"], t["ty"], ssx, ssy, "source sorted only  ")


def sort_by_cell(x, y, cell=0.5, tile=None):
    """The upper bound for an in-kernel counting sort of the source by cell: sources sorted by the
    row-major key of the 0.5 m cell they fall in under the initial pose (identity here: config 4 starts at zero)."""
    xs, ys = x.view(n_pairs, npts), y.view(n_pairs, npts)
    ix = torch.floor((xs - xs.min(dim=1, keepdim=True).values) / cell).to(torch.int64)
    iy = torch.floor((ys - ys.min(dim=1, keepdim=True).values) / cell).to(torch.int64)
    if tile:
        ix, iy = ix // tile, iy // tile
    order = torch.argsort(iy * 4096 + ix, dim=1, stable=True)
    return torch.gather(xs, 1, order).reshape(-1).contiguous(), torch.gather(ys, 1, order).reshape(-1).contiguous()


csx, csy = sort_by_cell(t["sx"], t["sy"])
r3 = run(t["tx"], t["ty"], csx, csy, "source sorted by cell (row-major)      ")
bsx, bsy = sort_by_cell(t["sx"], t["sy"], tile=8)
r4 = run(t["tx"], t["ty"], bsx, bsy, "source sorted by 8 x 8-cell block only ")
err = max(np.abs(np.array(a.pose) - np.array(b.pose)).max() for a, b in zip(r0, r1))
print("max pose difference between the orders (float32 summation order):", err)
