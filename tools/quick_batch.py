"""Scratch timing of the batch path (not the bench contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtBatch2D

npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 30
uniq = min(npairs, 16)
t0 = time.time()
ps = [synth.make_pair(4, pair_index=k, n_tgt=npts, n_src=npts) for k in range(uniq)]
print(f"generated {uniq} pairs in {time.time()-t0:.1f}s")
dev = torch.device("cuda:0")
rep = (npairs + uniq - 1) // uniq
cat = lambda key: torch.from_numpy(np.concatenate([p[key] for p in ps] * rep)[: npairs * npts]).to(dev)
tx, ty, sx, sy = cat("tx"), cat("ty"), cat("sx"), cat("sy")
off = torch.arange(npairs + 1, dtype=torch.int64, device=dev) * npts
init = torch.tensor([ps[k % uniq]["init"] for k in range(npairs)], dtype=torch.float64, device=dev)
for fixed in (K, 0):
    with NdtBatch2D(fixed_iterations=fixed) as b:
        s = torch.cuda.ExternalStream(b.stream)
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            out = b.align_dev(tx, ty, off, sx, sy, off, init)
            e1.record(s); e1.synchronize()
            ms = e0.elapsed_time(e1)
            res = b.decode(out)
            its = sum(r.iterations for r in res)
            print(f"fixed={fixed}: {npairs} pairs x {npts} pts: {ms:.3f} ms -> {npairs/ms*1e3:.0f} pairs/s, "
                  f"{its/ms*1e3:.0f} pair-iters/s, statuses {sorted(set(r.status for r in res))}, pose0 {res[0].pose}")
