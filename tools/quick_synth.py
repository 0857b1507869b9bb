"""How long the device generator takes for the full config-4 batch (4096 pairs x 100k/100k points)."""
import sys, time
sys.path.insert(0, ".")
import torch
from gtsam_ndt_amd import synth_dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter()
    t = synth_dev.config4_batch(0, n)
    torch.cuda.synchronize()
    print(f"{n} pairs: {1e3 * (time.perf_counter() - t0):.1f} ms, {sum(v.numel() * v.element_size() for v in t.values()) / 1e9:.2f} GB")
    del t
