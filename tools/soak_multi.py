"""Scratch: repeated multi-context batches (host threads inside the library) and 3D handles in
threads against their single-thread results."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gtsam_ndt_amd import synth, synth3d, matcher as M

pairs = [synth.make_pair(4, pair_index=k, n_tgt=3000 + 3000 * (k % 4), n_src=2500 + 1500 * (k % 3)) for k in range(24)]
T = [(p["tx"], p["ty"]) for p in pairs]; S = [(p["sx"], p["sy"]) for p in pairs]; I = [p["init"] for p in pairs]
with M.NdtBatch2D() as b:
    ref = b.align(T, S, I)
bad = 0
with M.NdtMulti2D(devices=[0, 0, 0, 0]) as mm:
    for rep in range(40):
        out = mm.align(T, S, I)
        for k, (a, r) in enumerate(zip(out, ref)):
            if not (a.status == r.status and a.iterations == r.iterations and a.pose == r.pose):
                bad += 1; print("MULTI mismatch rep", rep, "pair", k, a.iterations, r.iterations, a.status, r.status)
print("multi-context soak: 40 batches x 24 pairs on 4 contexts, mismatches:", bad)

d = synth3d.make_pair3d(16, 512)
with M.NdtMatcher3D() as m:
    m.set_target(d["tx"], d["ty"], d["tz"]); r3 = m.align(d["sx"], d["sy"], d["sz"], d["init"])
bad3 = []
def work():
    with M.NdtMatcher3D() as m:
        for _ in range(4):
            m.set_target(d["tx"], d["ty"], d["tz"])
            r = m.align(d["sx"], d["sy"], d["sz"], d["init"])
            if not (r.status == r3.status and r.iterations == r3.iterations and r.pose == r3.pose):
                bad3.append((r.iterations, r3.iterations))
for rep in range(10):
    th = [threading.Thread(target=work) for _ in range(6)]
    for t in th: t.start()
    for t in th: t.join()
print("3D soak: 10 rounds x 6 threads x 4 alignments, mismatches:", len(bad3), bad3[:3])
