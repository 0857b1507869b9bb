"""Soak of the round-2 host protocols: converged-mode multi-start chains (flags raised one launch after the
last start finishes), 3D async calls and the batch kernel's capacity hand-over,
many calls each with varying shapes; every call is checked against a reference computed once."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from gtsam_ndt_amd import synth, synth3d, dist as nd
from gtsam_ndt_amd.matcher import NdtMatcher2D, NdtMatcher3D, NdtBatch2D

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = np.random.default_rng(0)
d = synth.make_pair(2, n_tgt=60_000, n_src=30_000)
sx, sy = torch.from_numpy(d["sx"]).cuda(), torch.from_numpy(d["sy"]).cuda()
pool = [(d["init"][0] + 0.01 * k, d["init"][1] - 0.007 * k, 0.0005 * k) for k in range(64)]
t0 = time.time()
for team in (0,):
    with NdtMatcher2D() as m:
        m.set_target(d["tx"], d["ty"])
        ref = {k: m.align(sx, sy, pool[k]) for k in range(64)}
        bad = 0
        for it in range(N):
            mm = int(rng.integers(1, 65 if it % 10 == 0 else 17))
            ks = rng.choice(64, size=mm, replace=False)
            try:
                got = m.align_multi_start(sx, sy, [pool[k] for k in ks])
            except Exception as e:
                print("FAILED at call", it, "m =", mm, "starts", sorted(ks.tolist()), str(e)[:120], flush=True)
                raise
            for k, g in zip(ks, got):
                r = ref[k]
                if not (g.pose == r.pose and g.iterations == r.iterations and g.status == r.status):
                    bad += 1
        print(f"team={team}: {N} multi-start calls, mismatches {bad}, {time.time()-t0:.1f}s", flush=True)
        assert bad == 0
d3 = synth3d.make_pair3d(n_azim=512)
s3 = [torch.from_numpy(d3[k]).cuda() for k in ("sx", "sy", "sz")]
with NdtMatcher3D() as m:
    m.set_target(d3["tx"], d3["ty"], d3["tz"])
    want = m.align(*s3, d3["init"])
    for it in range(N // 3):
        m.align_async(*s3, d3["init"])
        if it % 3 == 0:
            m.align_async(*s3, d3["init"])
        got = m.finish()
        assert got.pose == want.pose and got.iterations == want.iterations, it
print(f"3d async ok, {time.time()-t0:.1f}s", flush=True)
big = synth.make_pair(3, n_tgt=200_000, n_src=20_000)
small = [synth.make_pair(4, pair_index=k, n_tgt=20_000, n_src=20_000) for k in range(6)]
batch = small[:3] + [big] + small[3:]
t = {k: torch.from_numpy(v).cuda() for k, v in nd.pack_pairs(batch).items()}
with NdtBatch2D() as b:
    want = b.decode(b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"]))
    for it in range(N // 10):
        got = b.decode(b.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"]))
        assert [g.pose for g in got] == [w.pose for w in want], it
print(f"batch hand-over ok, {time.time()-t0:.1f}s")
