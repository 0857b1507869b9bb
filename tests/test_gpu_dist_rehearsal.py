"""The one-process-per-GPU deployment rehearsed on one GPU: two gloo ranks share cuda:0, each aligns its shard of the
loop-closure candidates with the batch kernels (2D and 3D) and the result rows are gathered with
gtsam_ndt_amd.dist.gather_results - the code path bench.py --gpus N runs with RCCL.  The gathered rows must equal the
single-process batch bit for bit (a pair's result does not depend on which rank, or which CU, aligned it)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSES3 = [(0.30, -0.20, 0.05, 0.01, -0.01, 0.03), (-0.25, 0.15, -0.04, -0.008, 0.012, -0.02), (0.10, 0.28, 0.02, 0.0, 0.015, 0.035),
          (-0.12, -0.22, 0.06, 0.012, 0.0, -0.03), (0.22, 0.05, -0.03, -0.01, -0.012, 0.015)]


def _pairs():
    from gtsam_ndt_amd import synth, synth3d
    p2 = [synth.make_pair(4, pair_index=k, n_tgt=6000 + 500 * k, n_src=5000) for k in range(5)]
    p3 = [synth3d.make_pair3d(n_elev=16, n_azim=256 + 32 * k, pose=p) for k, p in enumerate(POSES3)]
    return p2, p3


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from gtsam_ndt_amd import dist as nd
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtBatch3D
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        dev = torch.device("cuda:0")
        p2, p3 = _pairs()
        mine = nd.shard_range(len(p2), rank, world)
        b = nd.pack_pairs([p2[k] for k in mine])
        t = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
        with NdtBatch2D() as ctx:
            rows2 = nd.gather_results(ctx.align_dev(t["tx"], t["ty"], t["toff"], t["sx"], t["sy"], t["soff"], t["init"]), len(p2))
        b = nd.pack_pairs3d([p3[k] for k in mine])
        cu = lambda a: torch.from_numpy(a).to(dev)
        with NdtBatch3D() as ctx:
            out = ctx.align_dev([cu(a) for a in b["t"]], cu(b["toff"]), [cu(a) for a in b["s"]], cu(b["soff"]), cu(b["init"]))
            rows3 = nd.gather_results(out, len(p3))
        if rank == 0:
            q.put((rows2.cpu().numpy(), rows3.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_the_single_process_batch(gpu_lib):
    import torch.multiprocessing as mp
    from gtsam_ndt_amd.matcher import NdtBatch2D, NdtBatch3D
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from conftest import free_port
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        rows2, rows3 = q.get(timeout=300)
        for p in procs:
            p.join(timeout=120)
    finally:
        # a rank that died before q.put must not leave its peer blocked in a gloo collective, holding the GPU (and
        # ~2 GB of batch slabs) for the rest of the run
        for p in procs:
            if p.is_alive():
                p.terminate()
            p.join(timeout=30)
    assert all(p.exitcode == 0 for p in procs)
    p2, p3 = _pairs()
    with NdtBatch2D() as b:
        want2 = b.align([(p["tx"], p["ty"]) for p in p2], [(p["sx"], p["sy"]) for p in p2], [p["init"] for p in p2])
    with NdtBatch3D() as b:
        want3 = b.align([(p["tx"], p["ty"], p["tz"]) for p in p3], [(p["sx"], p["sy"], p["sz"]) for p in p3], [p["init"] for p in p3])
    got2 = NdtBatch2D.decode(__import__("torch").from_numpy(rows2))
    got3 = NdtBatch3D.decode(__import__("torch").from_numpy(rows3))
    assert len(got2) == 5 and len(got3) == 5
    for g, w in zip(got2, want2):
        assert g.status == w.status and g.status in (0, 1), (g, w)
        assert g.pose == w.pose and np.array_equal(g.H, w.H) and g.iterations == w.iterations, (g, w)
    for g, w in zip(got3, want3):
        assert g.status == w.status and g.status in (0, 1), (g, w)
        assert g.pose == w.pose and np.array_equal(g.H, w.H) and g.iterations == w.iterations, (g, w)
