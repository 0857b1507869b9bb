"""The N > 1 path on CPU: world_size-2 gloo processes exercise the pair sharding and the one
result gather exactly as bench.py does on GPUs with RCCL (no compute: there is no GPU here)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gtsam_ndt_amd import dist as nd
from conftest import free_port


def test_shards_partition_the_batch():
    for total, world in ((4096, 8), (4096, 1), (1000, 3), (7, 8), (0, 2)):
        got = [i for r in range(world) for i in nd.shard_range(total, r, world)]
        assert got == list(range(total))
        sizes = nd.shard_sizes(total, world)
        assert max(sizes) - min(sizes) <= 1
        # strided shards (pair k -> rank k mod world) partition the batch too, with the same sizes
        st = [list(nd.shard_range(total, r, world, strided=True)) for r in range(world)]
        assert sorted(i for s in st for i in s) == list(range(total))
        assert [len(s) for s in st] == sizes
        assert all(i % world == r for r, s in enumerate(st) for i in s)
    with pytest.raises(ValueError):
        nd.shard_range(10, 2, 2)


def _fake_rows(idx, width=18):
    """Deterministic stand-in for ndt2d_result (18 doubles) / ndt3d_result (51) rows keyed by global pair index."""
    i = np.asarray(list(idx), dtype=np.float64)[:, None]
    return torch.from_numpy(i * 1000.0 + np.arange(width, dtype=np.float64)[None, :])


def _worker(rank, world, port, total, q):
    import datetime
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        mine = nd.shard_range(total, rank, world)
        allr = nd.gather_results(_fake_rows(mine), total)
        ok = bool(torch.equal(allr, _fake_rows(range(total))))
        all3 = nd.gather_results(_fake_rows(mine, 51), total)              # the 3D batch's rows through the same gather
        ok = ok and bool(torch.equal(all3, _fake_rows(range(total), 51)))
        # strided shards (converged-mode load balance): the gather restores global pair order
        mine_s = nd.shard_range(total, rank, world, strided=True)
        for width in (18, 51):
            alls = nd.gather_results(_fake_rows(mine_s, width), total, strided=True)
            ok = ok and bool(torch.equal(alls, _fake_rows(range(total), width)))
        t = nd.max_over_ranks(1.0 + rank)
        q.put((rank, ok, t))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [1024, 11])
def test_gather_restores_global_pair_order_world2(total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=180) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:                      # a rank that died must not leave its peer blocked in a collective
            if p.is_alive():
                p.terminate()
            p.join(timeout=30)
    assert all(p.exitcode == 0 for p in procs)
    assert sorted(r for r, _, _ in res) == [0, 1]
    assert all(ok for _, ok, _ in res)
    assert all(t == 2.0 for _, _, t in res)          # max over ranks of (1 + rank)


def test_pack_pairs_layout():
    from gtsam_ndt_amd import synth
    ps = [synth.make_pair(4, pair_index=k, n_tgt=100 + k, n_src=50 + 2 * k) for k in range(3)]
    b = nd.pack_pairs(ps)
    assert list(b["toff"]) == [0, 100, 201, 303] and list(b["soff"]) == [0, 50, 102, 156]
    assert np.array_equal(b["tx"][100:201], ps[1]["tx"]) and np.array_equal(b["sy"][102:156], ps[2]["sy"])
    assert b["init"].shape == (3, 3)
    from gtsam_ndt_amd import synth3d
    p3 = [synth3d.make_pair3d(n_elev=4, n_azim=16 + 8 * k) for k in range(2)]
    b3 = nd.pack_pairs3d(p3)
    assert list(b3["toff"]) == [0, 64, 160] and b3["init"].shape == (2, 6) and len(b3["t"]) == 3
    assert np.array_equal(b3["s"][2][64:160], p3[1]["sz"])
