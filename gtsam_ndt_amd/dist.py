"""Multi-GPU plumbing for the loop-closure batch: one process per GPU,
pairs sharded across ranks with no data-path collective, one all_gather of the per-pair
results at the end (RCCL over xGMI on GPUs - torch.distributed backend "nccl" - or gloo on
CPU for the tests).  A single alignment is never split across GPUs: it would need an
11-double all-reduce per ~5 us iteration."""
from __future__ import annotations

import numpy as np


def shard_range(n_pairs_total: int, rank: int, world: int, strided: bool = False) -> range:
    """Global pair indices owned by `rank`.  Default: a contiguous block (sizes differ by at most 1).
    strided=True: pair k goes to rank k mod world - in converged mode neighbouring
    candidates of a loop-closure list tend to need similar iteration counts (same place, same guess quality),
    and interleaving them evens the work of the ranks out; the shard sizes are the same either way."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    if strided:
        return range(rank, n_pairs_total, world)
    base, rem = divmod(n_pairs_total, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def shard_sizes(n_pairs_total: int, world: int) -> list[int]:
    return [len(shard_range(n_pairs_total, r, world)) for r in range(world)]


def gather_results(local, n_pairs_total: int, group=None, strided: bool = False):
    """all_gather the float64 result rows of every rank ([n_local, 18] for ndt2d_result, [n_local, 51] for
    ndt3d_result: the width is the tensor's) into global pair order.

    Shards may differ in length by one, so each rank pads to the longest shard, the padded
    blocks are gathered with one all_gather_into_tensor (the only collective of the path;
    ~72 KiB per rank at 512 pairs: latency-bound on xGMI), and the padding is dropped.
    strided=True undoes shard_range(..., strided=True): row j of rank r is global pair r + j * world,
    which is row r * longest + j of the gathered block - one index_select puts the rows back in order."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        local = local.cpu()            # rehearsal on a 1-GPU box: gather host copies
    sizes = shard_sizes(n_pairs_total, world)      # contiguous and strided shards have the same sizes
    assert local.shape[0] == sizes[dist.get_rank(group)], "local rows must match this rank's shard"
    width = local.shape[1]
    longest = max(sizes) if sizes else 0
    pad = torch.zeros((longest, width), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * longest, width), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    if strided:
        k = torch.arange(n_pairs_total, device=local.device)
        return out.index_select(0, (k % world) * longest + k // world)
    return torch.cat([out[r * longest: r * longest + sizes[r]] for r in range(world)], dim=0)


def max_over_ranks(value: float, device=None, group=None) -> float:
    import torch
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo":
        device = None
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def pack_pairs(pairs):
    """Concatenate a list of synth.make_pair() dicts into the batch layout (SoA + offsets)."""
    n = len(pairs)
    toff = np.zeros(n + 1, dtype=np.int64)
    soff = np.zeros(n + 1, dtype=np.int64)
    toff[1:] = np.cumsum([len(p["tx"]) for p in pairs])
    soff[1:] = np.cumsum([len(p["sx"]) for p in pairs])
    cat = lambda k: np.concatenate([p[k] for p in pairs])
    init = np.array([p["init"] for p in pairs], dtype=np.float64).reshape(n, 3)
    return {"tx": cat("tx"), "ty": cat("ty"), "toff": toff, "sx": cat("sx"), "sy": cat("sy"), "soff": soff,
            "init": init}


def pack_pairs3d(pairs):
    """The same for 3D pairs (synth3d.make_pair3d() dicts): the layout NdtBatch3D.align_dev takes."""
    n = len(pairs)
    toff = np.zeros(n + 1, dtype=np.int64)
    soff = np.zeros(n + 1, dtype=np.int64)
    toff[1:] = np.cumsum([len(p["tx"]) for p in pairs])
    soff[1:] = np.cumsum([len(p["sx"]) for p in pairs])
    cat = lambda k: np.concatenate([p[k] for p in pairs])
    init = np.array([p["init"] for p in pairs], dtype=np.float64).reshape(n, 6)
    return {"t": [cat("tx"), cat("ty"), cat("tz")], "toff": toff, "s": [cat("sx"), cat("sy"), cat("sz")], "soff": soff,
            "init": init}
