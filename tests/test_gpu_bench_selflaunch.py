"""bench.py --gpus N started WITHOUT a launcher must start its own N ranks: the N > 1 branch of
main() - process group, pair sharding, the result gather inside the timed step, the replica leg - runs here as two gloo
ranks sharing the one GPU of the test box (`--rehearse-on-one-gpu`; RCCL refuses two ranks on one device), through the
self-launch path, as a child process of the test."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "2", "--warmup", "1",
           "--pairs-per-rank", "8", *extra]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                     # ONE JSON line, rank 0's
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks(gpu_lib):
    out = _run(["--converged-batch"])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["metric"].startswith("NDT Gauss-Newton iters/sec") and out["unit"] == "iters/s"
    assert out["config"]["replicas"] == 2 and out["value"] > 0 and "replicas" in out["value_kind"]
    assert "roofline" in out and out["roofline"]["kernel"] == "k_iterate<GN>" and 0 < out["roofline"]["frac"] < 1
    b = out["batch"]
    assert b["n_gpus"] == 2 and b["config"]["pairs_total"] == 16 and b["config"]["pairs_per_gpu"] == 8
    assert "all_gather" in b["config"]["collective"]
    assert b["value"] > 0 and "roofline" in b and b["roofline"]["kernel"] == "k_batch<GN>"
    assert b["pose_err_vs_single_pair_max"]["pairs_checked"] >= 1 and b["pose_err_vs_single_pair_max"]["dx_m"] < 1e-6
    c = out["batch_converged"]                                    # strided shards, order restored by the gather
    assert c["n_gpus"] == 2 and c["pairs_converged"] == 16 and "strided" in c["sharding"]
    assert len(c["shard_balance"]["iterations_per_rank_strided"]) == 2
    assert c["pose_err_vs_truth_max"]["dx_m"] < 0.02              # rows matched with the right candidates after the gather


def test_bench_refuses_more_ranks_than_devices_without_the_rehearsal_flag(gpu_lib):
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "device(s) visible" in p.stderr and not p.stdout.strip()
