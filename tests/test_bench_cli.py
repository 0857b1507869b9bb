"""bench.py's behaviour where there is no GPU (this container): it must fail loudly, never fall back to a CPU path or print
a line, and the N > 1 self-launch must refuse before it starts any rank."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch
    return torch.cuda.device_count() == 0


@pytest.mark.skipif(not _no_gpu(), reason="needs a machine without a GPU")
def test_bench_refuses_to_launch_ranks_without_devices():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert "device" in r.stderr and "{" not in r.stdout


@pytest.mark.skipif(not _no_gpu(), reason="needs a machine without a GPU")
def test_bench_without_a_gpu_fails_instead_of_measuring_something_else():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert '"metric"' not in r.stdout
