"""N>1 path on CPU: two gloo ranks run bench.py's shard plan + counter reduce (see
tests/_gloo_worker.py).  No GPU involved; the GPU side of the same path is the driver's
multi-GPU bench."""
import subprocess
import sys
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent


@pytest.mark.parametrize("plan", ["weak", "strong"])
@pytest.mark.parametrize("world", [2, 3])
def test_shard_plan_and_reduce_world(world, plan):
    port = 29620 + world + (10 if plan == "strong" else 0)
    pr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                         "--master-addr", "127.0.0.1", "--master-port", str(port), str(HERE / "_gloo_worker.py"), plan],
                        capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stdout[-3000:] + pr.stderr[-3000:]
    assert f"GLOO_SHARD_OK {plan} {world}" in pr.stdout


def test_bench_self_launch_needs_gpus():
    """`python bench.py --gpus 2` launched bare (no WORLD_SIZE) decides before any GPU call: on a box
    without two MI355X it fails with exactly that diagnosis (with them it would start child ranks)."""
    import os
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present: the bare launch would run the real bench")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    pr = subprocess.run([sys.executable, str(HERE.parent / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                        capture_output=True, text=True, timeout=300, env=env)
    assert pr.returncode != 0
    assert "needs 2 MI355X" in pr.stderr, pr.stderr[-2000:]
