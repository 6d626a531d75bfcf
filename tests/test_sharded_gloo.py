"""N>1 path on CPU: two gloo ranks run bench.py's shard plan + counter reduce (see
tests/_gloo_worker.py).  No GPU involved; the GPU side of the same path is the driver's
multi-GPU bench."""
import subprocess
import sys
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent


@pytest.mark.parametrize("world", [2, 3])
def test_shard_plan_and_reduce_world(world):
    port = 29620 + world
    pr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                         "--master-addr", "127.0.0.1", "--master-port", str(port), str(HERE / "_gloo_worker.py")],
                        capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stdout[-3000:] + pr.stderr[-3000:]
    assert f"GLOO_SHARD_OK {world}" in pr.stdout
