"""bench.py on the GPU box: the contract line parses, carries the extra objects, and the
distributed path (RCCL init + the in-place u64 counter reduce) has executed on hardware."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    pr = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                        env=env)
    return pr


def test_bench_force_dist_one_rank_rccl():
    """--force-dist: a 1-rank RCCL communicator + dist.reduce of the bound counter tensor every step"""
    pr = _run(["--force-dist", "--force-weak-leg", "--reads", "2000000", "--steps", "2", "--warmup", "1", "--no-e2e", "--cpu-sample",
               "200000"])
    assert pr.returncode == 0, pr.stderr[-3000:]
    line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["unit"] == "reads/s" and d["value"] > 1e8
    assert d["stats_last_step"]["records"] == 2_000_000            # the reduce left rank 0's tables intact
    assert d["reduce_ms"] is not None and d["reduce_ms"] > 0
    assert d["per_rank"][0]["reads"] == 2_000_000
    w = d["weak_scaling"]                                            # the N > 1 companion leg, forced here
    assert w["scaling"] == "weak" and w["reads_per_gpu"] == 2_000_000 and w["value"] > 1e8 and w["reduce_ms"] > 0
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1.0
    assert d["cpu_baseline"]["cores"] == 1 and "bit-exact" in d["parity_check"]
    ac = d["cpu_baseline"]["all_cores"]                              # one reference process per core, by default
    assert "error" not in ac, ac
    assert ac["processes"] >= 2 and ac["cores"] > 1 and ac["value"] > 0 and "bit-exact" in ac["parity_check"]
    assert d["roofline"]["traffic"] is None or isinstance(d["roofline"]["traffic_stale"], bool)


def test_bench_bare_multi_gpu_launch_is_decided_before_any_gpu_call():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present")
    pr = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert pr.returncode != 0 and "needs 2 MI355X, 1 present" in pr.stderr


def test_bench_e2e_leg_small():
    """the file-to-tables leg on a small shape: tables of bin/pss-bam == the resident tally"""
    pr = _run(["--config", "C2", "--reads", "3000000", "--scale-genome", "0.02", "--steps", "2", "--warmup", "1",
               "--no-cpu-baseline"])
    assert pr.returncode == 0, pr.stderr[-3000:]
    d = json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][0])
    e = d["e2e"]
    assert "error" not in e, e
    assert e["reads"] == 3_000_000 and e["tables_check"].startswith("tables identical")
    assert e["wall_s"] > 0 and e["fasta_load_s"] is not None and len(e["wall_s_runs"]) == 5
    assert e["host_inflate_run"]["tables_identical"]
    assert 0 < e["gpu_busy_s"] < e["wall_s_foreground_exit"] and e["wall_s_foreground_exit"] > 0 and e["early_feed"]
    assert "error" not in e["level0"] and e["level0"]["reads"] == 3_000_000 and e["level0"]["bam_bytes"] > 5 * e["bam_bytes"]
    r = e["real_quals"]   # the same command on a BAM with sequencer-like quality strings
    assert "error" not in r, r
    assert r["reads"] == 3_000_000 and r["host_inflate_run"]["tables_identical"] and r["bam_bytes"] > 2 * e["bam_bytes"]
