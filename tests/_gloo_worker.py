"""Worker of tests/test_sharded_gloo.py (launched by torch.distributed.run, gloo, CPU only).

Exercises bench.py's N>1 plumbing -- shard_of() and reduce_counters() -- without a GPU: each
rank tallies ITS slot range of one synthetic stream (with the oracle standing in for the
device: the same u64 counter block layout [fwd | rev | k5 | k3]) and the blocks are summed
onto rank 0, where the result must equal the oracle's tables for the whole stream.
argv[1] = shard plan ("weak": 4000 reads per rank, "strong": 9001 reads in total -- not a
multiple of 2 or 3, so the ranges are uneven)."""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402
import pssbam_testlib as tl  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ge.load_pkg()
    from pss_bam_amd import synth
    plan = sys.argv[1] if len(sys.argv) > 1 else "weak"
    reads_arg = 4000 if plan == "weak" else 9001
    d = synth.config("C4", scale_genome=0.0003)
    region_len = d.pop("region_len")
    d["n_reads"], slot0, n = bench.shard_of(rank, world, reads_arg, plan)
    total = d["n_reads"]
    # the plan must tile the stream: contiguous, disjoint, complete
    spans = [bench.shard_of(r, world, reads_arg, plan) for r in range(world)]
    assert all(sp[0] == total for sp in spans) and spans[0][1] == 0
    assert all(spans[r][1] + spans[r][2] == spans[r + 1][1] for r in range(world - 1))
    assert spans[-1][1] + spans[-1][2] == total
    cfg = synth.make_cfg(**d)
    tmp = Path(tempfile.mkdtemp(prefix=f"gloo{rank}_"))
    fa, sam = tmp / "g.fa", tmp / "shard.sam"
    synth.fasta_host(cfg, fa)
    synth.sam_host(cfg, slot0, n, sam)
    orc = tl.Oracle()
    g = orc.load_genome(fa)
    po, ko = tl.PssOpts(region_len=region_len), tl.FkOpts(klen=3)
    f, r, _ = orc.pss(g, sam, po)
    k5, k3, _ = orc.fragkon(g, sam, ko)
    block = np.concatenate([f.ravel(), r.ravel(), k5.astype(np.uint64), k3.astype(np.uint64)])
    # make the top bit matter: u64 values beyond int64 range must survive the int64 transport
    block = block + np.uint64((1 << 63) // world + 12345)
    ctr = torch.from_numpy(block.view(np.int64).copy())
    bench.reduce_counters(ctr, world)
    if rank == 0:
        whole = tmp / "whole.sam"
        synth.sam_host(cfg, 0, total, whole)
        wf, wr, _ = orc.pss(g, whole, po)
        w5, w3, _ = orc.fragkon(g, whole, ko)
        want = np.concatenate([wf.ravel(), wr.ravel(), w5.astype(np.uint64), w3.astype(np.uint64)])
        with np.errstate(over="ignore"):
            want = want + np.uint64(world) * np.uint64((1 << 63) // world + 12345)
        got = ctr.numpy().view(np.uint64)
        assert np.array_equal(got, want), "sharded sum differs from the whole-stream tally"
        assert wf.sum() > 1000
        print("GLOO_SHARD_OK", plan, world, flush=True)
    orc.free_genome(g)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
