#!/usr/bin/env python3
"""bench.py -- pss-bam per-read hot path on MI355X: aligned reads/s (whole job) + HBM roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--reads R_per_gpu]

One "step" = one complete pass of the hot path over the whole synthetic record stream
held by this rank (every alignment record decoded, filtered and tallied, counters reduced
to rank 0), with the records and the reference genome ALREADY RESIDENT IN HBM when the
timed region starts.  The workload is BASELINE.json's metric configuration (200 M x 150 bp
reads, 3.0 Gb 24-contig reference, end window N=25, coordinate-sorted) generated on the
device by the counter-based model in pss-bam_amd/csrc/synth_model.h; with --gpus N every
rank holds its own 200 M-read shard of a 200*N M-read stream (weak scaling, no data-path
collective; one RCCL sum of the ~7 KB counter block per step).

Rank 0 prints ONE JSON line (contract in the task statement) carrying two extra objects:
  roofline     algorithmic bytes per launch / mean launch duration of the tally kernel
               (HIP events on the engine's stream, around every launch of the timed steps)
  cpu_baseline the UNMODIFIED reference (oracle/_ref, -O2 build) timed on this box's host
               on a bounded prefix of the same stream (1 core: the reference has no threads)
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s copy rate)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def shard_of(rank: int, world: int, per_gpu: int) -> tuple[int, int, int]:
    """Weak-scaling shard plan: the stream has per_gpu*world reads; rank r owns the contiguous
    slot range [r*per_gpu, (r+1)*per_gpu).  Returns (total_reads, first_slot, n_slots)."""
    return per_gpu * world, rank * per_gpu, per_gpu


def reduce_counters(ctr, world: int):
    """Sums the per-rank counter blocks onto rank 0.  The blocks are u64; they travel as int64
    (two's complement addition is the same bit pattern), RCCL on GPUs, gloo in the CPU test."""
    import torch.distributed as dist
    if world > 1:
        dist.reduce(ctr, dst=0, op=dist.ReduceOp.SUM)
    return ctr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU (default: the config's count)")
    ap.add_argument("--unsorted", action="store_true", help="shuffled record order (gather stress)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 simple, 2 tiled")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="reads timed on the host reference")
    ap.add_argument("--region-len", type=int, default=None, help="override the configuration's -r N (parity cases / large-N passes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=1,
                    help="> 1: also time one reference process per core on disjoint shards (SURVEY 8d's all-cores figure)")
    ap.add_argument("--scale-genome", type=float, default=1.0)
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the counter reduce even with one rank")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    if not pkg.LIB_HIP.exists():
        raise SystemExit("libpssbam_hip.so missing: run `python __graft_entry__.py` first (no CPU fallback exists)")
    from pss_bam_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU implementation")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    cd = synth.config(args.config, sorted_=not args.unsorted, scale_genome=args.scale_genome)
    region_len = cd.pop("region_len")
    if args.region_len is not None:
        region_len = args.region_len
    klen = cd.pop("klen", None)
    per_gpu = args.reads if args.reads is not None else cd["n_reads"]
    cd["n_reads"], slot0, _ = shard_of(rank, world, per_gpu)
    cfg = synth.make_cfg(**cd)
    S = synth.lib()
    stream = torch.cuda.current_stream().cuda_stream
    n_contigs = int(cfg.n_contigs)
    names = [synth.contig_name(cfg, k) for k in range(n_contigs)]

    # ---- reference genome: generated on the device, handed to the engine D2D ----------------
    t0 = time.time()
    eng = pkg.Engine(pss=dict(region_len=region_len), kmer=dict(klen=klen) if klen else None, kernel=args.kernel)
    eng.set_stream(stream)
    contig_t = []
    for k in range(n_contigs):
        ln = int(cfg.contig_len[k])
        t = torch.empty(ln + 64, dtype=torch.uint8, device=dev)
        assert S.synth_genome_device(C.byref(cfg), k, t.data_ptr(), ln, stream) == 0
        contig_t.append(t)
    eng.set_genome_device([(names[k], contig_t[k].data_ptr(), int(cfg.contig_len[k])) for k in range(n_contigs)])
    eng.set_references(names)
    del contig_t
    torch.cuda.empty_cache()
    log(f"[bench] genome {sum(int(cfg.contig_len[k]) for k in range(n_contigs)) / 1e9:.2f} Gb on device in "
        f"{time.time() - t0:.1f}s")

    # ---- alignment records: generated on the device in < 4 GiB blocks -------------------------
    t0 = time.time()
    blocks = []  # (records tensor, offsets tensor, nbytes, n)
    LIMIT = (1 << 32) - (1 << 16)
    fixed = cfg.len_min == cfg.len_max and not cfg.cigar_mix
    if fixed:
        rec_bytes = int(synth.sizes_host(cfg, slot0, 1)[0])
        per_block = LIMIT // rec_bytes
        a = 0
        while a < per_gpu:
            n = min(per_block, per_gpu - a)
            rt = torch.empty(n * rec_bytes + 64, dtype=torch.uint8, device=dev)
            ot = torch.empty(n + 1, dtype=torch.int32, device=dev)
            assert S.synth_offsets_linear_device(ot.data_ptr(), n + 1, rec_bytes, stream) == 0
            assert S.synth_records_device(C.byref(cfg), slot0 + a, n, ot.data_ptr(), rt.data_ptr(), stream) == 0
            blocks.append((rt, ot, n * rec_bytes, n))
            a += n
    else:
        sizes = synth.sizes_host(cfg, slot0, per_gpu, threads=os.cpu_count() or 8)
        cum = np.zeros(per_gpu + 1, dtype=np.uint64)
        np.cumsum(sizes, out=cum[1:])
        a = 0
        while a < per_gpu:
            b = int(np.searchsorted(cum, cum[a] + np.uint64(LIMIT), side="right")) - 1
            b = min(max(b, a + 1), per_gpu)
            offs = (cum[a:b + 1] - cum[a]).astype(np.uint32)
            nbytes = int(offs[-1])
            rt = torch.empty(nbytes + 64, dtype=torch.uint8, device=dev)
            ot = torch.from_numpy(offs.view(np.int32)).to(dev)
            assert S.synth_records_device(C.byref(cfg), slot0 + a, b - a, ot.data_ptr(), rt.data_ptr(), stream) == 0
            blocks.append((rt, ot, nbytes, b - a))
            a = b
        del sizes, cum
    torch.cuda.synchronize()
    total_rec_bytes = sum(b[2] for b in blocks)
    log(f"[bench] {per_gpu / 1e6:.1f} M reads ({total_rec_bytes / 1e9:.2f} GB) on device in {len(blocks)} blocks, "
        f"{time.time() - t0:.1f}s")

    # ---- counters live in a torch tensor so RCCL can sum them in place -------------------------
    _, n_u64 = eng.counters_device()
    ctr = torch.zeros(n_u64, dtype=torch.int64, device=dev)
    eng.bind_counters(ctr.data_ptr(), n_u64)

    def step():
        ctr.zero_()
        for rt, ot, nbytes, n in blocks:
            eng.submit_device(rt.data_ptr(), nbytes, ot.data_ptr(), n)
        reduce_counters(ctr, 2 if use_dist else 1)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.kernel_time(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    kernel_ms, n_launch = eng.kernel_time(reset=True)
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    lay = eng.counter_layout()
    host_ctr = ctr.cpu().numpy().view(np.uint64)
    stats = {nm: int(host_ctr[lay["stats"] + i]) for i, nm in enumerate(pkg.ST_NAMES)}

    # algorithmic bytes: record incl. its length word + one u32 index entry +
    # 2*(N+2) reference bytes per read (+ 2k when the k-mer tally is fused in)
    alg_per_step = total_rec_bytes + per_gpu * (4 + 2 * (region_len + 2) + (2 * klen if klen else 0))
    achieved = alg_per_step * args.steps / (kernel_ms / 1e3) / 1e9 if kernel_ms > 0 else 0.0
    traffic = None
    tj = ROOT / "profiles" / "traffic.json"   # PMC-derived HBM bytes per read, collected in separate --pmc passes
    if tj.exists():
        try:
            per_read = json.loads(tj.read_text()).get(args.config, {}).get("hbm_bytes_per_read")
            if per_read is not None:
                traffic = per_read * per_gpu / len(blocks)   # per launch, like `achieved`
        except Exception:
            traffic = None

    out = {
        "metric": "aligned reads/s (whole node) + HBM GB/s fraction, 200M x 150bp BAM",
        "value": per_gpu * world * args.steps / dt,
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.config}: {per_gpu / 1e6:g} M x {cfg.len_min}"
                        + (f"-{cfg.len_max}" if cfg.len_max != cfg.len_min else "")
                        + f" bp BAM records per GPU, {sum(int(cfg.contig_len[k]) for k in range(n_contigs)) / 1e9:.2f} Gb "
                        f"{n_contigs}-contig reference, end window N={region_len}"
                        + (f", fragkon k={klen}" if klen else "")
                        + (", shuffled order" if args.unsorted else ", coordinate-sorted"),
            "reads_per_gpu": per_gpu,
            "record_bytes_per_gpu": total_rec_bytes,
            "launches_per_step": len(blocks),
            "sharding": f"contiguous record blocks, {world} rank(s), RCCL sum of {n_u64 * 8} B counters per step",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": alg_per_step / len(blocks),
            "kernel_ms_per_launch": kernel_ms / max(n_launch, 1),
            "launches_timed": n_launch,
        },
        "stats_last_step": stats,   # slow_path = records the tiled kernel read from global memory
    }

    # ---- CPU baseline: the reference itself on this box's host, on a bounded prefix -----------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"], out["parity_check"] = cpu_baseline(pkg, synth, eng, cfg, cd, region_len, klen,
                                                                    min(args.cpu_sample, per_gpu), names, args.cpu_cores)
        except Exception as ex:  # the baseline must never sink the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": 1, "kind": "reference",
                                   "sample": f"failed: {ex!r}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    eng.close()
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(pkg, synth, eng, cfg, cd, region_len, klen, n_sample, names, cpu_cores=1):
    """Times oracle/_ref/pss-bam.O2 (unmodified reference, compiled in the build container)
    on the first n_sample reads of the SORTED stream, fed as SAM text (no inflate cost charged
    to it), and checks the engine's tables for the same reads against the reference's."""
    import numpy as np
    import pssbam_testlib as tl

    tmp = Path(tempfile.mkdtemp(prefix="pssbam_cpu_"))
    cds = dict(cd)
    cds["sorted_"] = True
    scfg = synth.make_cfg(**cds)
    threads = os.cpu_count() or 8
    recs, offs = synth.records_host(scfg, 0, n_sample, threads=threads)
    last_contig = int(np.frombuffer(recs[int(offs[-2]) + 4:int(offs[-2]) + 8].tobytes(), dtype="<i4")[0])
    fa, sam, empty = tmp / "ref.fa", tmp / "sample.sam", tmp / "empty.sam"
    synth.fasta_host(scfg, fa, 0, last_contig + 1, threads=threads)
    synth.sam_host(scfg, 0, n_sample, sam)
    synth.sam_host(scfg, 0, 0, empty)
    o = tl.PssOpts(region_len=region_len)
    res = {}
    have_ref = tl.have_ref()
    variants = [("pss-bam.O2", "O2"), ("pss-bam", "ref_flags")] if have_ref else []
    want = None
    for exe, tag in variants:
        t = time.perf_counter()
        tl.run_ref_pss(fa, empty, tmp / f"e_{tag}", o, variant=exe, timeout=900)
        t_load = time.perf_counter() - t
        t = time.perf_counter()
        f, r, *_ = tl.run_ref_pss(fa, sam, tmp / f"s_{tag}", o, variant=exe, timeout=1800)
        t_full = time.perf_counter() - t
        res[tag] = n_sample / max(t_full - t_load, 1e-9)
        res[tag + "_load_s"] = t_load
        want = (f, r)
    if have_ref:
        kind, value = "reference", res["O2"]
        sample = (f"first {n_sample} reads of the sorted stream as SAM text + FASTA of contigs 0..{last_contig}; "
                  f"oracle/_ref/pss-bam.O2 (unmodified reference, gcc -O2), genome-load time "
                  f"({res['O2_load_s']:.1f}s, measured with an empty SAM) subtracted; same sources with the "
                  f"reference's own flags (-g, no -O): {res['ref_flags']:.0f} reads/s")
    else:
        orc = tl.Oracle()
        g = orc.load_genome(fa)
        t = time.perf_counter()
        f, r, _ = orc.pss(g, sam, o)
        value = n_sample / (time.perf_counter() - t)
        orc.free_genome(g)
        want = (f, r)
        kind = "port"
        sample = f"first {n_sample} reads of the sorted stream; oracle/liboracle.so (CPU restatement, -O2)"
    # parity of the HIP path against the same reads (separate engine: independent counters)
    e2 = pkg.Engine(pss=dict(region_len=region_len))
    g0 = [(names[k], synth.genome_host(scfg, k, threads=threads)) for k in range(last_contig + 1)]
    e2.set_genome_arrays(g0)
    e2.set_references(names)
    e2.submit(recs, offs)
    got = e2.finish()
    e2.close()
    ok = bool(np.array_equal(got.fwd, want[0]) and np.array_equal(got.rev, want[1]))
    parity = (f"bit-exact vs {kind} on the {n_sample}-read CPU sample" if ok
              else f"MISMATCH vs {kind} on the CPU sample")
    out = {"value": value, "unit": "reads/s", "cores": 1, "kind": kind, "sample": sample}
    if cpu_cores > 1 and have_ref:
        out["all_cores"] = cpu_all_cores(pkg, synth, scfg, region_len, names, tmp, o, cpu_cores,
                                         max(200_000, n_sample // 4))
    for p in tmp.iterdir():
        p.unlink()
    tmp.rmdir()
    return (out, parity)


def cpu_all_cores(pkg, synth, scfg, region_len, names, tmp, o, cores, per_shard):
    """one unmodified reference process per core on disjoint consecutive shards of the
    sorted stream, tables summed (tallies are additive) and checked against the engine's tables
    for the same reads.  Load time (every process parses the FASTA) is measured by the same
    number of concurrent processes on an empty SAM and subtracted."""
    import numpy as np
    import pssbam_testlib as tl
    from concurrent.futures import ThreadPoolExecutor

    threads = os.cpu_count() or 8
    total = cores * per_shard
    recs, offs = synth.records_host(scfg, total - 1, 1, threads=1)   # the last read tells how many contigs are needed
    last_contig = int(np.frombuffer(recs[4:8].tobytes(), dtype="<i4")[0])
    fa, empty = tmp / "ref_all.fa", tmp / "empty_all.sam"
    synth.fasta_host(scfg, fa, 0, last_contig + 1, threads=threads)
    synth.sam_host(scfg, 0, 0, empty)
    with ThreadPoolExecutor(min(threads, 32)) as ex:   # the writers release the GIL
        list(ex.map(lambda k: synth.sam_host(scfg, k * per_shard, per_shard, tmp / f"shard{k}.sam"), range(cores)))

    variant = os.environ.get("PSSBAM_REF_VARIANT", "pss-bam.O2")

    def wave(inputs, tag):
        t = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            res = list(ex.map(lambda kv: tl.run_ref_pss(fa, kv[1], tmp / f"{tag}{kv[0]}", o, variant=variant,
                                                        timeout=3000)[:2], enumerate(inputs)))
        return time.perf_counter() - t, res

    t_load, _ = wave([empty] * cores, "l")
    t_full, res = wave([tmp / f"shard{k}.sam" for k in range(cores)], "s")
    fwd = sum(r[0].astype(np.uint64) for r in res)
    rev = sum(r[1].astype(np.uint64) for r in res)
    e2 = pkg.Engine(pss=dict(region_len=region_len))
    e2.set_genome_arrays([(names[k], synth.genome_host(scfg, k, threads=threads)) for k in range(last_contig + 1)])
    e2.set_references(names)
    step = 8_000_000   # record blocks stay below 4 GiB
    for a in range(0, total, step):
        recs, offs = synth.records_host(scfg, a, min(step, total - a), threads=threads)
        e2.submit(recs, offs)
    got = e2.finish()
    e2.close()
    ok = bool(np.array_equal(got.fwd, fwd) and np.array_equal(got.rev, rev))
    return {"value": total / max(t_full - t_load, 1e-9), "unit": "reads/s", "cores": cores,
            "sample": f"{cores} concurrent oracle/_ref/{variant} processes x {per_shard} reads each (consecutive shards of "
                      f"the sorted stream, FASTA of contigs 0..{last_contig}); wall {t_full:.1f}s minus {t_load:.1f}s for the "
                      f"same {cores} processes on an empty SAM",
            "parity_check": "summed tables bit-exact vs the engine" if ok else "MISMATCH vs the engine"}


if __name__ == "__main__":
    main()
