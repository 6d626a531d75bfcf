#!/usr/bin/env python3
"""bench.py -- pss-bam per-read hot path on MI355X: aligned reads/s (whole job) + HBM roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--scaling strong|weak]

One "step" = one complete pass of the hot path over the synthetic record stream: every
alignment record of every rank's shard decoded, filtered and tallied, counters reduced to
rank 0 -- with the records and the reference genome ALREADY RESIDENT IN HBM when the timed
region starts (HBM-resident records: BGZF inflate and PCIe are NOT in `value`; the `e2e`
object carries the file-to-tables figure of the same shape).  The workload is BASELINE.json's
metric configuration (C3: 200 M x 150 bp reads, 3.1 Gb 24-contig reference, end window N=25,
coordinate-sorted) generated on the device by the counter-based model in
pss-bam_amd/csrc/synth_model.h.

--gpus N (the loop being sharded is /root/reference/pss-bam.c:764-783):
  * launched bare (no WORLD_SIZE in the environment) bench.py starts N child ranks itself
    through `python -m torch.distributed.run` BEFORE any GPU call and relays rank 0's line;
    under a launcher (the driver's torch.distributed.run) it is a rank.
  * --scaling strong (default): the configuration's reads IN TOTAL; rank r owns the contiguous
    slot range [r*total/N, (r+1)*total/N) of the one sorted stream.  --scaling weak: the
    configuration's reads PER GPU.  For N > 1 the strong line also carries a `weak_scaling`
    object measured in the same run.  No data-path collective; one RCCL sum of the ~7 KB
    counter block per step.

Rank 0 prints ONE JSON line (contract in the task statement) with these extra objects:
  roofline     algorithmic bytes per launch / mean launch duration of the tally kernel
               (HIP events on the engine's stream, around every launch of the timed steps);
               `frac` against the 8 TB/s spec peak, `frac_of_traffic` from the PMC-measured bytes
  cpu_baseline the UNMODIFIED reference (oracle/_ref, -O2 build) timed on this box's host
               on a bounded prefix of the same stream (1 core: the reference has no threads)
  e2e          (N=1) bin/pss-bam on a generated level-1 BGZF BAM + FASTA of the named shape:
               wall seconds, reads/s, per-stage seconds, tables checked against the resident tally
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import re
import shutil
import signal
import socket
import subprocess
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


def shard_of(rank: int, world: int, reads: int, scaling: str = "weak") -> tuple[int, int, int]:
    """Shard plan -> (total_reads, first_slot, n_slots) of rank `rank`.
    weak:   the stream has reads*world slots, rank r owns [r*reads, (r+1)*reads)
    strong: the stream has `reads` slots in total, rank r owns [r*reads//world, (r+1)*reads//world)
    Either way the ranges are contiguous, disjoint and cover the stream (a sorted BAM keeps
    every GPU's reference working set local)."""
    if scaling == "weak":
        return reads * world, rank * reads, reads
    if scaling != "strong":
        raise ValueError(f"unknown scaling {scaling!r}")
    first = rank * reads // world
    return reads, first, (rank + 1) * reads // world - first


def reduce_counters(ctr, world: int):
    """Sums the per-rank counter blocks onto rank 0.  The blocks are u64; they travel as int64
    (two's complement addition is the same bit pattern), RCCL on GPUs, gloo in the CPU test."""
    import torch.distributed as dist
    if world > 1:
        dist.reduce(ctr, dst=0, op=dist.ReduceOp.SUM)
    return ctr


def effective_cpus() -> float:
    """CPUs this process may actually use: the cgroup quota (cpu.max) when there is one,
    else the affinity mask.  A gpurun box shows 128-256 logical CPUs but grants 16."""
    try:
        n = float(len(os.sched_getaffinity(0)))
    except AttributeError:
        n = float(os.cpu_count() or 1)
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(p).read_text().split()
            if p.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, int(txt[0]) / int(txt[1]))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, q / int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text()))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def worker_threads() -> int:
    return max(1, int(round(effective_cpus())))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: the configuration's reads in total over all GPUs; weak: per GPU")
    ap.add_argument("--reads", type=int, default=None,
                    help="read count (total for --scaling strong, per GPU for weak; default: the config's)")
    ap.add_argument("--unsorted", action="store_true", help="shuffled record order (gather stress)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 simple, 2 tiled")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="reads timed on the host reference")
    ap.add_argument("--region-len", type=int, default=None, help="override the configuration's -r N (parity cases / large-N passes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=0,
                    help="processes of the all-cores CPU figure (SURVEY 8d: one reference process per core on disjoint shards); "
                         "0 = the CPUs this job may use (cgroup cpu.max), 1 = skip it")
    ap.add_argument("--scale-genome", type=float, default=1.0)
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the counter reduce even with one rank")
    ap.add_argument("--no-weak-leg", action="store_true", help="N > 1, strong: skip the additional weak-scaling measurement")
    ap.add_argument("--force-weak-leg", action="store_true", help="run the weak-scaling companion leg even with one rank (test hook)")
    ap.add_argument("--e2e-reads", type=int, default=None,
                    help="reads of the file-to-tables leg (N=1; default: the configuration's count, bounded by free disk)")
    ap.add_argument("--no-e2e", action="store_true")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh child ranks (a child process,
    never a re-exec, and before this process has touched the GPU) and relay their output."""
    import torch  # device_count() does not initialise the GPU on this image
    have = torch.cuda.device_count()
    if have < args.gpus:
        print(f"bench.py --gpus {args.gpus} needs {args.gpus} MI355X, {have} present "
              "(the product path has no CPU implementation)", file=sys.stderr)
        return 1
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}).returncode


class Workload:
    """One rank's shard of a synthetic stream, resident in HBM: < 4 GiB record blocks + offsets."""

    def __init__(self, synth, torch, dev, stream, cfg, slot0: int, n: int):
        import numpy as np
        S = synth.lib()
        self.blocks = []  # (records tensor, offsets tensor, nbytes, n)
        self.n_reads = n
        LIMIT = (1 << 32) - (1 << 16)
        fixed = cfg.len_min == cfg.len_max and not cfg.cigar_mix
        if fixed and n:
            rec_bytes = int(synth.sizes_host(cfg, slot0, 1)[0])
            per_block = LIMIT // rec_bytes
            a = 0
            while a < n:
                m = min(per_block, n - a)
                rt = torch.empty(m * rec_bytes + 64, dtype=torch.uint8, device=dev)
                ot = torch.empty(m + 1, dtype=torch.int32, device=dev)
                assert S.synth_offsets_linear_device(ot.data_ptr(), m + 1, rec_bytes, stream) == 0
                assert S.synth_records_device(C.byref(cfg), slot0 + a, m, ot.data_ptr(), rt.data_ptr(), stream) == 0
                self.blocks.append((rt, ot, m * rec_bytes, m))
                a += m
        elif n:
            sizes = synth.sizes_host(cfg, slot0, n, threads=worker_threads())
            cum = np.zeros(n + 1, dtype=np.uint64)
            np.cumsum(sizes, out=cum[1:])
            a = 0
            while a < n:
                b = int(np.searchsorted(cum, cum[a] + np.uint64(LIMIT), side="right")) - 1
                b = min(max(b, a + 1), n)
                offs = (cum[a:b + 1] - cum[a]).astype(np.uint32)
                nbytes = int(offs[-1])
                rt = torch.empty(nbytes + 64, dtype=torch.uint8, device=dev)
                ot = torch.from_numpy(offs.view(np.int32)).to(dev)
                assert S.synth_records_device(C.byref(cfg), slot0 + a, b - a, ot.data_ptr(), rt.data_ptr(), stream) == 0
                self.blocks.append((rt, ot, nbytes, b - a))
                a = b
        torch.cuda.synchronize()
        self.rec_bytes = sum(b[2] for b in self.blocks)


def git_blob_id(path: Path) -> str:
    """what `git hash-object` prints for the file (the GPU box has no .git)"""
    import hashlib
    data = path.read_bytes()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def kernel_blobs() -> dict:
    d = ROOT / "pss-bam_amd" / "csrc"
    return {n: git_blob_id(d / n) for n in ("tally_kernels.h", "record_decode.h")}


def traffic_entry(config: str, unsorted: bool, klen, region_len: int, reads_per_launch: float):
    """PMC-derived HBM bytes per read for THIS configuration and launch size, or None.
    profiles/traffic.json holds one entry per (config, order, launch size), each stamped with the
    command and git SHA that produced it (tools/pmc_traffic.py: separate --pmc passes, FETCH_SIZE
    doubled per the microarch guide's gfx950 note); an entry counts when its launch size is within
    10 % of the benchmarked one."""
    tj = ROOT / "profiles" / "traffic.json"
    try:
        entries = json.loads(tj.read_text()).get("entries", [])
    except (OSError, ValueError):
        return None
    for en in entries:
        try:
            if (en["config"] == config and bool(en.get("unsorted", False)) == bool(unsorted)
                    and en.get("klen") == klen and en.get("region_len") == region_len
                    and abs(en["reads_per_launch"] - reads_per_launch) <= 0.10 * reads_per_launch):
                return en
        except (KeyError, TypeError):
            continue
    return None


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

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
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    reads_arg = args.reads if args.reads is not None else cd["n_reads"]
    S = synth.lib()
    stream = torch.cuda.current_stream().cuda_stream

    def cfg_for(scaling):
        total, slot0, n = shard_of(rank, world, reads_arg, scaling)
        d = dict(cd)
        d["n_reads"] = total
        return synth.make_cfg(**d), total, slot0, n

    cfg, total_reads, slot0, my_reads = cfg_for(args.scaling)
    n_contigs = int(cfg.n_contigs)
    names = [synth.contig_name(cfg, k) for k in range(n_contigs)]
    genome_gb = sum(int(cfg.contig_len[k]) for k in range(n_contigs)) / 1e9

    # ---- reference genome: generated on the device, handed to the engine D2D (replicated per GPU)
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
    log(f"[bench] genome {genome_gb:.2f} Gb on device in {time.time() - t0:.1f}s")

    # ---- counters live in a torch tensor so RCCL can sum them in place
    _, n_u64 = eng.counters_device()
    ctr = torch.zeros(n_u64, dtype=torch.int64, device=dev)
    eng.bind_counters(ctr.data_ptr(), n_u64)
    lay = eng.counter_layout()

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(wl: Workload, steps: int, warmup: int) -> dict:
        """W untimed + EXACTLY K timed steps over one resident shard, barrier + synchronize on both
        sides, MAX over ranks; plus the counter reduce timed on its own and per-rank kernel times."""
        def step():
            ctr.zero_()
            for rt, ot, nbytes, n in wl.blocks:
                eng.submit_device(rt.data_ptr(), nbytes, ot.data_ptr(), n)
            reduce_counters(ctr, 2 if use_dist else 1)

        if use_dist:            # communicator set-up is lazy: keep it out of the timed region even with --warmup 0
            reduce_counters(ctr, 2)
            fence()
        for _ in range(warmup):
            step()
        fence()
        eng.kernel_time(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        kernel_ms, n_launch = eng.kernel_time(reset=True)
        res = {"dt_local": dt, "kernel_ms": kernel_ms, "n_launch": n_launch}
        host_ctr = ctr.cpu().numpy().view(np.uint64).copy()   # rank 0: the whole job's tables of the last step
        res["counters"] = host_ctr
        if use_dist:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            mine = torch.tensor([kernel_ms / max(n_launch, 1), float(n_launch), float(wl.n_reads), float(wl.rec_bytes)],
                                dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            res["per_rank"] = [[float(x) for x in t.tolist()] for t in allr]
            # the collective alone: same tensor, same call, nothing else on the stream
            fence()
            R = 50
            t1 = time.perf_counter()
            for _ in range(R):
                reduce_counters(ctr, 2)
            fence()
            res["reduce_ms"] = (time.perf_counter() - t1) / R * 1e3
        else:
            res["per_rank"] = [[kernel_ms / max(n_launch, 1), float(n_launch), float(wl.n_reads), float(wl.rec_bytes)]]
            res["reduce_ms"] = None
        res["dt"] = dt
        return res

    t0 = time.time()
    wl = Workload(synth, torch, dev, stream, cfg, slot0, my_reads)
    log(f"[bench] {args.scaling}: {my_reads / 1e6:.1f} M of {total_reads / 1e6:.1f} M reads ({wl.rec_bytes / 1e9:.2f} GB) "
        f"on device in {len(wl.blocks)} blocks, {time.time() - t0:.1f}s")
    m = measure(wl, args.steps, args.warmup)
    stats = {nm: int(m["counters"][lay["stats"] + i]) for i, nm in enumerate(pkg.ST_NAMES)}

    # algorithmic bytes: record incl. its length word + one u32 index entry +
    # 2*(N+2) reference bytes per read (+ 2k when the k-mer tally is fused in)
    per_read_extra = 4 + 2 * (region_len + 2) + (2 * klen if klen else 0)
    n_blocks = max(len(wl.blocks), 1)
    alg_per_step = wl.rec_bytes + my_reads * per_read_extra
    kernel_ms_per_launch = m["kernel_ms"] / max(m["n_launch"], 1)
    achieved = alg_per_step * args.steps / (m["kernel_ms"] / 1e3) / 1e9 if m["kernel_ms"] > 0 else 0.0
    en = traffic_entry(args.config, args.unsorted, klen, region_len, my_reads / n_blocks)
    traffic = en["hbm_bytes_per_read"] * my_reads / n_blocks if en else None

    lens = f"{cfg.len_min}" + (f"-{cfg.len_max}" if cfg.len_max != cfg.len_min else "")
    out = {
        "metric": "aligned reads/s (whole node) + HBM GB/s fraction, 200M x 150bp BAM",
        "value": total_reads * args.steps / m["dt"],
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": m["dt"] / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.config}: {total_reads / 1e6:g} M x {lens} bp BAM records in total "
                        f"({my_reads / 1e6:g} M on rank 0), HBM-resident, {genome_gb:.2f} Gb {n_contigs}-contig reference "
                        f"(replicated per GPU), end window N={region_len}"
                        + (f", fragkon k={klen}" if klen else "")
                        + (", shuffled order" if args.unsorted else ", coordinate-sorted"),
            "reads_total": total_reads,
            "reads_rank0": my_reads,
            "record_bytes_rank0": wl.rec_bytes,
            "launches_per_step": len(wl.blocks),
            "sharding": f"{args.scaling}: contiguous slot ranges of one stream, {world} rank(s), no data-path collective, "
                        f"one RCCL sum of {n_u64 * 8} B counters per step",
            "value_excludes": "BGZF inflate and PCIe (records are resident in HBM); see e2e for file-to-tables",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": alg_per_step / n_blocks,
            "kernel_ms_per_launch": kernel_ms_per_launch,
            "launches_timed": m["n_launch"],
            "rank": 0,
        },
        "stats_last_step": stats,   # slow_path = records the tiled kernel read from global memory
    }
    if traffic is not None and kernel_ms_per_launch > 0:
        # the same launch priced by what the memory system actually moved (PMC), not by the algorithm
        out["roofline"]["traffic_GBps"] = traffic / (kernel_ms_per_launch / 1e3) / 1e9
        out["roofline"]["frac_of_traffic"] = out["roofline"]["traffic_GBps"] / HBM_PEAK_GBS
        out["roofline"]["traffic_source"] = {k: en.get(k) for k in ("command", "git_sha", "reads_per_launch", "date", "kernel_blobs")}
        # the PMC bytes were measured on THESE kernel sources? (a lookup, not a measurement of this run)
        out["roofline"]["traffic_stale"] = en.get("kernel_blobs") != kernel_blobs()
    if use_dist:
        out["per_rank"] = [{"rank": r, "kernel_ms_per_launch": p[0], "launches": int(p[1]), "reads": int(p[2]),
                            "record_bytes": int(p[3])} for r, p in enumerate(m["per_rank"])]
        out["reduce_ms"] = m["reduce_ms"]
    strong_counters = m["counters"]

    # ---- N > 1: the weak-scaling companion measurement, in the same run ------------------------
    if (world > 1 or args.force_weak_leg) and args.scaling == "strong" and not args.no_weak_leg:
        del wl
        torch.cuda.empty_cache()
        wcfg, wtotal, wslot0, wn = cfg_for("weak")
        wl = Workload(synth, torch, dev, stream, wcfg, wslot0, wn)
        wm = measure(wl, args.steps, args.warmup)
        out["weak_scaling"] = {
            "scaling": "weak", "value": wtotal * args.steps / wm["dt"], "unit": "reads/s",
            "ms_per_step": wm["dt"] / args.steps * 1e3, "reads_total": wtotal, "reads_per_gpu": wn,
            "kernel_ms_per_launch_by_rank": [p[0] for p in wm["per_rank"]], "reduce_ms": wm["reduce_ms"],
        }

    # ---- CPU baseline: the reference itself on this box's host, on a bounded prefix -----------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"], out["parity_check"] = cpu_baseline(pkg, synth, eng, cfg, cd, region_len, klen,
                                                                    min(args.cpu_sample, my_reads), names,
                                                                    args.cpu_cores or min(worker_threads(), 32))
        except Exception as ex:  # the baseline must never sink the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": 1, "kind": "reference",
                                   "sample": f"failed: {ex!r}"}
    # ---- file-to-tables on the named shape (never `value`) ---------------------------------------
    if rank == 0 and world == 1 and not args.no_e2e and not klen and not args.unsorted and args.config != "C1":
        try:
            del wl
            torch.cuda.empty_cache()
            out["e2e"] = e2e_leg(pkg, synth, cd, region_len, args.e2e_reads or total_reads, total_reads, strong_counters, lay)
        except Exception as ex:
            out["e2e"] = {"error": repr(ex)}
    # ---- N > 1: the command itself over all GPUs of the node (one process, one engine per GPU) ----------
    if world > 1 and not args.no_e2e and not klen and not args.unsorted and args.config != "C1":
        flag = Path(tempfile.gettempdir()) / f"pssbam_e2e_done_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
        if rank == 0:
            try:
                out["e2e"] = e2e_multi(pkg, synth, cd, region_len, args.e2e_reads or total_reads, total_reads, strong_counters, lay, world)
            except Exception as ex:
                out["e2e"] = {"error": repr(ex)}
            flag.write_text("done")
        else:   # the other ranks keep their hands off the GPUs meanwhile (no collective pending: a host-side wait)
            t_wait = time.time()
            while not flag.exists() and time.time() - t_wait < 3000:
                time.sleep(0.2)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist and world > 1:
        dist.barrier()
        if rank == 0:
            try:
                (Path(tempfile.gettempdir()) / f"pssbam_e2e_done_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}").unlink()
            except OSError:
                pass
    eng.close()
    if use_dist:
        dist.destroy_process_group()


def e2e_leg(pkg, synth, cd, region_len, n_reads, resident_reads, resident_counters, lay):
    """The whole command a user runs: bin/pss-bam -F ref.fa -B reads.bam on a generated level-1 BGZF
    BAM (htslib's block layout) + FASTA of the benchmarked configuration -- process start, FASTA
    load (f2, reference: init_genome fasta-genome-io.c:221-238), BGZF inflate on the host threads,
    PCIe, kernels, reports (reference equivalent: pss-bam.c:650-805).  The tables it writes are
    compared with the resident-records tally of the same slots when the read counts agree."""
    import numpy as np
    import pssbam_testlib as tl

    threads = worker_threads()
    tmp = Path(tempfile.mkdtemp(prefix="pssbam_e2e_", dir=os.environ.get("PSSBAM_E2E_DIR", os.environ.get("TMPDIR", "/tmp"))))
    try:
        d = dict(cd)
        d["sorted_"] = True
        free = shutil.disk_usage(tmp).free
        est = lambda n: 3.2e9 * (sum(d["contig_lens"]) / 3.1e9) + 60.0 * n   # FASTA + ~60 B of BAM per read, generous
        note = None
        if est(n_reads) > 0.8 * free:
            fit = int(max(1_000_000, (0.8 * free - est(0)) / 60.0))
            note = f"bounded by free disk ({free / 1e9:.0f} GB in {tmp.parent}): {fit} of {n_reads} reads"
            n_reads = min(n_reads, fit)
        d["n_reads"] = resident_reads   # the stream's shape; a prefix of it goes into the file
        cfg = synth.make_cfg(**d)
        fa, bam = tmp / "ref.fa", tmp / "reads.bam"
        t = time.perf_counter()
        synth.fasta_host(cfg, fa, threads=threads)
        t_fa = time.perf_counter() - t
        t = time.perf_counter()
        synth.bam_file_host(cfg, 0, n_reads, bam, level=1, threads=threads)
        t_bam = time.perf_counter() - t
        def run_cli(extra_env, prefix):
            env = {**os.environ, "PSSBAM_STATS": "1", **extra_env}
            time.sleep(1.0)   # the previous run's teardown (a forked worker, 30 GB of device buffers) is out of the way, as for a user's command
            t = time.perf_counter()
            pr = subprocess.run([str(pkg.PKG_DIR / "bin" / "pss-bam"), "-F", str(fa), "-B", str(bam), "-o", str(tmp / prefix),
                                 "-r", str(region_len)], capture_output=True, text=True, env=env, timeout=1500)
            return pr, time.perf_counter() - t

        # the same command with the inflate kept on the host threads (round 1's feed), for comparison
        pr_h, wall_h = run_cli({"PSSBAM_DEVICE_INFLATE": "0"}, "out_host")
        # five runs of the command as a user would type it; the MEDIAN one is reported (its stages too): the feed's share
        # of the wall clock moves with what the box's other tenants do to the host side (about one run in eight has its
        # feeding thread held up for 0.1 s; with three runs the median was one of those every other time)
        runs = sorted((run_cli({}, "out") for _ in range(5)), key=lambda r: r[1])
        for r in runs:
            if r[0].returncode != 0:
                return {"error": r[0].stderr[-1500:]}
        pr, wall = runs[2]
        # the same with the teardown in the foreground (by default the process the caller started returns when the
        # reports are written and a forked worker is dismantled behind it: host/frontend.c frontend_detach_start)
        pr_fg, wall_fg = run_cli({"PSSBAM_DETACH_EXIT": "0"}, "out_fg")
        grab = lambda pat: (lambda mm: mm.group(1) if mm else None)(re.search(pat, pr.stderr))
        tally_s = float(grab(r"total_s=([\d.]+)") or 0)
        stages = {}
        for line_pat in (r"phases: (.*) s\n", r"reader thread: (.*) s\n", r"genome: (.*) s\n", r"device: (.*) s\n"):
            txt = grab(line_pat)
            if txt:
                for k, v in re.findall(r"([A-Za-z+()\-_0-9 ]+?) ([\d.]+)(?: |$)", txt):
                    stages[k.strip()] = float(v)
        busy = re.search(r"gpu busy: inflate\+crc\+index ([\d.]+) tally ([\d.]+) s", pr.stderr)
        gpu_busy_s = (float(busy.group(1)) + float(busy.group(2)) + 0.006) if busy else None   # + genome encode / pack kernels
        got_f, got_r = tl.parse_counts_text((tmp / "out.pss.counts.txt").read_text())
        check = None
        if n_reads == resident_reads:
            rows = lay["rows"]
            want_f = resident_counters[lay["fwd"]:lay["fwd"] + rows * 16].reshape(rows, 16)
            want_r = resident_counters[lay["rev"]:lay["rev"] + rows * 16].reshape(rows, 16)
            ok = bool(np.array_equal(got_f, want_f) and np.array_equal(got_r, want_r))
            check = "tables identical to the HBM-resident tally of the same slots" if ok else "MISMATCH vs the resident tally"
        # The named configurations have constant QUAL, which DEFLATE turns into one long match per read.
        # A second, smaller file with 40-level quality strings at level 6 shows the same command on what a
        # sequencer's BAM looks like to the feed: ~4x the compressed bytes per read, a literal-heavy stream.
        bam_bytes = bam.stat().st_size
        # SURVEY 8d asks for the uncompressed-BGZF case separately: level 0, every record byte crosses PCIe
        level0 = None
        n_l0 = min(n_reads, int(os.environ.get("PSSBAM_E2E_LEVEL0_READS", "50000000")))
        if n_l0 > 0 and shutil.disk_usage(tmp).free > 300.0 * n_l0 + bam_bytes:
            try:
                bam.unlink(missing_ok=True)
                t = time.perf_counter()
                synth.bam_file_host(cfg, 0, n_l0, bam, level=0, threads=threads)
                t_bam0 = time.perf_counter() - t
                r0 = sorted((run_cli({}, "l0") for _ in range(3)), key=lambda r: r[1])
                if all(r[0].returncode == 0 for r in r0):
                    level0 = {"what": "the same command on a level-0 (stored) BGZF BAM: PCIe carries every record byte",
                              "reads": n_l0, "bam_bytes": bam.stat().st_size, "wall_s": r0[1][1], "reads_per_s": n_l0 / r0[1][1],
                              "wall_s_runs": [r[1] for r in r0], "wall_s_is": "median of 3 runs", "workload_gen_s": t_bam0,
                              "device_feed": (lambda mm: mm.group(1) if mm else None)(re.search(r"device feed: (.*)\n", r0[1][0].stderr))}
                else:
                    level0 = {"error": r0[0][0].stderr[-500:]}
            except Exception as ex:
                level0 = {"error": repr(ex)}
        real = None
        n_real = min(n_reads, int(os.environ.get("PSSBAM_E2E_REAL_READS", "50000000")))
        if n_real > 0:
            try:
                bam.unlink(missing_ok=True)
                t = time.perf_counter()
                synth.bam_file_host(cfg, 0, n_real, bam, level=6, threads=threads, quals="full")
                t_bam2 = time.perf_counter() - t
                r_h = run_cli({"PSSBAM_DEVICE_INFLATE": "0"}, "real_host")
                rr = sorted((run_cli({}, "real") for _ in range(3)), key=lambda r: r[1])
                if all(r[0].returncode == 0 for r in rr) and r_h[0].returncode == 0:
                    same = all(np.array_equal(a, b) for a, b in zip(tl.parse_counts_text((tmp / "real_host.pss.counts.txt").read_text()),
                                                                     tl.parse_counts_text((tmp / "real.pss.counts.txt").read_text())))
                    mm = re.search(r"device feed: (.*)\n", rr[1][0].stderr)
                    real = {"what": "the same command on a BAM with 40-level quality strings, deflate level 6 (synth.bam_file_host quals='full')",
                            "reads": n_real, "bam_bytes": bam.stat().st_size, "wall_s": rr[1][1], "reads_per_s": n_real / rr[1][1],
                            "wall_s_runs": [r[1] for r in rr], "wall_s_is": "median of 3 runs", "device_feed": mm.group(1) if mm else None,
                            "engine_feed": (lambda m2: m2.group(1) if m2 else None)(re.search(r"engine feed: (.*)\n", rr[1][0].stderr)),
                            "device_feed_note": "the kernel seconds quoted in device_feed are launch durations; on this file the feed is PCIe-bound and "
                                                "flushes partial rounds when the device runs dry (engine_feed: share of the lanes filled), so "
                                                "bytes / kernel seconds understates the kernel (tools/inflate_bench.py: 134 GB/s at whole rounds)",
                            "host_inflate_run": {"wall_s": r_h[1], "reads_per_s": n_real / r_h[1], "tables_identical": bool(same)},
                            "workload_gen_s": t_bam2}
                else:
                    real = {"error": (rr[0][0].stderr or r_h[0].stderr)[-500:]}
            except Exception as ex:   # the headline e2e object must not depend on this one
                real = {"error": repr(ex)}
        return {
            "command": "bin/pss-bam -F ref.fa -B reads.bam -o out -r %d" % region_len,
            "reads": n_reads, "bam_bytes": bam_bytes, "fasta_bytes": fa.stat().st_size,
            "deflate_level": 1, "block_layout": "htslib", "host_cpus_effective": effective_cpus(),
            "wall_s": wall, "reads_per_s": n_reads / wall, "wall_s_runs": [r[1] for r in runs], "wall_s_is": "median of 5 runs",
            "gpu_busy_s": gpu_busy_s, "gpu_busy_frac": gpu_busy_s / wall if gpu_busy_s else None,
            "gpu_busy_is": "kernel time by HIP events: the union of the super-batches' inflate + CRC + record-index intervals (consecutive "
                           "inflate launches overlap on purpose: a sum would count that twice) + every tally launch + 6 ms genome encode / "
                           "4-bit pack, of the median run, over its wall seconds",
            "wall_s_foreground_exit": wall_fg if pr_fg.returncode == 0 else None,
            "wall_s_note": "wall_s = until the process the caller started returns (reports written; a forked worker is torn down "
                           "behind it); wall_s_foreground_exit = PSSBAM_DETACH_EXIT=0, one process, teardown included",
            "early_feed": grab(r"early feed \(helper thread[^:]*\): (.*)\n"),
            "tally_phase_s": tally_s, "reads_per_s_tally_phase": n_reads / tally_s if tally_s else None,
            "fasta_load_s": stages.get("fasta load"), "stages_s": stages,
            "feed": "device inflate" if "device feed:" in pr.stderr and "falling back" not in pr.stderr else "host inflate",
            "device_feed": grab(r"device feed: (.*)\n"),
            "feed_thread": grab(r"device feed, this thread: (.*)\n"),
            "engine_feed": grab(r"engine feed: (.*)\n"),
            "host_inflate_run": ({"wall_s": wall_h, "reads_per_s": n_reads / wall_h,
                                  "tables_identical": bool(all(np.array_equal(a, b) for a, b in zip(tl.parse_counts_text((tmp / "out_host.pss.counts.txt").read_text()), tl.parse_counts_text((tmp / "out.pss.counts.txt").read_text()))))}
                                 if pr_h.returncode == 0 else {"error": pr_h.stderr[-500:]}),
            "tables_check": check, "note": note,
            "workload_gen_s": {"fasta": t_fa, "bam": t_bam},
            "real_quals": real, "level0": level0,
        }
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


E2E_MULTI_TIMEOUT_S = 240   # one run of the n-GPU command (seconds on one GPU with the same file)


def e2e_multi(pkg, synth, cd, region_len, n_reads, resident_reads, resident_counters, lay, world):
    """N > 1: `PSSBAM_NGPU=N bin/pss-bam` on the generated level-1 BAM + FASTA of the benchmarked shape -- ONE process
    driving one engine per GPU: the compressed file is dealt in runs of whole super-batches, every GPU inflates, indexes
    and tallies its runs, pssbam_reduce_counters sums the counter blocks with one grouped ncclReduce over xGMI
    (csrc/engine.hip).  Run on rank 0 while the other ranks wait on the host; tables checked against the reduced
    tally of the distributed measurement when the read counts agree."""
    import numpy as np
    import pssbam_testlib as tl

    threads = worker_threads()
    tmp = Path(tempfile.mkdtemp(prefix="pssbam_e2e_", dir=os.environ.get("PSSBAM_E2E_DIR", os.environ.get("TMPDIR", "/tmp"))))
    try:
        d = dict(cd)
        d["sorted_"] = True
        d["n_reads"] = resident_reads
        free = shutil.disk_usage(tmp).free
        est = 3.2e9 * (sum(d["contig_lens"]) / 3.1e9) + 60.0 * n_reads
        note = None
        if est > 0.8 * free:
            n_reads = int(max(1_000_000, (0.8 * free - 3.3e9) / 60.0))
            note = f"bounded by free disk: {n_reads} reads"
        cfg = synth.make_cfg(**d)
        fa, bam = tmp / "ref.fa", tmp / "reads.bam"
        synth.fasta_host(cfg, fa, threads=threads)
        synth.bam_file_host(cfg, 0, n_reads, bam, level=1, threads=threads)

        def run_cli(extra):
            # (a session of its own, killed as a group when it overruns: the command forks a worker that would outlive
            #  a kill of the process this one started -- and this n-GPU path has never run on n physical GPUs)
            time.sleep(1.0)   # (as in e2e_leg)
            t = time.perf_counter()
            po = subprocess.Popen([str(pkg.PKG_DIR / "bin" / "pss-bam"), "-F", str(fa), "-B", str(bam), "-o", str(tmp / "out"), "-r", str(region_len)],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True,
                                  env={**os.environ, "PSSBAM_STATS": "1", "PSSBAM_NGPU": str(world), **extra})
            try:
                so, se = po.communicate(timeout=E2E_MULTI_TIMEOUT_S)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(po.pid, signal.SIGKILL)
                except OSError:
                    pass
                so, se = po.communicate()
                return subprocess.CompletedProcess(po.args, -9, so, (se or "") + f"\n[bench] killed after {E2E_MULTI_TIMEOUT_S} s"), time.perf_counter() - t
            return subprocess.CompletedProcess(po.args, po.returncode, so, se), time.perf_counter() - t
        runs = []
        for _ in range(3):
            runs.append(run_cli({}))
            if runs[-1][0].returncode != 0:   # (no second try of a run that failed or hung)
                return {"error": runs[-1][0].stderr[-1500:]}
        runs.sort(key=lambda r: r[1])
        pr, wall = runs[1]
        grab = lambda pat: (lambda mm: mm.group(1) if mm else None)(re.search(pat, pr.stderr))
        got_f, got_r = tl.parse_counts_text((tmp / "out.pss.counts.txt").read_text())
        check = None
        if n_reads == resident_reads:
            rows = lay["rows"]
            ok = bool(np.array_equal(got_f, resident_counters[lay["fwd"]:lay["fwd"] + rows * 16].reshape(rows, 16)) and
                      np.array_equal(got_r, resident_counters[lay["rev"]:lay["rev"] + rows * 16].reshape(rows, 16)))
            check = "tables identical to the reduced tally of the distributed measurement" if ok else "MISMATCH vs the reduced tally"
        one, wall_one = run_cli({"PSSBAM_NGPU": "1"})
        per = grab(r"submits per engine: (.*)\n")
        return {"command": f"PSSBAM_NGPU={world} bin/pss-bam -F ref.fa -B reads.bam -o out -r {region_len}", "reads": n_reads,
                "bam_bytes": bam.stat().st_size, "gpus": int(grab(r"gpus=(\d+)") or 0), "wall_s": wall, "reads_per_s": n_reads / wall,
                "wall_s_runs": [r[1] for r in runs], "wall_s_is": "median of 3 runs",
                "submits_per_engine": [int(x) for x in per.split()] if per else None, "device_feed": grab(r"device feed: (.*)\n"),
                "feed": "device inflate" if "device feed:" in pr.stderr and "falling back" not in pr.stderr else "host inflate",
                "gpu_busy": grab(r"gpu busy: (.*)\n"), "early_feed": grab(r"early feed \(helper thread[^:]*\): (.*)\n"),
                "one_gpu_same_file": {"wall_s": wall_one, "reads_per_s": n_reads / wall_one} if one.returncode == 0 else {"error": one.stderr[-300:]},
                "tables_check": check, "note": note}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(pkg, synth, eng, cfg, cd, region_len, klen, n_sample, names, cpu_cores=1):
    """Times oracle/_ref/pss-bam.O2 (unmodified reference, compiled in the build container)
    on the first n_sample reads of the SORTED stream, fed as SAM text (no inflate cost charged
    to it), and checks the engine's tables for the same reads against the reference's."""
    import numpy as np
    import pssbam_testlib as tl

    tmp = Path(tempfile.mkdtemp(prefix="pssbam_cpu_"))
    cds = dict(cd)
    cds["sorted_"] = True
    cds["n_reads"] = int(cfg.n_reads)
    scfg = synth.make_cfg(**cds)
    threads = worker_threads()
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
        try:   # bounded: <= 500 k reads per process, so the default run stays within minutes
            out["all_cores"] = cpu_all_cores(pkg, synth, scfg, region_len, names, tmp, o, cpu_cores,
                                             min(500_000, max(100_000, n_sample // 4), max(1, int(scfg.n_reads) // cpu_cores)))
        except Exception as ex:
            out["all_cores"] = {"error": repr(ex)}
    shutil.rmtree(tmp, ignore_errors=True)
    return (out, parity)


def cpu_all_cores(pkg, synth, scfg, region_len, names, tmp, o, procs, per_shard):
    """one unmodified reference process per core on disjoint consecutive shards of the
    sorted stream, tables summed (tallies are additive) and checked against the engine's tables
    for the same reads.  Load time (every process parses the FASTA) is measured by the same
    number of concurrent processes on an empty SAM and subtracted.  `cores` in the result is what
    the processes could actually use: min(process count, cgroup CPU quota)."""
    import numpy as np
    import pssbam_testlib as tl
    from concurrent.futures import ThreadPoolExecutor

    threads = worker_threads()
    total = procs * per_shard
    recs, offs = synth.records_host(scfg, total - 1, 1, threads=1)   # the last read tells how many contigs are needed
    last_contig = int(np.frombuffer(recs[4:8].tobytes(), dtype="<i4")[0])
    fa, empty = tmp / "ref_all.fa", tmp / "empty_all.sam"
    synth.fasta_host(scfg, fa, 0, last_contig + 1, threads=threads)
    synth.sam_host(scfg, 0, 0, empty)
    with ThreadPoolExecutor(min(threads, 32)) as ex:   # the writers release the GIL
        list(ex.map(lambda k: synth.sam_host(scfg, k * per_shard, per_shard, tmp / f"shard{k}.sam"), range(procs)))

    variant = os.environ.get("PSSBAM_REF_VARIANT", "pss-bam.O2")

    def wave(inputs, tag):
        t = time.perf_counter()
        with ThreadPoolExecutor(procs) as ex:
            res = list(ex.map(lambda kv: tl.run_ref_pss(fa, kv[1], tmp / f"{tag}{kv[0]}", o, variant=variant,
                                                        timeout=3000)[:2], enumerate(inputs)))
        return time.perf_counter() - t, res

    t_load, _ = wave([empty] * procs, "l")
    t_full, res = wave([tmp / f"shard{k}.sam" for k in range(procs)], "s")
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
    eff = effective_cpus()
    return {"value": total / max(t_full - t_load, 1e-9), "unit": "reads/s", "cores": min(float(procs), eff),
            "processes": procs, "host_cpus_effective": eff, "host_cpus_logical": os.cpu_count(),
            "sample": f"{procs} concurrent oracle/_ref/{variant} processes x {per_shard} reads each (consecutive shards of "
                      f"the sorted stream, FASTA of contigs 0..{last_contig}) on {eff:g} effective CPUs (cgroup cpu.max); "
                      f"wall {t_full:.1f}s minus {t_load:.1f}s for the same {procs} processes on an empty SAM",
            "parity_check": "summed tables bit-exact vs the engine" if ok else "MISMATCH vs the engine"}


if __name__ == "__main__":
    main()
