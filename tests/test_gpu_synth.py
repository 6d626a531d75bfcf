"""The synthetic workloads of bench.py (pss-bam_amd/csrc/synth_model.h) on the GPU:
* the device generator and its host twin produce identical bytes,
* the HIP path on device-generated records == the oracle on the model's independent SAM/FASTA
  text twin (every named configuration, scaled down),
* size-independent properties at larger sizes: record-order invariance (sorted vs shuffled
  stream), additivity over shards."""
import ctypes as C

import numpy as np
import pytest
import torch

import __graft_entry__ as ge
import pssbam_testlib as tl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    pkg = ge.load_pkg()
    from pss_bam_amd import synth
    return pkg, synth


def _device_workload(pkg, synth, cfg, slot0, n, dev):
    """genome + records generated on the device; returns tensors and host copies"""
    S = synth.lib()
    stream = torch.cuda.current_stream().cuda_stream
    contigs = []
    for k in range(int(cfg.n_contigs)):
        ln = int(cfg.contig_len[k])
        t = torch.zeros(ln + 64, dtype=torch.uint8, device=dev)
        assert S.synth_genome_device(C.byref(cfg), k, t.data_ptr(), ln, stream) == 0
        contigs.append(t)
    sizes = synth.sizes_host(cfg, slot0, n)
    offs = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(sizes, out=offs[1:])
    ot = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).to(dev)
    rt = torch.zeros(int(offs[-1]) + 64, dtype=torch.uint8, device=dev)
    assert S.synth_records_device(C.byref(cfg), slot0, n, ot.data_ptr(), rt.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    return contigs, rt, ot, int(offs[-1])


def _run_engine_device(pkg, synth, cfg, contigs, rt, ot, nbytes, n, pss=None, kmer=None, kernel=0):
    names = [synth.contig_name(cfg, k) for k in range(int(cfg.n_contigs))]
    eng = pkg.Engine(pss=pss, kmer=kmer, kernel=kernel)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_genome_device([(names[k], contigs[k].data_ptr(), int(cfg.contig_len[k])) for k in range(len(names))])
    eng.set_references(names)
    eng.submit_device(rt.data_ptr(), nbytes, ot.data_ptr(), n)
    out = eng.finish()
    eng.close()
    return out


CASES = [
    ("C1", dict(n_reads=20000), {}),
    ("C2", dict(n_reads=30000), dict(scale_genome=0.0004)),
    ("C2", dict(n_reads=30000, sorted_=False), dict(scale_genome=0.0004)),
    ("C4", dict(n_reads=40000), dict(scale_genome=0.0004)),
    ("C5", dict(n_reads=25000), dict(scale_genome=0.0004)),
]


@pytest.mark.parametrize("name,over,kw", CASES, ids=[f"{c[0]}-{i}" for i, c in enumerate(CASES)])
def test_generated_workload_matches_oracle(env, oracle, tmp_path, name, over, kw):
    pkg, synth = env
    dev = torch.device("cuda", 0)
    d = synth.config(name, **kw)
    d.update(over)
    region_len, klen = d.pop("region_len"), d.pop("klen", 4)
    cfg = synth.make_cfg(**d)
    n = int(cfg.n_reads)
    contigs, rt, ot, nbytes = _device_workload(pkg, synth, cfg, 0, n, dev)
    # device bytes == host twin bytes
    h_recs, h_offs = synth.records_host(cfg, 0, n)
    assert np.array_equal(rt[:nbytes].cpu().numpy(), h_recs)
    for k in range(int(cfg.n_contigs)):
        assert np.array_equal(contigs[k][:int(cfg.contig_len[k])].cpu().numpy(), synth.genome_host(cfg, k))
    # oracle on the text twin
    fa, sam = tmp_path / "g.fa", tmp_path / "a.sam"
    synth.fasta_host(cfg, fa)
    synth.sam_host(cfg, 0, n, sam)
    g = oracle.load_genome(fa)
    po, ko = tl.PssOpts(region_len=region_len), tl.FkOpts(klen=klen)
    wf, wr, st = oracle.pss(g, sam, po)
    w5, w3, _ = oracle.fragkon(g, sam, ko)
    if name == "C4":
        po20 = tl.PssOpts(region_len=region_len, min_mq=20)
        wf20, wr20, _ = oracle.pss(g, sam, po20)
    oracle.free_genome(g)
    assert st[tl.ST_OK] > n // 4
    for kern in (pkg.KERNEL_SIMPLE, pkg.KERNEL_TILED):
        got = _run_engine_device(pkg, synth, cfg, contigs, rt, ot, nbytes, n, pss=dict(region_len=region_len),
                                 kmer=dict(klen=klen), kernel=kern)
        assert np.array_equal(got.fwd, wf) and np.array_equal(got.rev, wr)
        assert np.array_equal(got.k5, w5.astype(np.uint64)) and np.array_equal(got.k3, w3.astype(np.uint64))
        assert got.stats["pss_ok"] == st[tl.ST_OK] and got.stats["records"] == n
    if name == "C4":
        got = _run_engine_device(pkg, synth, cfg, contigs, rt, ot, nbytes, n,
                                 pss=dict(region_len=region_len, min_mq=20))
        assert np.array_equal(got.fwd, wf20) and np.array_equal(got.rev, wr20)


def test_order_invariance_and_additivity_large(env):
    """2 M reads, no oracle: (a) the shuffled stream is a permutation of the sorted one, so the
    tables must be identical; (b) tallying two half-shards separately and adding == one pass;
    (c) both kernels agree."""
    pkg, synth = env
    dev = torch.device("cuda", 0)
    n = 2_000_000
    tabs = {}
    for srt in (True, False):
        d = synth.config("C2", n_reads=n, sorted_=srt, scale_genome=0.02)
        region_len = d.pop("region_len")
        cfg = synth.make_cfg(**d)
        contigs, rt, ot, nbytes = _device_workload(pkg, synth, cfg, 0, n, dev)
        for kern in (pkg.KERNEL_SIMPLE, pkg.KERNEL_TILED):
            tabs[(srt, kern)] = _run_engine_device(pkg, synth, cfg, contigs, rt, ot, nbytes, n,
                                                   pss=dict(region_len=region_len), kmer=dict(klen=4), kernel=kern)
        if srt:
            # additivity: two engines on the two halves (what two ranks would do), summed
            half = n // 2
            rec_bytes = nbytes // n
            parts = []
            for a, b in ((0, half), (half, n)):
                sub_o = torch.arange(0, (b - a + 1) * rec_bytes, rec_bytes, dtype=torch.int64, device=dev).to(torch.int32)
                sub_r = rt[a * rec_bytes:]
                parts.append(_run_engine_device(pkg, synth, cfg, contigs, sub_r, sub_o, (b - a) * rec_bytes, b - a,
                                                pss=dict(region_len=region_len), kmer=dict(klen=4)))
            ref = tabs[(True, pkg.KERNEL_TILED)]
            assert np.array_equal(parts[0].fwd + parts[1].fwd, ref.fwd)
            assert np.array_equal(parts[0].rev + parts[1].rev, ref.rev)
            assert np.array_equal(parts[0].k5 + parts[1].k5, ref.k5)
        del contigs, rt, ot
    base = tabs[(True, pkg.KERNEL_TILED)]
    assert base.fwd.sum() > n  # something was tallied
    for key, t in tabs.items():
        assert np.array_equal(t.fwd, base.fwd) and np.array_equal(t.rev, base.rev), key
        assert np.array_equal(t.k5, base.k5) and np.array_equal(t.k3, base.k3), key


def test_full_size_baseline_config_properties(env):
    """BASELINE.json's metric configuration at its FULL size (C3: 200 M x 150 bp reads, 3.1 Gb
    24-contig reference, N=25, k=4), where no CPU oracle finishes in reasonable time.  Checked
    through properties that do not depend on the size:
      (a) order invariance: the shuffled stream is a permutation of the sorted one -> same tables;
      (b) additivity: four contiguous 50 M-read shards tallied by four engines (what four ranks
          do) and summed == one engine over everything;
      (c) the two kernels (tiled production kernel, generic lane-per-read kernel) agree;
      (d) bookkeeping: every record is accounted for exactly once in the status counters, each
          table row holds at most one count per tallied read, and the two context rows of a
          table agree with each other up to non-ACGT reference bases (<= 1 %)."""
    pkg, synth = env
    dev = torch.device("cuda", 0)
    S = synth.lib()
    stream = torch.cuda.current_stream().cuda_stream
    n_total, n_block, n_shards = 200_000_000, 12_500_000, 4
    cfgs = {}
    for srt in (True, False):
        d = synth.config("C3", sorted_=srt)
        region_len = d.pop("region_len")
        d.pop("klen", None)
        assert d["n_reads"] == n_total
        cfgs[srt] = synth.make_cfg(**d)
    cfg = cfgs[True]
    names = [synth.contig_name(cfg, k) for k in range(int(cfg.n_contigs))]
    contigs = []
    for k in range(int(cfg.n_contigs)):
        ln = int(cfg.contig_len[k])
        t = torch.empty(ln + 64, dtype=torch.uint8, device=dev)
        assert S.synth_genome_device(C.byref(cfg), k, t.data_ptr(), ln, stream) == 0
        contigs.append(t)

    def engine(kernel=0):
        e = pkg.Engine(pss=dict(region_len=region_len), kmer=dict(klen=4), kernel=kernel)
        e.set_stream(stream)
        e.set_genome_device([(names[k], contigs[k].data_ptr(), int(cfg.contig_len[k])) for k in range(len(names))])
        e.set_references(names)
        return e

    e_sorted, e_shuffled, e_simple = engine(pkg.KERNEL_TILED), engine(pkg.KERNEL_TILED), engine(pkg.KERNEL_SIMPLE)
    e_shard = [engine() for _ in range(n_shards)]
    del contigs
    rec_bytes = int(synth.sizes_host(cfg, 0, 1)[0])
    rt = torch.empty(n_block * rec_bytes + 64, dtype=torch.uint8, device=dev)
    ot = torch.empty(n_block + 1, dtype=torch.int32, device=dev)
    assert S.synth_offsets_linear_device(ot.data_ptr(), n_block + 1, rec_bytes, stream) == 0
    for srt in (True, False):
        for a in range(0, n_total, n_block):
            assert S.synth_records_device(C.byref(cfgs[srt]), a, n_block, ot.data_ptr(), rt.data_ptr(), stream) == 0
            targets = [e_sorted, e_simple, e_shard[a // (n_total // n_shards)]] if srt else [e_shuffled]
            for e in targets:
                e.submit_device(rt.data_ptr(), n_block * rec_bytes, ot.data_ptr(), n_block)
            torch.cuda.synchronize()   # the block buffer is regenerated next
    base = e_sorted.finish()
    shuf, simple = e_shuffled.finish(), e_simple.finish()
    parts = [e.finish() for e in e_shard]
    for e in [e_sorted, e_shuffled, e_simple] + e_shard:
        e.close()

    for other, what in ((shuf, "shuffled order"), (simple, "generic kernel")):
        for f in ("fwd", "rev", "k5", "k3"):
            assert np.array_equal(getattr(other, f), getattr(base, f)), (what, f)
    for f in ("fwd", "rev", "k5", "k3"):
        assert np.array_equal(sum(getattr(p, f).astype(np.uint64) for p in parts), getattr(base, f).astype(np.uint64)), f
    st = base.stats
    assert st["records"] == n_total and sum(p.stats["records"] for p in parts) == n_total
    assert st["pss_ok"] + st["pss_filtered"] + st["no_contig"] + st["parse_skip"] == n_total
    assert st["slow_path"] == 0 and shuf.stats == st
    assert 0.98 * n_total < st["pss_ok"] <= n_total
    for tab in (base.fwd, base.rev):
        rows = tab.reshape(region_len + 2, 16).sum(axis=1)
        assert rows.max() <= st["pss_ok"] and rows.min() >= 0.97 * st["pss_ok"]
        assert abs(int(rows[0]) - int(rows[1])) <= 0.01 * st["pss_ok"]
    assert int(base.k5.sum()) <= st["kmer_ok"] + st["kmer_fail"] and int(base.k5.sum()) >= 0.97 * n_total
