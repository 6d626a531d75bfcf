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
