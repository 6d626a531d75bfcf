"""ThreadSanitizer and AddressSanitizer + UBSan builds of the GPU-free host code
(`make -C pss-bam_amd sanitize`): the three-stage, multi-threaded BGZF/BAM reader
(host/bam_reader.c: producer + inflate pool + indexer, locks and atomics), the hand-written
DEFLATE decoder with its unchecked fast loop (host/inflate_fast.c), the parallel FASTA loader and
the threaded SAM-text reader.  Run over the golden fixtures and over a generated multi-batch BAM
whose BGZF blocks are cut regardless of records, with 70 KB batches and 8 inflate threads: the
sanitizers must stay silent and the output must equal the plain build's.  CPU only (GPU ASan /
XNACK are not available on the pool)."""
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

import __graft_entry__ as ge
import pssbam_testlib as tl

GOLD = Path(__file__).resolve().parent / "golden"
STRESS_ENV = {"PSSBAM_BATCH_BYTES": "70000", "PSSBAM_INFLATE_THREADS": "8",
              "TSAN_OPTIONS": "halt_on_error=1 exitcode=66", "ASAN_OPTIONS": "detect_leaks=1 exitcode=67",
              "UBSAN_OPTIONS": "print_stacktrace=1 halt_on_error=1"}


@pytest.fixture(scope="module")
def san():
    ge.build()
    pkg = ge.load_pkg()
    pr = subprocess.run(["make", "-s", "-C", str(pkg.PKG_DIR), "sanitize"], capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr[-3000:]
    b = pkg.PKG_DIR / "bin"
    probe = subprocess.run([str(b / "san" / "hostcheck.tsan")], capture_output=True, text=True)
    if "unexpected memory mapping" in probe.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow in this environment")
    return b


def _run(exe, args, env_extra=None):
    pr = subprocess.run([str(exe)] + [str(a) for a in args], capture_output=True, text=True, timeout=900,
                        env={**os.environ, **STRESS_ENV, **(env_extra or {})})
    report = [ln for ln in pr.stderr.splitlines() if "Sanitizer" in ln or "runtime error" in ln]
    assert pr.returncode == 0 and not report, f"{exe.name} {args}: rc {pr.returncode}\n" + pr.stderr[-3000:]
    return pr.stdout


@pytest.fixture(scope="module")
def stress_files(tmp_path_factory):
    """a ragged multi-batch BAM (blocks cut every 3000-9000 bytes regardless of records, level 1),
    its SAM text (plain + gzip) and FASTA (plain + gzip)"""
    import gzip
    d = tmp_path_factory.mktemp("san_inputs")
    contigs, refs, recs = tl.fuzz_dataset(7700, 6000, contig_lens=(40000, 9000, 1200), with_rg=True)
    tl.write_fasta(d / "g.fa", contigs)
    (d / "g.fa.gz").write_bytes(gzip.compress((d / "g.fa").read_bytes()))
    tl.write_sam(d / "a.sam", refs, recs)
    (d / "a.sam.gz").write_bytes(gzip.compress((d / "a.sam").read_bytes()))
    tl.write_bam(d / "ragged.bam", refs, recs, level=1, rng=np.random.default_rng(5), block=6000)
    tl.write_bam(d / "whole.bam", refs, recs, level=6)
    # FASTA files above the parallel .gz loaders' 1 MiB threshold (host/genome_load.c: fa_bgzf_worker /
    # load_bgzf_parallel / load_gzip_whole), in the three shapes they tell apart: bgzip blocks, one gzip
    # member, two gzip members
    rng = np.random.default_rng(99)
    big = [(f"big{k}", tl.random_contig(rng, n)) for k, n in enumerate((1_300_000, 700_000, 40_000, 300_000))]
    tl.write_fasta(d / "big.fa", big)
    text = (d / "big.fa").read_bytes()
    assert len(text) > 2 << 20
    (d / "big.bgzf.fa.gz").write_bytes(b"".join(tl.bgzf_block(text[i:i + 0xFF00], 6) for i in range(0, len(text), 0xFF00)) + tl.BGZF_EOF)
    (d / "big.one.fa.gz").write_bytes(gzip.compress(text, 6))
    cut = text.index(b"\n", len(text) // 2) + 1
    (d / "big.two.fa.gz").write_bytes(gzip.compress(text[:cut], 6) + gzip.compress(text[cut:], 1))
    return d


@pytest.mark.parametrize("flavour", ["tsan", "asan"])
def test_host_readers_under_sanitizers(san, stress_files, flavour):
    plain_hc, plain_b2s = san / "hostcheck", san / "bam2sam"
    hc, b2s = san / "san" / f"hostcheck.{flavour}", san / "san" / f"bam2sam.{flavour}"
    d = stress_files
    cases = [["-f", GOLD / "setA.fa", "-a", GOLD / "setA.bam"], ["-f", GOLD / "setB.fa", "-a", GOLD / "setB.sam"],
             ["-f", d / "g.fa", "-a", d / "ragged.bam"], ["-f", d / "g.fa.gz", "-a", d / "whole.bam"],
             ["-a", d / "a.sam"], ["-a", d / "a.sam.gz"]]
    # the threaded .gz FASTA loaders: same genome digest as the plain file, eight parser / inflate threads
    want_big = _run(plain_hc, ["-f", d / "big.fa"])
    for gz in ("big.bgzf.fa.gz", "big.one.fa.gz", "big.two.fa.gz"):
        assert _run(plain_hc, ["-f", d / gz]) == want_big, gz
        assert _run(hc, ["-f", d / gz], {"PSSBAM_FASTA_THREADS": "8"}) == want_big, gz
    assert _run(hc, ["-f", d / "big.fa"], {"PSSBAM_FASTA_THREADS": "8"}) == want_big
    for args in cases:
        want = _run(plain_hc, args)
        assert _run(hc, args) == want, args
        # and with the default (large) batches / thread count
        assert _run(hc, args, {"PSSBAM_BATCH_BYTES": "0", "PSSBAM_INFLATE_THREADS": "3"}) == want, args
    for bam in (GOLD / "setA.bam", GOLD / "setB.bam", d / "ragged.bam"):
        want = _run(plain_b2s, [bam])
        assert want.count("\n") > 100
        assert _run(b2s, [bam]) == want
        assert _run(b2s, ["-r", "grpA", bam]) == _run(plain_b2s, ["-r", "grpA", bam])


def test_ragged_bam_digest_is_independent_of_batching(san, stress_files):
    """the reader hands out the same record stream whatever the batch size / thread count
    (digest over all record bytes; the plain build, many geometries)"""
    hc = san / "hostcheck"
    outs = set()
    for bb, th in (("70000", "8"), ("262144", "1"), ("1000000", "5"), ("0", "0")):
        out = _run(hc, ["-a", stress_files / "ragged.bam"], {"PSSBAM_BATCH_BYTES": bb, "PSSBAM_INFLATE_THREADS": th})
        outs.add(out)
    assert len(outs) == 1, outs
    assert _run(hc, ["-a", stress_files / "whole.bam"]) in outs      # same records, different BGZF layout


def test_inflate_loop_schedule_in_the_compiled_kernel():
    """the one-wait-per-step inflate loop only pays while the compiler keeps its hands off the stretch between the
    LDS-DMA requests and the hand-written wait: tools/isa_inflate_check.py asserts that on hipcc's listing (a hipcc
    update that changes it should fail here, not show up as a slower kernel)"""
    import shutil
    import subprocess
    import sys
    from pathlib import Path
    if not Path("/opt/rocm/bin/hipcc").exists() and not shutil.which("hipcc"):
        pytest.skip("no hipcc")
    root = Path(__file__).resolve().parent.parent
    pr = subprocess.run([sys.executable, str(root / "tools" / "isa_inflate_check.py")], capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0 and pr.stdout.startswith("ok:"), pr.stdout + pr.stderr
