#!/usr/bin/env python3
"""tools/next_rows.py -- GPU-box measurement of the SURVEY 8f rows that bench.py does not time:
  f4  genome-kmer-count: the kernel over the full 3.1 Gb device genome (k = 4 and 8), and the
      command on a FASTA file next to the unmodified reference's command on the same file;
  f3  SAM-text input: bin/pss-bam fed SAM text instead of BAM (same tables as from the BAM).
Prints one JSON object (kept as profiles/r01_next_rows.json)."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_pkg()
from pss_bam_amd import synth  # noqa: E402
import pssbam_testlib as tl  # noqa: E402

out = {}
dev = torch.device("cuda", 0)
threads = os.cpu_count() or 8

# ---- f4: kernel over the full-size genome ------------------------------------------------------
d = synth.config("C3")
d.pop("region_len")
d.pop("klen", None)
cfg = synth.make_cfg(**d)
S = synth.lib()
stream = torch.cuda.current_stream().cuda_stream
names = [synth.contig_name(cfg, k) for k in range(int(cfg.n_contigs))]
contigs = []
for k in range(int(cfg.n_contigs)):
    ln = int(cfg.contig_len[k])
    t = torch.empty(ln + 64, dtype=torch.uint8, device=dev)
    assert S.synth_genome_device(C.byref(cfg), k, t.data_ptr(), ln, stream) == 0
    contigs.append(t)
eng = pkg.Engine(kmer=dict(klen=4))
eng.set_stream(stream)
eng.set_genome_device([(names[k], contigs[k].data_ptr(), int(cfg.contig_len[k])) for k in range(len(names))])
bases = sum(int(cfg.contig_len[k]) for k in range(len(names)))
del contigs
for k in (4, 8):
    eng.genome_kmer_count(k)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    counts = eng.genome_kmer_count(k)
    dt = time.perf_counter() - t0   # includes the 4^k-bin read-back
    out[f"gkc_kernel_k{k}"] = {"bases": bases, "seconds": dt, "GB_per_s": bases / dt / 1e9, "windows_counted": int(counts.sum())}
eng.close()

# ---- f4: the command on a file, next to the reference's ------------------------------------------
tmp = Path(tempfile.mkdtemp(prefix="pssbam_next_", dir=os.environ.get("TMPDIR", "/tmp")))
d = synth.config("C2", n_reads=2_000_000, scale_genome=0.1)
region_len = d.pop("region_len")
d.pop("klen", None)
cfg = synth.make_cfg(**d)
fa = tmp / "ref.fa"
synth.fasta_host(cfg, fa, threads=threads)
t0 = time.perf_counter()
mine = subprocess.run([str(pkg.PKG_DIR / "bin" / "genome-kmer-count"), "-f", str(fa), "-k", "4"], capture_output=True, text=True, check=True).stdout
t_mine = time.perf_counter() - t0
row = {"fasta_bytes": fa.stat().st_size, "k": 4, "seconds_whole_command": t_mine}
if tl.have_ref():
    t0 = time.perf_counter()
    _, ref_out = tl.run_ref_gkc(fa, 4, timeout=1800)
    row["reference_seconds_whole_command"] = time.perf_counter() - t0
    row["stdout_identical"] = ref_out == mine
out["gkc_command"] = row

# ---- f3: SAM text in ----------------------------------------------------------------------------
n = int(cfg.n_reads)
sam, bam = tmp / "reads.sam", tmp / "reads.bam"
synth.sam_host(cfg, 0, n, sam)
synth.bam_file_host(cfg, 0, n, bam, level=1, threads=threads)
res = {}
for tag, path in (("sam", sam), ("bam", bam)):
    t0 = time.perf_counter()
    pr = subprocess.run([str(pkg.PKG_DIR / "bin" / "pss-bam"), "-F", str(fa), "-B", str(path), "-o", str(tmp / tag), "-r", str(region_len)],
                        capture_output=True, text=True, env={**os.environ, "PSSBAM_STATS": "1"})
    res[tag] = time.perf_counter() - t0
    assert pr.returncode == 0, pr.stderr[-1000:]
same = (tmp / "sam.pss.counts.txt").read_text().split("\n", 4)[4] == (tmp / "bam.pss.counts.txt").read_text().split("\n", 4)[4]
out["sam_text_input"] = {"reads": n, "sam_bytes": sam.stat().st_size, "seconds_whole_command": res["sam"],
                         "reads_per_s": n / res["sam"], "same_reads_from_bam_seconds": res["bam"], "tables_identical": bool(same)}
print(json.dumps(out))
for p in tmp.iterdir():
    p.unlink()
tmp.rmdir()
