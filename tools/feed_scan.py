#!/usr/bin/env python3
"""tools/feed_scan.py -- GPU-box helper: one generated BAM + FASTA, bin/pss-bam run under several
environment settings (inflate thread counts, copy streams); prints the front end's phase lines."""
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_pkg()
from pss_bam_amd import synth  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000_000
d = synth.config("C2", n_reads=reads, scale_genome=0.1)
region_len = d.pop("region_len")
d.pop("klen", None)
cfg = synth.make_cfg(**d)
tmp = Path(tempfile.mkdtemp(prefix="pssbam_scan_", dir=os.environ.get("TMPDIR", "/tmp")))
fa, bam = tmp / "ref.fa", tmp / "reads.bam"
threads = os.cpu_count() or 8
synth.fasta_host(cfg, fa, threads=threads)
synth.bam_file_host(cfg, 0, reads, bam, level=1, threads=threads)
settings = [{"PSSBAM_INFLATE_THREADS": t} for t in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["32", "48", "64"])]
for env in settings:
    for rep in range(2):
        t = time.time()
        pr = subprocess.run([str(pkg.PKG_DIR / "bin" / "pss-bam"), "-F", str(fa), "-B", str(bam), "-o", str(tmp / "out"), "-r", str(region_len)],
                            capture_output=True, text=True, env={**os.environ, "PSSBAM_STATS": "1", **env})
        wall = time.time() - t
        lines = [ln for ln in pr.stderr.splitlines() if "phases" in ln or "reader thread" in ln]
        print(env, f"wall {wall:.3f}", " | ".join(ln.replace("[pssbam] ", "") for ln in lines), flush=True)
for p in tmp.iterdir():
    p.unlink()
tmp.rmdir()
