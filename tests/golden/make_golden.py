#!/usr/bin/env python3
"""Regenerates tests/golden/* by running the UNMODIFIED reference (oracle/_ref, built by
oracle/Makefile from /root/reference) on small synthetic inputs made by
tests/pssbam_testlib.py.  Only data is written here: inputs (FASTA, SAM, and the BAM
encoding of the same records) and the reference's outputs for them.

    python tests/golden/make_golden.py        # needs oracle/_ref (i.e. /root/reference)

The committed fixtures were produced by exactly this script.
"""
import json
import sys
from dataclasses import asdict
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))

from pssbam_testlib import (FkOpts, PssOpts, build_oracle, fuzz_dataset, have_ref, ref_safe,  # noqa: E402
                            run_ref_fragkon, run_ref_gkc, run_ref_pss, write_bam, write_fasta, write_sam)
import numpy as np  # noqa: E402


def main():
    build_oracle()
    assert have_ref(), "oracle/_ref missing: this script needs the reference sources"
    manifest = {"note": "expected outputs produced by oracle/_ref (unmodified reference); see make_golden.py",
                "datasets": {}, "cases": []}

    # dataset A: branch-coverage fuzz set, no aux tags; dataset B: with RG / NM / XA tags
    for name, seed, n, with_rg in (("A", 424242, 700, False), ("B", 515151, 500, True)):
        contigs, refs, recs = fuzz_dataset(seed, n, with_rg=with_rg)
        recs = ref_safe(recs, klen=5)   # keeps the set valid for every k <= 5 case below
        fa = HERE / f"set{name}.fa"
        sam = HERE / f"set{name}.sam"
        bam = HERE / f"set{name}.bam"
        write_fasta(fa, contigs, width=60)
        write_sam(sam, refs, recs)
        write_bam(bam, refs, recs, level=6, rng=np.random.default_rng(seed))
        manifest["datasets"][name] = {"fasta": fa.name, "sam": sam.name, "bam": bam.name, "n_records": len(recs)}

    pss_cases = [
        ("A", PssOpts()),
        ("A", PssOpts(region_len=25, min_mq=20)),
        ("A", PssOpts(region_len=8, min_read_len=20, max_read_len=70, up_ctx="CT", down_ctx="AG", merged_only=True)),
        ("A", PssOpts(region_len=40, up_ctx="ACGTN")),
        ("B", PssOpts(region_len=10)),
        ("B", PssOpts(region_len=10, read_group="grpA")),
        ("B", PssOpts(region_len=15, read_group="grpB", min_mq=10)),
    ]
    for i, (ds, o) in enumerate(pss_cases):
        d = manifest["datasets"][ds]
        tag = f"pss_{i}"
        # -F/-B/-o strings are echoed into the report headers: run with cwd-relative names
        import os
        cwd = os.getcwd()
        os.chdir(HERE)
        try:
            run_ref_pss(Path(d["fasta"]), Path(d["sam"]), Path(tag), o)
        finally:
            os.chdir(cwd)
        manifest["cases"].append({"tool": "pss-bam", "dataset": ds, "opts": asdict(o), "prefix": tag,
                                  "counts": f"{tag}.pss.counts.txt", "rates": f"{tag}.pss.rates.txt"})

    fk_cases = [("A", FkOpts(klen=4)), ("A", FkOpts(klen=3, min_mq=20)), ("A", FkOpts(klen=5, merged_only=True)),
                ("A", FkOpts(klen=2, min_read_len=30, max_read_len=80)), ("B", FkOpts(klen=1))]
    for i, (ds, o) in enumerate(fk_cases):
        d = manifest["datasets"][ds]
        tag = f"fragkon_{i}.txt"
        import os
        cwd = os.getcwd()
        os.chdir(HERE)
        try:
            _, _, out, _ = run_ref_fragkon(Path(d["fasta"]), Path(d["sam"]), o)
        finally:
            os.chdir(cwd)
        (HERE / tag).write_text(out)
        manifest["cases"].append({"tool": "fragkon", "dataset": ds, "opts": asdict(o), "stdout": tag})

    for ds, k in (("A", 4), ("A", 3), ("B", 6)):
        d = manifest["datasets"][ds]
        tag = f"gkc_{ds}_{k}.txt"
        import os
        cwd = os.getcwd()
        os.chdir(HERE)
        try:
            _, out = run_ref_gkc(Path(d["fasta"]), k)
        finally:
            os.chdir(cwd)
        (HERE / tag).write_text(out)
        manifest["cases"].append({"tool": "genome-kmer-count", "dataset": ds, "klen": k, "stdout": tag})

    (HERE / "manifest.json").write_text(json.dumps(manifest, indent=1) + "\n")
    print(f"wrote {len(manifest['cases'])} golden cases to {HERE}")


if __name__ == "__main__":
    main()
