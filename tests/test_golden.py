"""Oracle (CPU restatement) against the committed golden vectors in tests/golden/, which
hold the UNMODIFIED reference's outputs (see tests/golden/make_golden.py).  Runs without
/root/reference and without oracle/_ref."""
import json
import os
from pathlib import Path

import numpy as np
import pytest

from pssbam_testlib import FkOpts, PssOpts, parse_counts_text, parse_fragkon_text

GOLD = Path(__file__).resolve().parent / "golden"
MANIFEST = json.loads((GOLD / "manifest.json").read_text())
PSS_CASES = [c for c in MANIFEST["cases"] if c["tool"] == "pss-bam"]
FK_CASES = [c for c in MANIFEST["cases"] if c["tool"] == "fragkon"]


def _sam_for(case, tmp_path):
    """-R is applied by `samtools view -r` in the reference pipeline, i.e. before the
    text reaches line2saml: emulate that on the SAM fixture."""
    ds = MANIFEST["datasets"][case["dataset"]]
    sam = GOLD / ds["sam"]
    rg = case["opts"].get("read_group")
    if rg is None:
        return sam
    out = tmp_path / "rg.sam"
    want = f"RG:Z:{rg}"
    with open(sam) as fi, open(out, "w") as fo:
        for ln in fi:
            if ln.startswith("@") or want in ln.rstrip("\n").split("\t")[11:]:
                fo.write(ln)
    return out


@pytest.mark.parametrize("case", PSS_CASES, ids=[c["prefix"] for c in PSS_CASES])
def test_oracle_pss_golden(case, oracle, tmp_path):
    ds = MANIFEST["datasets"][case["dataset"]]
    o = PssOpts(**case["opts"])
    g = oracle.load_genome(GOLD / ds["fasta"])
    fwd, rev, _ = oracle.pss(g, _sam_for(case, tmp_path), o)
    oracle.free_genome(g)
    want_counts = (GOLD / case["counts"]).read_text()
    wf, wr = parse_counts_text(want_counts)
    assert np.array_equal(fwd, wf) and np.array_equal(rev, wr)
    # report text: byte-exact, including the echoed -F/-B/-o strings (relative names)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        oracle.write_reports(ds["fasta"], ds["sam"], case["prefix"], fwd, rev)
        assert Path(case["counts"]).read_text() == want_counts
        assert Path(case["rates"]).read_text() == (GOLD / case["rates"]).read_text()
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize("case", FK_CASES, ids=[c["stdout"] for c in FK_CASES])
def test_oracle_fragkon_golden(case, oracle):
    ds = MANIFEST["datasets"][case["dataset"]]
    o = FkOpts(**case["opts"])
    g = oracle.load_genome(GOLD / ds["fasta"])
    k5, k3, _ = oracle.fragkon(g, GOLD / ds["sam"], o)
    oracle.free_genome(g)
    w5, w3 = parse_fragkon_text((GOLD / case["stdout"]).read_text())
    assert np.array_equal(k5, w5) and np.array_equal(k3, w3)


@pytest.mark.parametrize("case", [c for c in MANIFEST["cases"] if c["tool"] == "genome-kmer-count"], ids=lambda c: c["stdout"])
def test_oracle_genome_kmer_count_golden(case, oracle):
    ds = MANIFEST["datasets"][case["dataset"]]
    g = oracle.load_genome(GOLD / ds["fasta"])
    got = oracle.genome_kmer_count(g, case["klen"])
    oracle.free_genome(g)
    want = np.array([int(ln.split("\t")[1]) for ln in (GOLD / case["stdout"]).read_text().splitlines()[1:]], dtype=np.uint32)
    assert np.array_equal(got, want)
