#!/usr/bin/env python3
"""tools/isa_inflate_check.py -- the schedule the one-wait-per-step inflate loop depends on, asserted on the compiled kernel.

The loop (pss-bam_amd/csrc/inflate_kernels.h, inflate_block<true, true>) issues five 16-byte LDS-DMA requests (four for a copy
piece's source, one for the stream) at the top of a step and waits for them with ONE hand-written `s_waitcnt vmcnt(0)` behind
the decode phase.  That only pays while
the compiler neither waits nor touches global memory in between.  Checked on the hipcc -S listing of engine.hip:

  * bgzf_inflate_kernel<true, true> contains exactly five global_load_lds_dwordx4 (one group, the top of the step);
  * every one of them sits between a save and a restore of m0 (the compiler does not model the write);
  * from the last request to the first s_waitcnt that names vmcnt there is no vector-memory instruction, there are
    LDS reads (the decode phase) and at least MIN_GAP instructions;
  * (scratch traffic -- the block-header code keeps a saved BitReader there -- counts as vector memory: none in that
    stretch.)

usage: isa_inflate_check.py [engine.s]      (no argument: compiles csrc/engine.hip with --save-temps into a temp dir)"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
MIN_GAP = 150


def listing() -> str:
    if len(sys.argv) > 1:
        return Path(sys.argv[1]).read_text()
    with tempfile.TemporaryDirectory(prefix="pssbam_isa_") as d:
        src = ROOT / "pss-bam_amd" / "csrc" / "engine.hip"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
                        f"-I{ROOT / 'include'}", "-c", str(src), "--save-temps", "-o", "engine.o"], cwd=d, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return next(Path(d).glob("*gfx950.s")).read_text()


def main() -> int:
    s = listing()
    names = re.findall(r"^(_ZN6pssbam19bgzf_inflate_kernelILb1ELb1E\w*):", s, re.M)
    assert names, "bgzf_inflate_kernel<true, true> not found in the listing"
    name = names[0]
    i = s.index("\n" + name + ":")
    lines = s[i:s.index("s_endpgm", i)].split("\n")
    ins = [(k, l.strip()) for k, l in enumerate(lines) if l.strip() and l.strip()[0] not in ";." and not l.strip().endswith(":")]
    dma = [n for n, (k, l) in enumerate(ins) if l.startswith("global_load_lds_dwordx4")]
    assert len(dma) == 5, f"expected 5 LDS-DMA requests, found {len(dma)}"
    assert dma[-1] - dma[0] < 70, "the requests are not one group"
    for n in dma:
        before = [l for _, l in ins[max(0, n - 4):n]]
        after = [l for _, l in ins[n + 1:n + 3]]
        assert any(re.match(r"s_mov_b32 s\d+, m0", l) for l in before) and any(re.match(r"s_mov_b32 m0, s\d+", l) for l in before), \
            f"m0 not saved/set in front of request {n}: {before}"
        assert any(re.match(r"s_mov_b32 m0, s\d+", l) for l in after), f"m0 not restored behind request {n}: {after}"
    w = next(n for n in range(dma[-1] + 1, len(ins)) if ins[n][1].startswith("s_waitcnt") and "vmcnt" in ins[n][1])
    between = [l for _, l in ins[dma[-1] + 1:w]]
    vmem = [l for l in between if re.match(r"(global|flat|scratch|buffer)_", l)]
    assert not vmem, f"vector-memory instructions between the requests and the wait: {vmem[:3]}"
    assert sum(l.startswith("ds_read") for l in between) >= 3, "no decode phase (LDS reads) between the requests and the wait"
    assert len(between) >= MIN_GAP, f"only {len(between)} instructions between the requests and the wait"
    assert ins[w][1].split()[1].startswith("vmcnt(0)"), ins[w][1]
    k = s.index(".amdhsa_kernel " + name)
    vgpr = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", s[k:k + 4000]).group(1))
    scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", s[k:k + 4000]).group(1))
    print(f"ok: 5 requests, {len(between)} instructions (no VMEM, {sum(l.startswith('ds_read') for l in between)} LDS reads) to the wait, {vgpr} VGPRs, {scratch} B scratch")
    return 0


if __name__ == "__main__":
    sys.exit(main())
