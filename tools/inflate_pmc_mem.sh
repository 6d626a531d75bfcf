# tools/inflate_pmc_mem.sh -- GPU-box helper: L2/fabric traffic of the device inflate (60 M-read level-1 BAM).
# Two PMC passes only (FETCH_SIZE needs a pass of its own).  A pass with a counter set the hardware cannot
# schedule aborts inside rocprofv3 and then sits there until the silence limit kills the call -- do not add
# TA_*/TCP_* sets here without trying them on a tiny input first.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_inflate_mem
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA_[A-Z_0-9a-z]+|TCP_[A-Z_0-9a-z]+|TCC_[A-Z_0-9a-z]+)\b" | sort -u | tr '\n' ' ' | cut -c1-3000 > $OUT/counters.txt
B="python3 tools/inflate_bench.py --reads ${INF_READS:-60000000} --no-output --no-crc --repeats 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1 || tail -3 $OUT/p1.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p2 -- $B > $OUT/p2.log 2>&1 || tail -3 $OUT/p2.log
python3 - <<'PY'
import csv,glob
from collections import defaultdict
for p in ("p1","p2"):
    for f in glob.glob(f"gpurun_out/prof_inflate_mem/{p}/**/*counter_collection.csv", recursive=True):
        acc=defaultdict(float); cnt=defaultdict(int)
        for row in csv.DictReader(open(f)):
            if "bgzf_inflate" in row["Kernel_Name"]:
                acc[row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[row["Counter_Name"]]+=1
        print(p,{c:round(x/cnt[c]/1e6,3) for c,x in acc.items()},"(millions per dispatch)")
PY
