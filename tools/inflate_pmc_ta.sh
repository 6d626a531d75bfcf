# tools/inflate_pmc_ta.sh -- GPU-box helper: is the per-CU address path (TA) what the inflate kernel saturates?
# Every pass sits under its own `timeout`: a counter set the hardware cannot schedule aborts inside
# rocprofv3 and then hangs.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_inflate_ta
mkdir -p $OUT
B="python3 tools/inflate_bench.py --reads ${INF_READS:-20000000} --no-output --no-crc --repeats 1"
pass() { n=$1; shift; timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- $B > $OUT/$n.log 2>&1; echo "pass $n ($*): rc=$?" >> $OUT/progress.txt; tail -1 $OUT/progress.txt; }
: > $OUT/progress.txt
pass a TA_TA_BUSY_sum GRBM_GUI_ACTIVE
pass b TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
pass c TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum
pass d TA_BUSY_avr TA_BUSY_max
python3 - <<'PY'
import csv,glob
from collections import defaultdict
for p in "abcd":
    for f in glob.glob(f"gpurun_out/prof_inflate_ta/{p}/**/*counter_collection.csv", recursive=True):
        acc=defaultdict(float); cnt=defaultdict(int)
        for row in csv.DictReader(open(f)):
            if "bgzf_inflate" in row["Kernel_Name"]:
                acc[row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[row["Counter_Name"]]+=1
        print(p,{c:round(x/cnt[c]/1e6,3) for c,x in acc.items()},"(millions per dispatch)")
PY
