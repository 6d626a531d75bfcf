# tools/crc_pmc.sh -- GPU-box helper: L2 / fabric counters of bgzf_crc_kernel (60 M-read level-1 BAM): does every 128-byte line
# come from L2 once, or once per 16-byte load of the lane that owns it?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_crc
mkdir -p $OUT
B="python3 tools/inflate_bench.py --reads ${INF_READS:-60000000} --no-output --repeats 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1 || tail -3 $OUT/p1.log
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/p2 -- $B > $OUT/p2.log 2>&1 || tail -3 $OUT/p2.log
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/p3 -- $B > $OUT/p3.log 2>&1 || tail -3 $OUT/p3.log
python3 - <<'PY'
import csv,glob
from collections import defaultdict
for p in ("p1","p2","p3"):
    for f in glob.glob(f"gpurun_out/prof_crc/{p}/**/*counter_collection.csv", recursive=True):
        acc=defaultdict(float); cnt=defaultdict(int)
        for row in csv.DictReader(open(f)):
            if "bgzf_crc" in row["Kernel_Name"]:
                acc[row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[row["Counter_Name"]]+=1
        print(p,{c:round(x/cnt[c]/1e6,3) for c,x in acc.items()},"(millions per dispatch)")
    for f in glob.glob(f"gpurun_out/prof_crc/{p}/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "bgzf_crc" in row["Kernel_Name"]:
                print(p,"crc kernel ms", (int(row["End_Timestamp"])-int(row["Start_Timestamp"]))/1e6)
PY
