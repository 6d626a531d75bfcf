# tools/inflate_prof.sh -- GPU-box helper: rocprofv3 stats + SQ counters of the device inflate on a 100 M-read BAM
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_inflate
mkdir -p $OUT
B="python3 tools/inflate_bench.py --reads ${INF_READS:-100000000} --no-output ${INF_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1 || tail -5 $OUT/stats.log
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc1 -- $B > $OUT/pmc1.log 2>&1 || tail -5 $OUT/pmc1.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc2 -- $B > $OUT/pmc2.log 2>&1 || tail -5 $OUT/pmc2.log
tail -1 $OUT/stats.log
find $OUT/stats -name "*kernel_stats.csv" -exec head -6 {} \;
python3 - <<'PY'
import csv,glob
from collections import defaultdict
for p in ("pmc1","pmc2"):
    for f in glob.glob(f"gpurun_out/prof_inflate/{p}/**/*counter_collection.csv", recursive=True):
        acc=defaultdict(lambda: defaultdict(float)); cnt=defaultdict(int)
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"][:40]
            if "bgzf" in k:
                acc[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
        for k,v in acc.items():
            print(p,k,{c:round(x/cnt[(k,c)]/1e6,2) for c,x in v.items()},"(millions per dispatch)")
PY
