# tools/prof1.sh -- GPU-box helper: rocprofv3 kernel stats + PMC passes on a short bench run
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof
mkdir -p $OUT
B="python3 bench.py --reads ${PROF_READS:-30000000} --steps 3 --warmup 1 --no-cpu-baseline --no-e2e"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1 || tail -5 $OUT/stats.log
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc1 -- $B > $OUT/pmc1.log 2>&1 || tail -5 $OUT/pmc1.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc2 -- $B > $OUT/pmc2.log 2>&1 || tail -5 $OUT/pmc2.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- $B > $OUT/pmc3.log 2>&1 || tail -5 $OUT/pmc3.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -- $B > $OUT/pmc4.log 2>&1 || tail -5 $OUT/pmc4.log
find $OUT -name "*.csv" | head -30
python3 tools/prof_summary.py $OUT
