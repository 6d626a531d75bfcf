# tools/e2e_prof.sh -- GPU-box helper: rocprofv3 kernel stats of the whole bin/pss-bam command on a generated BAM + FASTA
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_e2e
rm -rf $OUT/stats; mkdir -p $OUT /tmp/e2e_in
python3 - <<'PY'
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as ge, bench
pkg = ge.load_pkg()
from pss_bam_amd import synth
d = synth.config("C3"); d.pop("region_len")
cfg = synth.make_cfg(**d)
n = int(os.environ.get("E2E_READS", "200000000"))
synth.fasta_host(cfg, "/tmp/e2e_in/ref.fa", threads=bench.worker_threads())
synth.bam_file_host(cfg, 0, n, "/tmp/e2e_in/reads.bam", level=int(os.environ.get("E2E_LEVEL", "1")), threads=bench.worker_threads(),
                    ragged=bool(int(os.environ.get("E2E_RAGGED", "0"))), quals=os.environ.get("E2E_QUALS", "const"))
PY
PSSBAM_CLEAN_EXIT=1 PSSBAM_STATS=1 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/stats -- pss-bam_amd/bin/pss-bam -F /tmp/e2e_in/ref.fa -B /tmp/e2e_in/reads.bam -o /tmp/e2e_in/out -r 25 > $OUT/run.log 2>&1
grep -E "device feed|phases|gpus=|early feed|engine feed|process creation" $OUT/run.log
find $OUT/stats -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-200
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; cp $(find $OUT/stats -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv
python3 tools/timeline.py $OUT/stats --bin ${E2E_BIN:-20} > $OUT/timeline.txt; cat $OUT/timeline.txt
rm -rf /tmp/e2e_in
