# tools/bench_full.sh -- GPU-box helper: the default bench (BASELINE metric config) + rocprofv3 evidence
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
python3 bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err || { tail -20 gpurun_out/full/bench.err; exit 1; }
cat gpurun_out/full/bench.json
# kernel-trace summary of the same workload (the CPU-baseline leg is a child process without kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/full/stats -- python3 bench.py --no-cpu-baseline --no-e2e > gpurun_out/full/stats.log 2>&1 || tail -5 gpurun_out/full/stats.log
tail -2 gpurun_out/full/stats.log
find gpurun_out/full/stats -name "*kernel_stats.csv" -exec cat {} \;
PROF_READS=${PROF_READS:-50000000} bash tools/prof1.sh > gpurun_out/full/prof.txt 2>&1 || tail -20 gpurun_out/full/prof.txt
tail -60 gpurun_out/full/prof.txt
