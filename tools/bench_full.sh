# tools/bench_full.sh -- GPU-box helper: the default bench (BASELINE metric config) + rocprofv3 evidence
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
python3 bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err || { tail -20 gpurun_out/full/bench.err; exit 1; }
cat gpurun_out/full/bench.json
PROF_READS=${PROF_READS:-50000000} bash tools/prof1.sh > gpurun_out/full/prof.txt 2>&1 || tail -20 gpurun_out/full/prof.txt
tail -60 gpurun_out/full/prof.txt
