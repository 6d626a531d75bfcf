# tools/pmc_all.sh -- GPU-box helper: rocprofv3 kernel stats + PMC passes for every named configuration
# (tools/pmc_profile.py each); summaries land in gpurun_out/prof_<tag>/summary.json
set -x
python3 tools/pmc_profile.py C3_200M -- --config C3 --steps 3 --warmup 1 > gpurun_out/pmc_C3_200M.txt 2>&1
python3 tools/pmc_profile.py C2 -- --config C2 --steps 3 --warmup 1 > gpurun_out/pmc_C2.txt 2>&1
python3 tools/pmc_profile.py C3_shuffled -- --config C3 --unsorted --reads 100000000 --steps 3 --warmup 1 > gpurun_out/pmc_C3_shuffled.txt 2>&1
python3 tools/pmc_profile.py C4 -- --config C4 --steps 3 --warmup 1 > gpurun_out/pmc_C4.txt 2>&1
python3 tools/pmc_profile.py C5 -- --config C5 --reads 100000000 --steps 3 --warmup 1 > gpurun_out/pmc_C5.txt 2>&1
python3 tools/pmc_profile.py C1 -- --config C1 --reads 20000000 --steps 3 --warmup 1 > gpurun_out/pmc_C1.txt 2>&1
for f in gpurun_out/pmc_*.txt; do tail -n 4 "$f"; done
