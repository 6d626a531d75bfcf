# tools/wave_prof.sh -- GPU-box helper: per-kernel time of the wave-per-block inflate path on generated BAMs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_wave
mkdir -p $OUT
for cs in ${AB_CASES:-const:1 full:6}; do
  q=${cs%%:*}; lvl=${cs##*:}
  rm -rf $OUT/s_$q
  PSSBAM_INFLATE_WAVE=${WAVE:-2} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s_$q -- python3 tools/inflate_bench.py --reads ${AB_READS:-20000000} --level $lvl --quals $q --no-output --no-crc --repeats 1 > $OUT/$q.log 2>&1 || tail -3 $OUT/$q.log
  echo "== quals $q level $lvl"; tail -1 $OUT/$q.log | cut -c1-300
  find $OUT/s_$q -name "*kernel_stats.csv" -exec head -5 {} \; | cut -c1-160
done
