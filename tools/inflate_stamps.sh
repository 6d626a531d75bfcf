# tools/inflate_stamps.sh -- GPU-box helper: the diagnostic (stamped) build of the inflate loop on generated BAMs:
# where the cycles of a step go (PSSBAM_INFLATE_STAMPS, csrc/inflate_kernels.h StepStamps)
OUT=gpurun_out/inflate_stamps.txt
mkdir -p gpurun_out; : > $OUT
for cs in ${AB_CASES:-const:1 binned:6 full:6}; do
  q=${cs%%:*}; lvl=${cs##*:}
  for pc in ${AB_PIECES:-0 1}; do
  echo "== quals $q level $lvl pieces $pc" >> $OUT
  PSSBAM_INFLATE_PIECES=$pc PSSBAM_INFLATE_STAMPS=1 python3 tools/inflate_bench.py --reads ${AB_READS:-40000000} --level $lvl --quals $q --no-output --no-crc --repeats 1 2> /tmp/stamps.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  %.2f GB -> %.2f GB (%.1fx), stamped build %.1f GB/s, %.2f ms' % (d['bam_bytes']/1e9, d['inflated_bytes']/1e9, d['inflated_bytes']/d['bam_bytes'], d['device_GBps_inflated'], d['device_kernel_ms']))" >> $OUT
  grep "inflate stamps" /tmp/stamps.err >> $OUT
  done
done
cat $OUT
