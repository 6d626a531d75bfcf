# tools/touch_ab.sh -- GPU-box helper: bgzf_inflate_kernel<true> with and without the stream's line touch (PSSBAM_INFLATE_TOUCH)
# on the same generated BAMs, same box.  AB_CASES: "quals:level" pairs; AB_VAR / AB_VALUES: the variable scanned
OUT=gpurun_out/${AB_OUT:-touch_ab.txt}
mkdir -p gpurun_out; : > $OUT
VAR=${AB_VAR:-PSSBAM_INFLATE_TOUCH}
for cs in ${AB_CASES:-const:1 binned:6 full:6}; do
  q=${cs%%:*}; lvl=${cs##*:}
  for v in ${AB_VALUES:-0 1}; do
    env $VAR=$v python3 tools/inflate_bench.py --reads ${AB_READS:-40000000} --level $lvl --quals $q --no-output ${AB_CRC:---no-crc} --repeats 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('quals $q level $lvl $VAR=$v: %.2f GB -> %.2f GB (%.1fx), %.1f GB/s inflated, %.2f ms' % (d['bam_bytes']/1e9, d['inflated_bytes']/1e9, d['inflated_bytes']/d['bam_bytes'], d['device_GBps_inflated'], d['device_kernel_ms']))" >> $OUT
    tail -1 $OUT
  done
done
