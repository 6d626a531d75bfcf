# tools/loop_ab.sh -- GPU-box helper: the two data loops of bgzf_inflate_kernel (PSSBAM_INFLATE_LOOP=0 in place, 1 one wait per
# step) on the same generated BAMs, same box.  LOOP_CASES: "quals:level" pairs
OUT=gpurun_out/loop_ab.txt
mkdir -p gpurun_out; : > $OUT
for cs in ${LOOP_CASES:-const:1 const:6 binned:6 full:6}; do
  q=${cs%%:*}; lvl=${cs##*:}
  for loop in 0 1; do
    PSSBAM_INFLATE_LOOP=$loop python3 tools/inflate_bench.py --reads ${LOOP_READS:-40000000} --level $lvl --quals $q --no-output --no-crc --repeats 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('quals $q level $lvl loop $loop: %.2f GB -> %.2f GB (%.1fx), %.1f GB/s inflated, %.2f ms' % (d['bam_bytes']/1e9, d['inflated_bytes']/1e9, d['inflated_bytes']/d['bam_bytes'], d['device_GBps_inflated'], d['device_kernel_ms']))" >> $OUT
    tail -1 $OUT
  done
done
