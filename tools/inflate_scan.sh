# tools/inflate_scan.sh -- GPU-box helper: device inflate vs streams in flight (waves per CU; 5 is what LDS allows)
OUT=gpurun_out/inflate_waves_scan.txt
mkdir -p gpurun_out; : > $OUT
for w in ${SCAN_WAVES:-2 3 4 5}; do
  echo "== ${INF_READS:-60000000} reads, level 1, $w waves/CU, loop ${PSSBAM_INFLATE_LOOP:-default}" >> $OUT
  PSSBAM_INFLATE_WAVES_PER_CU=$w python3 tools/inflate_bench.py --reads ${INF_READS:-60000000} --no-output --no-crc --repeats 3 ${INF_ARGS:-} 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f GB/s inflated, %.2f ms, %d blocks' % (d['device_GBps_inflated'], d['device_kernel_ms'], d['n_blocks']))" >> $OUT
done
cat $OUT
