# tools/inflate_scan.sh -- GPU-box helper: device inflate vs waves per CU and file size
for w in 1 2 3; do
  echo "== 20M reads, $w waves/CU"
  PSSBAM_INFLATE_WAVES_PER_CU=$w python3 tools/inflate_bench.py --reads 20000000 --no-output 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f GB/s inflated, %.1f ms; host inflate stage %.1f GB/s' % (d['device_GBps_inflated'], d['device_kernel_ms'], d['host_inflate_stage_GBps'] or 0))"
done
echo "== 100M reads, 3 waves/CU"
python3 tools/inflate_bench.py --reads 100000000 --no-output 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f GB/s inflated, %.1f ms; host inflate stage %.1f GB/s; blocks %d' % (d['device_GBps_inflated'], d['device_kernel_ms'], d['host_inflate_stage_GBps'] or 0, d['n_blocks']))"
