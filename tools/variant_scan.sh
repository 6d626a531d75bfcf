# tools/variant_scan.sh -- GPU-box helper: inflate throughput of pre-built variants of libpssbam_hip.so
# (pss-bam_amd/variants/lib_*.so, built with -DINF_...=..., see csrc/inflate_kernels.h); results -> gpurun_out/variant_scan.txt
# VAR_CASES: "quals:level" pairs (synth.bam_file_host quals model : deflate level)
OUT=gpurun_out/variant_scan.txt
mkdir -p gpurun_out; : > $OUT
cp pss-bam_amd/libpssbam_hip.so /tmp/lib_orig.so
for v in pss-bam_amd/variants/lib_*.so; do
  cp $v pss-bam_amd/libpssbam_hip.so
  for cs in ${VAR_CASES:-const:1 const:6}; do
    q=${cs%%:*}; lvl=${cs##*:}
    python3 tools/inflate_bench.py --reads ${VAR_READS:-60000000} --level $lvl --quals $q --no-output --no-crc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v quals $q level $lvl: %.1f GB/s, %.2f ms' % (d['device_GBps_inflated'], d['device_kernel_ms']))" >> $OUT
    tail -1 $OUT
  done
done
cp /tmp/lib_orig.so pss-bam_amd/libpssbam_hip.so
