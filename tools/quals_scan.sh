# tools/quals_scan.sh -- GPU-box helper: device inflate and the whole command on BAMs with sequencer-like QUAL strings
# (synth.bam_file_host quals="binned" / "full"; the named configurations have constant QUAL)
OUT=gpurun_out/quals_scan.txt
mkdir -p gpurun_out; : > $OUT
for cs in ${Q_CASES:-const:1 const:6 binned:1 binned:6 full:1 full:6}; do
  q=${cs%%:*}; lvl=${cs##*:}
  python3 tools/inflate_bench.py --reads ${Q_READS:-60000000} --level $lvl --quals $q --no-output --repeats 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('inflate+crc, quals $q level $lvl: %.2f GB bam -> %.2f GB (%.1fx), device %.1f GB/s inflated (%.1f GB/s compressed), %.2f ms; host inflate stage %.1f GB/s on %s cpus' % (d['bam_bytes']/1e9, d['inflated_bytes']/1e9, d['inflated_bytes']/d['bam_bytes'], d['device_GBps_inflated'], d['device_GBps_compressed'], d['device_kernel_ms'], d['host_inflate_stage_GBps'] or 0, d['host_cpus_effective']))" >> $OUT
  tail -1 $OUT
done
python3 tools/feed_scan.py --reads ${Q_E2E_READS:-100000000} --level 6 --quals full -- "" "PSSBAM_DEVICE_INFLATE=0" >> $OUT 2>&1
tail -3 $OUT | cut -c1-700
