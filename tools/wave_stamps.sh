# tools/wave_stamps.sh -- GPU-box helper: phase stamps of the token kernel (PSSBAM_WAVE_STAMPS) on generated BAMs
for cs in ${AB_CASES:-const:1 binned:6 full:6}; do
  q=${cs%%:*}; lvl=${cs##*:}
  echo "== quals $q level $lvl"
  PSSBAM_INFLATE_WAVE=2 PSSBAM_WAVE_STAMPS=1 python3 tools/inflate_bench.py --reads ${AB_READS:-20000000} --level $lvl --quals $q --no-output --no-crc --repeats 1 2>&1 | grep -o "token kernel stamps.*\|resolve kernel stamps.*\|device_GBps_inflated[^,]*"
done
