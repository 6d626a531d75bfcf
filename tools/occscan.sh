# tools/occscan.sh -- GPU-box helper: tile size x workgroups-per-CU target
for cfg in "128 3" "96 3" "80 4" "64 4" "64 5"; do set -- $cfg
  echo "== T=$1 want_wgs=$2"
  PSSBAM_TILE_READS=$1 PSSBAM_WANT_WGS=$2 python bench.py --reads 50000000 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('reads/s %.3e  GB/s %.0f  ms/launch %.3f' % (d['value'], d['roofline']['achieved'], d['roofline']['kernel_ms_per_launch']))"
done
