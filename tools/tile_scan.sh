# tools/tile_scan.sh -- GPU-box helper: reads per tile vs occupancy (PSSBAM_TILE_READS) for the short-window configs
for cfg in "--config C4" "--config C1 --reads 20000000"; do
  for t in 128 112 96 80; do
    echo "== $cfg T=$t"
    PSSBAM_TILE_READS=$t python bench.py $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('reads/s %.3e  frac %.3f  ms/launch %.3f' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch']))"
  done
done
