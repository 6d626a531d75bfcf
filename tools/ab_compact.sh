# tools/ab_compact.sh -- GPU-box helper: tally_compact (N <= 16) vs tally_tiled on the short-window configurations
for cfg in "--config C4" "--config C1 --reads 20000000" "--config C4 --unsorted" "--config C2 --region-len 15"; do
  for c in 1 0; do
    echo "== $cfg PSSBAM_COMPACT=$c"
    PSSBAM_COMPACT=$c python bench.py $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('reads/s %.3e  GB/s %.0f frac %.3f  ms/launch %.3f  ok=%d filt=%d slow=%d' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'], d['stats_last_step']['pss_ok'], d['stats_last_step']['pss_filtered'], d['stats_last_step']['slow_path']))"
  done
done
