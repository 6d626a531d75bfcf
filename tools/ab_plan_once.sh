# tools/ab_plan_once.sh -- GPU-box helper: tally_compact with the header decode + filters once per READ (plan through LDS, one more
# barrier per tile; PSSBAM_COMPACT_PLAN_ONCE=1) against once per LANE of the read's pair (=0), on the short-window configurations
for cfg in "--config C4" "--config C1 --reads 20000000" "--config C4 --unsorted" "--config C2 --region-len 15"; do
  for v in 0 1 0 1; do
    echo "== $cfg PSSBAM_COMPACT_PLAN_ONCE=$v"
    PSSBAM_COMPACT_PLAN_ONCE=$v python bench.py $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('reads/s %.3e  GB/s %.0f frac %.3f  ms/launch %.4f  ok=%d filt=%d slow=%d' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'], d['stats_last_step']['pss_ok'], d['stats_last_step']['pss_filtered'], d['stats_last_step']['slow_path']))" || exit 1
  done
done
