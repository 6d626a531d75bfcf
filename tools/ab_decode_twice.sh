# tools/ab_decode_twice.sh -- GPU-box helper: what the header decode + filters of tally_compact cost: the kernel as it
# is against a diagnostic build that does them TWICE per lane (PSSBAM_COMPACT_DECODE_TWICE); the difference is one decode pass per tile,
# i.e. twice what "decode once per read instead of once per lane" could save before its own hand-off costs
for cfg in "--config C4" "--config C1 --reads 20000000"; do
  for v in "" 1 "" 1; do
    echo "== $cfg PSSBAM_COMPACT_DECODE_TWICE=$v"
    if [ -n "$v" ]; then export PSSBAM_COMPACT_DECODE_TWICE=1; else unset PSSBAM_COMPACT_DECODE_TWICE; fi
    python bench.py $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('reads/s %.3e  GB/s %.0f frac %.3f  ms/launch %.4f  ok=%d filt=%d' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'], d['stats_last_step']['pss_ok'], d['stats_last_step']['pss_filtered']))" || exit 1
  done
done
