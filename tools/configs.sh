# tools/configs.sh -- GPU-box helper: one bench line per named configuration (no CPU baseline)
for args in "--config C2" "--config C3 --unsorted --reads 100000000" "--config C4" "--config C4 --unsorted" "--config C5 --reads 100000000" "--config C1 --reads 20000000"; do
  echo "== $args"
  python bench.py $args --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('reads/s %.3e  GB/s %.0f frac %.3f  ms/launch %.3f  ok=%d filt=%d kmer_ok=%d slow=%d' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'], d['stats_last_step']['pss_ok'], d['stats_last_step']['pss_filtered'], d['stats_last_step']['kmer_ok'], d['stats_last_step']['slow_path']))"
done
