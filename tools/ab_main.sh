# tools/ab_main.sh -- GPU-box helper: the long-window configurations (tally_tiled) before/after a kernel change
for cfg in "--config C3" "--config C2" "--config C5 --reads 100000000" "--config C3 --unsorted --reads 100000000" "--config C3 --region-len 62 --reads 50000000"; do
  echo "== $cfg"
  python bench.py $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('reads/s %.3e  GB/s %.0f frac %.3f  ms/launch %.3f  ok=%d kmer_ok=%d' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'], d['stats_last_step']['pss_ok'], d['stats_last_step']['kmer_ok']))"
done
