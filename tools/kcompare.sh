# tools/kcompare.sh -- GPU-box helper: tiled vs gather kernel on the named configurations
for args in "--config C3 --reads 50000000" "--config C3 --reads 50000000 --unsorted" "--config C4" "--config C5 --reads 50000000" "--config C1 --reads 20000000"; do
 for k in 2 3; do
  echo "== kernel=$k $args"
  python bench.py $args --kernel $k --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('reads/s %.3e  GB/s %.0f frac %.3f  ms/launch %.3f  ok=%d' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'], d['stats_last_step']['pss_ok']))"
 done
done
