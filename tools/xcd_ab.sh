# tools/xcd_ab.sh -- GPU-box helper: workgroup->tile mapping A/B (round-robin vs XCD-contiguous)
for x in 0 1 0 1; do for args in "" "--config C2" "--config C3 --unsorted --reads 100000000" "--config C4"; do
  PSSBAM_XCD_MAP=$x python bench.py $args --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('xcd_map=$x %-45s reads/s %.4e  ms/launch %.4f' % ('$args', d['value'], d['roofline']['kernel_ms_per_launch']))"
done; done
