# tools/ablate.sh -- GPU-box helper: which phase of tally_tiled costs what (results are wrong when ablated)
for a in 0 1 2 3 4 5; do
  echo "== ablate=$a"
  PSSBAM_ABLATE=$a python bench.py --reads 50000000 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('reads/s %.3e  GB/s %.0f  ms/launch %.3f' % (d['value'], d['roofline']['achieved'], d['roofline']['kernel_ms_per_launch']))"
done
