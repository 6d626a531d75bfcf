# tools/gridscan.sh -- GPU-box helper: kernel time vs number of resident workgroups
for g in 256 512 768; do for a in 0 4; do
  echo "== grid=$g ablate=$a"
  PSSBAM_GRID_WGS=$g PSSBAM_ABLATE=$a python bench.py --reads 50000000 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('reads/s %.3e  GB/s %.0f  ms/launch %.3f' % (d['value'], d['roofline']['achieved'], d['roofline']['kernel_ms_per_launch']))"
done; done
