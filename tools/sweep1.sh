set -e
python -m pytest tests -m gpu -x -q > gpurun_out/t3.log 2>&1 || { tail -40 gpurun_out/t3.log; exit 1; }
tail -3 gpurun_out/t3.log
for cfg in "2 128" "2 64" "1 128" "1 64" "1 192" "1 256" "2 256"; do
  set -- $cfg
  echo "== bufs=$1 T=$2"
  PSSBAM_STAGE_BUFS=$1 PSSBAM_TILE_READS=$2 python bench.py --reads 50000000 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('reads/s %.3e  GB/s %.0f frac %.3f  ms/launch %.3f' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch']))"
done
