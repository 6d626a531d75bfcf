# tools/sweep1.sh -- GPU-box helper: parity tests, then a tile-geometry sweep of bench.py
set -e
python -m pytest tests -m gpu -x -q > gpurun_out/t3.log 2>&1 || { tail -40 gpurun_out/t3.log; exit 1; }
tail -3 gpurun_out/t3.log
for T in ${SWEEP_T:-64 128 192}; do
  echo "== T=$T"
  PSSBAM_TILE_READS=$T python bench.py --reads ${SWEEP_READS:-50000000} --steps 5 --warmup 2 --no-cpu-baseline ${SWEEP_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('reads/s %.3e  GB/s %.0f frac %.3f  ms/launch %.3f' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch']))"
done
