# tools/ablate.sh -- GPU-box helper: which phase of tally_tiled costs what (results are wrong when ablated)
# bits: 1 no COLUMNS, 2 no position loop, 4 no CODES, 8 no k-mer atomic, 16 no k-mer work at all, 128 no reference-window gathers
for a in ${ABLATE_SET:-0 1 2 3 4 5}; do
  echo "== ablate=$a"
  PSSBAM_ABLATE=$a python bench.py --reads 50000000 --steps 5 --warmup 2 --no-cpu-baseline ${ABLATE_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('reads/s %.3e  GB/s %.0f  ms/launch %.3f' % (d['value'], d['roofline']['achieved'], d['roofline']['kernel_ms_per_launch']))"
done
