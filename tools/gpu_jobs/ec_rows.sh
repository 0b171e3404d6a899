mkdir -p gpurun_out
for rep in 1 2; do for v in 16 32 64; do
  CSTARK_EC_TILE_ROWS=$v python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows $v', d['value'], d['ms_per_step'], {k: round(v, 2) for k, v in d['stage_ms'].items()})"
done; done
