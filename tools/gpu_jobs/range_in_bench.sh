mkdir -p gpurun_out
for e in 0 1; do CSTARK_RANGE_GENERIC=$e python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GENERIC=$e', [(o['config'], o['ms_per_proof']) for o in d['other_configs']])"; done
