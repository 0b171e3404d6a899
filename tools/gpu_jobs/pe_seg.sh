mkdir -p gpurun_out; P=$PWD/certificate-stark_amd
for rep in 1 2; do for v in "" _seg64k _seg256k; do
  CSTARK_LIB=$P/libcstark_hip$v.so python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib$v', d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['stage_ms'].items() if k in ('ood','deep','fri','composition')})"
done; done
