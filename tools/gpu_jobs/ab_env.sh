# A/B of one environment switch on the same box: bash tools/gpu_jobs/ab_env.sh VAR=1 [VAR=0]
mkdir -p gpurun_out
A=${1:-X=1}; B=${2:-X=0}
for rep in 1 2 3; do for v in "$A" "$B"; do
  env $v python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], {k: round(v, 2) for k, v in d['stage_ms'].items()})"
done; done
