mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_prove.py tests/test_gpu_pinned_proofs.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k: round(v, 2) for k, v in d['stage_ms'].items()})"; done
