for i in 1 2 3; do python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps'], d['warmup'])"; done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('5/2:', d['value'], d['ms_per_step'])"
