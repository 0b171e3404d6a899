mkdir -p gpurun_out
time python bench.py > gpurun_out/${1:-r03_default}.json 2> gpurun_out/${1:-r03_default}.err; echo rc=$?
python3 - <<PY
import json
d=json.loads(open("gpurun_out/${1:-r03_default}.json").read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], d.get('two_proofs_in_flight'), d['cpu_baseline']['value'], [ (o['config'], o['ms_per_proof']) for o in d['other_configs']])
PY
timeout -k 10 300 python -m pytest tests/test_gpu_cpp_host.py -m gpu -q 2>&1 | tail -2
