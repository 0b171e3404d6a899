mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sharding.py -m gpu -x -q 2>&1 | tail -8
{ python tools/bench_shard_sim.py; CSTARK_SHARD_SPLIT=0 python tools/bench_shard_sim.py; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_shard_sim.txt
