mkdir -p gpurun_out
export CSTARK_BENCH_REHEARSE=1
timeout -k 10 300 python bench.py --gpus 2 --steps 4 --warmup 1 > gpurun_out/r04_rehearse_replica2.json 2> gpurun_out/r04_rehearse_replica2.err; echo "replica rc=$?"; tail -c 600 gpurun_out/r04_rehearse_replica2.json; tail -3 gpurun_out/r04_rehearse_replica2.err
timeout -k 10 300 python bench.py --gpus 2 --steps 4 --warmup 1 --mode shard > gpurun_out/r04_rehearse_shard2.json 2> gpurun_out/r04_rehearse_shard2.err; echo "shard rc=$?"; tail -c 600 gpurun_out/r04_rehearse_shard2.json; tail -3 gpurun_out/r04_rehearse_shard2.err
