mkdir -p gpurun_out
python bench.py --steps 5 --warmup 2 ${BENCH_ARGS:-} > gpurun_out/${1:-r03_bench}.json 2> gpurun_out/${1:-r03_bench}.err; echo rc=$?
tail -c 3000 gpurun_out/${1:-r03_bench}.json; tail -3 gpurun_out/${1:-r03_bench}.err
