mkdir -p gpurun_out
CSTARK_RB_PROF=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/rb_in_bench.json 2> gpurun_out/rb_in_bench.err; echo rc=$?
grep "cstark range batch" gpurun_out/rb_in_bench.err | sed -n 80,104p
