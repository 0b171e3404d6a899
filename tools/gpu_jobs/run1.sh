set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_t1_pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/r03_t1_rc.txt
tail -5 gpurun_out/r03_t1_pytest.log
for v in "" "CSTARK_LDE_BATCH_MB=0" "CSTARK_NTT_GROUP=1" "CSTARK_NTT_GROUP=2" "CSTARK_NTT_GROUP=4" "CSTARK_NTT_GROUP=8" "CSTARK_NTT_GROUP=16"; do
  echo "== $v" ; env $v python tools/bench_ntt.py 20
done 2>&1 | tee gpurun_out/r03_t1_ntt.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_t1_bench.json 2> gpurun_out/r03_t1_bench.err; tail -c 1500 gpurun_out/r03_t1_bench.json
CSTARK_LDE_BATCH_MB=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_t1_bench_old.json 2>> gpurun_out/r03_t1_bench.err; tail -c 600 gpurun_out/r03_t1_bench_old.json
