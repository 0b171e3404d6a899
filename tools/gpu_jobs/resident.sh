export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/r03_pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r03_pytest_gpu.log
[ $rc -eq 0 ] && python3 tools/bench_merkle.py 15 2>/dev/null | tail -1 && python3 tools/bench_merkle.py 31 2>/dev/null | tail -1 && python3 tools/bench_schnorr.py 2>/dev/null | tail -1 && bash tools/gpu_jobs/bench_default.sh r03_default
