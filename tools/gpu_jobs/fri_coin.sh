export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/r03_pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r03_pytest_gpu.log
[ $rc -eq 0 ] && bash tools/gpu_jobs/ab_env.sh CSTARK_FRI_DEVICE_COIN=1 CSTARK_FRI_DEVICE_COIN=0 && for e in 1 0; do echo "CSTARK_FRI_DEVICE_COIN=$e"; CSTARK_FRI_DEVICE_COIN=$e python3 tools/bench_merkle.py 15 2>/dev/null | tail -1; CSTARK_FRI_DEVICE_COIN=$e python3 tools/bench_schnorr.py 2>/dev/null | tail -1; done
