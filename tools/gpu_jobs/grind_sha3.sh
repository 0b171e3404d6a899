export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_pinned_proofs.py tests/test_gpu_prove.py -m gpu -x -q > $O/r03_grind_sha3_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r03_grind_sha3_pytest.log
[ $rc -eq 0 ] && timeout -k 10 300 bash tools/gpu_jobs/options_sweep.sh 2>&1 | tail -3
