# the Rescue-window kernel on the matrix cores (rounds_mfma.hip) against the vector-ALU kernel: parity first, then an A/B on one box
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_constraints.py tests/test_gpu_composition.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -5 || exit 1
bash tools/gpu_jobs/ab_env.sh CSTARK_ROUNDS_MFMA=1 CSTARK_ROUNDS_MFMA=0 2>&1 | tee gpurun_out/r04_rounds_mfma_ab.txt
