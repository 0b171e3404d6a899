mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pinned_proofs.py tests/test_gpu_prove_small_airs.py tests/test_gpu_small_airs.py -m gpu -x -q -k "schnorr or sub_air" 2>&1 | tail -6
for v in 1 0; do echo "CSTARK_SCHNORR_SPLIT=$v"; CSTARK_SCHNORR_SPLIT=$v python3 tools/bench_schnorr.py 2>/dev/null | tail -1; done | tee gpurun_out/r03_schnorr_split.txt
