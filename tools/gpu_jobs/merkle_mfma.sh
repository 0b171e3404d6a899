# sub-AIR round gadgets on the matrix cores: parity (whole proofs against the CPU prover and the golden digests), then an A/B
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_small_airs.py tests/test_gpu_prove_small_airs.py tests/test_gpu_pinned_proofs.py tests/test_gpu_options.py -m gpu -x -q 2>&1 | tail -3 || exit 1
{
for rep in 1 2; do for v in 1 0; do for cfg in merkle_2_18 schnorr_2_18; do
  echo -n "CSTARK_ROUNDS_MFMA=$v "; CSTARK_ROUNDS_MFMA=$v python3 tools/bench_air_one.py $cfg 40 | tail -1
done; done; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_merkle_mfma_ab.txt
