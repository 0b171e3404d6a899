# k_hash_rows with / without the next block's loads requested before the compression; the acc_mad forms micro-benchmark
mkdir -p gpurun_out
R=$PWD
{
for rep in 1 2; do
for v in hnp hpf; do echo "== [$v]"; CSTARK_LIB=$R/certificate-stark_amd/libcstark_hip_$v.so python3 tools/bench_hash.py | tail -1; done
done
echo "== parity [hpf]"; CSTARK_LIB=$R/certificate-stark_amd/libcstark_hip_hpf.so python3 -m pytest -q -x -m gpu tests/test_gpu_commit.py 2>&1 | tail -2
hipcc --offload-arch=gfx950 -O3 -o /tmp/accmad_bench tools/micro/accmad_bench.hip 2>/dev/null && /tmp/accmad_bench
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_hash_prefetch.txt
