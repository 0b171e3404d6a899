# A/B of k_lin_all builds (tools/build_variant_fast.py NAME constraints.hip -D...): stage timing on random data + parity of the stage
# usage: bash tools/gpu_jobs/lin_variants.sh OUT.txt name1 name2 ...   ("base" = the regular library)
mkdir -p gpurun_out
R=$PWD
OUT=$1; shift
{
for rep in 1 2; do
for v in "$@"; do
  L=$R/certificate-stark_amd/libcstark_hip_$v.so; [ "$v" = base ] && L=$R/certificate-stark_amd/libcstark_hip.so
  echo "== k_lin_all [$v]"; CSTARK_LIB=$L python3 tools/bench_ce.py 20 5 split | tail -2
done
done
for v in "$@"; do
  [ "$v" = base ] && continue
  echo "== parity [$v]"; CSTARK_LIB=$R/certificate-stark_amd/libcstark_hip_$v.so python3 -m pytest -q -x -m gpu tests/test_gpu_constraints.py tests/test_gpu_composition.py 2>&1 | tail -2
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$OUT
