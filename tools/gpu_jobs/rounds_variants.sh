# stage timing of builds of rounds_mfma.hip (tools/build_variant_fast.py NAME rounds_mfma.hip -D...): OUT.txt name1 name2 ... ("base" = the regular library)
mkdir -p gpurun_out
R=$PWD
OUT=$1; shift
{
for rep in 1 2; do
for v in "$@"; do
  L=$R/certificate-stark_amd/libcstark_hip_$v.so; [ "$v" = base ] && L=$R/certificate-stark_amd/libcstark_hip.so
  echo "== [$v]"; CSTARK_LIB=$L python3 tools/bench_rounds.py 20 15 | tail -1
done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$OUT
