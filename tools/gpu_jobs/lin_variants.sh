mkdir -p gpurun_out
R=$PWD
{
for rep in 1 2; do
echo "== old three launches"; CSTARK_LIN_MERGED=0 python3 tools/bench_ce.py 20 5 split | tail -2
for v in "" linw2 linw4 linu7 linu7b; do
  echo "== merged [$v]"; CSTARK_LIB=$R/certificate-stark_amd/libcstark_hip${v:+_$v}.so python3 tools/bench_ce.py 20 5 split | tail -2
done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_lin_variants.txt
