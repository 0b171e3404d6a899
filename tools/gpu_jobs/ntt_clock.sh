# The transform kernels with one phase compiled out (tools/build_variant_fast.py: CS_NTT_SKIP / CS_NTT_NOMATH), un-profiled time and,
# under one counter pass, the clock the kernel ran at (GRBM_GUI_ACTIVE / 8 XCDs / duration) and its vector-instruction count.
set -o pipefail
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; TAG=${1:-r03_ntt_clock}
mkdir -p $O
for v in "" skip24 skip31 nomath; do
  lib=$R/certificate-stark_amd/libcstark_hip${v:+_$v}.so
  echo "== variant [$v]: $(CSTARK_LIB=$lib python3 tools/bench_ntt.py 20 2>/dev/null | tail -1)"
  rm -rf $O/pmc_${TAG}_$v
  (cd /tmp && CSTARK_LIB=$lib rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${TAG}_$v/p1 -- python3 $R/tools/bench_ntt.py 20 > $O/pmc_${TAG}_$v.log 2>&1) || tail -3 $O/pmc_${TAG}_$v.log
  python3 tools/pmc_passes.py $O/pmc_${TAG}_$v $O/${TAG}_${v:-default}_pmc.csv "k_ntt" | python3 -c "
import sys, csv
for r in csv.DictReader(sys.stdin):
    ns=float(r['avg_ns_under_pmc']); g=float(r['GRBM_GUI_ACTIVE'])
    print('   %-34s %8.3f ms  clock %.2f GHz  VALU insts/wave %6.0f  wait_inst/wave_cycles %.2f' % (r['kernel'], ns/1e6, g/8/ns, float(r['SQ_INSTS_VALU'])/float(r['SQ_WAVES']), float(r['SQ_WAIT_INST_ANY'])/float(r['SQ_WAVE_CYCLES'])))"
done 2>&1 | tee $O/${TAG}.txt
python3 tools/modmul_bench.py 2>/dev/null | tail -3 | tee -a $O/${TAG}.txt
