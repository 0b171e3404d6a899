# counter passes over the forward transform kernels (tools/bench_ntt.py: interpolation + LDE of 94 x 2^20 x 8)
set -o pipefail
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; TAG=${1:-r03_ntt}
mkdir -p $O; rm -rf $O/pmc_$TAG
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $line --output-format csv -d $O/pmc_$TAG/p$i -- python3 $R/tools/bench_ntt.py 20 > $O/pmc_$TAG.p$i.log 2>&1) || { echo "pass $i failed: $line"; tail -5 $O/pmc_$TAG.p$i.log; }
done <<'LIST'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_64B_sum
SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_CSN_BUSY SPI_CSN_NUM_THREADGROUPS
LIST
python3 tools/pmc_passes.py $O/pmc_$TAG $O/${TAG}_pmc.csv k_ntt
