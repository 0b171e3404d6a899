# counter passes over the constraint stage (tools/bench_ce.py split): per kernel
set -o pipefail
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; TAG=${1:-r03_ce}; FILT=${2:-k_lin}
mkdir -p $O; rm -rf $O/pmc_$TAG
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $line --output-format csv -d $O/pmc_$TAG/p$i -- python3 $R/tools/bench_ce.py 20 2 split > $O/pmc_$TAG.p$i.log 2>&1) || { echo "pass $i failed: $line"; tail -5 $O/pmc_$TAG.p$i.log; }
done <<'LIST'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE
SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_ACTIVE_INST_VALU
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum
LIST
python3 tools/pmc_passes.py $O/pmc_$TAG $O/${TAG}_pmc.csv $FILT > /dev/null
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/${TAG}_pmc.csv")):
    ns=float(r['avg_ns_under_pmc']); w=float(r['SQ_WAVES'])
    print("%-28s %.3f ms clock %.2f GHz VALU/wave %6.0f SALU/wave %6.0f SMEM/wave %5.0f VMEM_RD/wave %4.0f branch/wave %5.0f wait_inst %.2f fetch %.2f GB (x2 %.2f) write %.2f GB L2 hit %.2f avg_rd_lat %.0f" % (
        r['kernel'], ns/1e6, float(r['GRBM_GUI_ACTIVE'])/8/ns, float(r['SQ_INSTS_VALU'])/w, float(r['SQ_INSTS_SALU'])/w, float(r['SQ_INSTS_SMEM'])/w,
        float(r['SQ_INSTS_VMEM_RD'])/w, float(r['SQ_INSTS_BRANCH'])/w, float(r['SQ_WAIT_INST_ANY'])/float(r['SQ_WAVE_CYCLES']),
        float(r['FETCH_SIZE'])*1024/1e9, float(r['FETCH_SIZE'])*2048/1e9, float(r['WRITE_SIZE'])*1024/1e9,
        float(r['TCC_HIT_sum'])/(float(r['TCC_HIT_sum'])+float(r['TCC_MISS_sum'])), float(r['TCP_TCC_READ_REQ_LATENCY_sum'])/max(1,float(r['TCP_TCC_READ_REQ_sum']))))
PY
