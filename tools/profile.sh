#!/bin/bash
# Measurement set of one build, run on the GPU box:  bash tools/profile.sh <tag>      (e.g. r02_v1)
# Writes under gpurun_out/ (copy what is to be judged into profiles/):
#   <tag>_bench_prove.json          bench.py, default arguments (un-profiled: the number that counts)
#   <tag>_kernel_stats.csv          rocprofv3 --kernel-trace --stats of the same command
#   <tag>_hbm_traffic_pmc.csv       FETCH_SIZE / WRITE_SIZE per kernel, SEPARATE passes (they do not fit one pass); on gfx950
#                                   FETCH_SIZE reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM): the last
#                                   column is 2 * FETCH + WRITE
#   <tag>_valu_pmc.csv              SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES, SQ_WAVES, SQ_WAVE_CYCLES, GRBM_GUI_ACTIVE per kernel
#   <tag>_modmul.txt                tools/modmul_bench.py: raw Montgomery-product rate of the chip
#   <tag>_<config>_{kernel_stats,hbm_traffic_pmc,valu_pmc}.csv   the same three for merkle_2_18, schnorr_2_18, range_2_16 (tools/bench_air_one.py)
# rocprofv3 gets the program itself after `--` (python3 bench.py): no env / bash -c hop (the profiler initialises the GPU first).
set -o pipefail
TAG=${1:-r02}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
python3 bench.py > $O/${TAG}_bench_prove.json 2> $O/${TAG}_bench_prove.err || exit 1
python3 tools/modmul_bench.py > $O/${TAG}_modmul.txt 2>&1 || exit 1
rm -rf $O/prof_$TAG
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $O/${TAG}_bench_under_rocprof.json 2> $O/prof_$TAG.stats.err) || exit 1
cp $(ls $O/prof_$TAG/stats/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/prof_$TAG/pmc_$c -- python3 $R/bench.py $ARGS > $O/prof_$TAG.pmc_$c.log 2>&1) || exit 1
done
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_$TAG/pmc_valu -- python3 $R/bench.py $ARGS > $O/prof_$TAG.pmc_valu.log 2>&1) || exit 1
python3 tools/pmc_summary.py $O/prof_$TAG $O/$TAG 4   # --steps 3 --warmup 1 = 4 proofs per profiled run
# BASELINE's sub-AIR configurations (bench.py other_configs): kernel stats and the same counter passes over tools/bench_air_one.py,
# 4 proofs per run -> <tag>_<config>_{kernel_stats,hbm_traffic_pmc,valu_pmc}.csv (bench.py fills other_configs[*].roofline.traffic from them)
for CFG in merkle_2_18 schnorr_2_18 range_2_16; do
  P=$O/prof_${TAG}_$CFG
  rm -rf $P
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 $R/tools/bench_air_one.py $CFG 4 > $P.stats.log 2>&1) || exit 1
  cp $(ls $P/stats/*/*kernel_stats.csv | head -1) $O/${TAG}_${CFG}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/pmc_$c -- python3 $R/tools/bench_air_one.py $CFG 4 > $P.pmc_$c.log 2>&1) || exit 1
  done
  (cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/pmc_valu -- python3 $R/tools/bench_air_one.py $CFG 4 > $P.pmc_valu.log 2>&1) || exit 1
  python3 tools/pmc_summary.py $P $O/${TAG}_$CFG 4
done
