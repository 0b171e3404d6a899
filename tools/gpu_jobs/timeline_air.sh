#!/bin/bash
# kernel trace of a few sub-AIR proofs and the idle-time report of the last one:  bash tools/gpu_jobs/timeline_air.sh <config> [env assignments...]
set -o pipefail
CFG=${1:-merkle_2_18}; shift
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
rm -rf $O/tl_$CFG
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/tl_$CFG -- python3 $R/tools/bench_air_one.py $CFG 6 > $O/tl_${CFG}.log 2> $O/tl_${CFG}.err) || exit 1
F=$(ls $O/tl_$CFG/*/*kernel_trace.csv | head -1)
python3 tools/timeline.py "$F" > $O/r04_timeline_${CFG}.txt
cat $O/tl_${CFG}.log $O/r04_timeline_${CFG}.txt
rm -rf $O/tl_$CFG
