#!/bin/bash
# kernel trace of a few proofs and the idle-time report of the last one:  bash tools/gpu_jobs/timeline.sh <tag> [env assignments...]
set -o pipefail
TAG=${1:-timeline}; shift
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
cp "$F" "$O/${TAG}_kernel_trace.csv"
rm -rf $O/tl_$TAG
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/tl_$TAG -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs > $O/${TAG}_bench.json 2> $O/${TAG}.err) || exit 1
F=$(ls $O/tl_$TAG/*/*kernel_trace.csv | head -1)
python3 tools/timeline.py $F > $O/${TAG}_timeline.txt
cat $O/${TAG}_timeline.txt
cp "$F" "$O/${TAG}_kernel_trace.csv"
rm -rf $O/tl_$TAG
