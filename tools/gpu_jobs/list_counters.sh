mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r03_counters.txt 2>&1
wc -l $GRAFT_REPO_ROOT/gpurun_out/r03_counters.txt
