#!/bin/bash
# usage: tools/run_variants.sh <kernel-name-filter> variant...   (on the GPU box; libs built by build.build_variant)
export TMPDIR=/tmp
R=$PWD
filt=$1; shift
for v in "$@"; do
  d=$R/gpurun_out/var_$v
  rm -rf $d
  (cd /tmp && CSTARK_LIB=$R/certificate-stark_amd/libcstark_hip_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/bench_ce.py 20 3 ${CE_MODE:-} > $d.log 2>&1)
  echo "== $v: $(tail -1 $d.log | grep -o 'constraints.*')"
  cut -d, -f1,4 $d/*/*kernel_stats.csv | grep "$filt" | sed 's/void cs::(anonymous namespace):://' | cut -c1-90
done
