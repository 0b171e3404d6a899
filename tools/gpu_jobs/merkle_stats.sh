export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/prof_merkle
python -m pytest tests/test_gpu_pinned_proofs.py tests/test_gpu_prove_small_airs.py tests/test_gpu_baseline_configs.py -m gpu -x -q -k "range or batch" > $O/r03_range_single.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r03_range_single.log
python3 - <<PY
import time, sys
sys.path.insert(0, "$R")
from certificate_stark_amd.backend import Backend
from certificate_stark_amd.prover import ProofOptions, RangeProofExample
b = Backend()
one = RangeProofExample(ProofOptions(42, 8, 0, 0, 0, 4, 256), 12345 << 3, b)
one.prove(); one.prove()
t0 = time.perf_counter()
for _ in range(50): p = one.prove()
print("range 64 rows, one call per proof: %.3f ms, %d bytes" % ((time.perf_counter() - t0) / 50 * 1e3, len(p)))
PY
python3 tools/bench_merkle.py 2>/dev/null | tail -1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_merkle -- python3 $R/tools/bench_merkle.py > $O/prof_merkle.log 2>&1)
python3 - <<PY
import csv, glob
f=glob.glob("$O/prof_merkle/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:22]:
    n=r['Name'].replace('void ','').replace('cs::(anonymous namespace)::','').split('(')[0][:44]
    print("%-46s calls %4s avg %9.3f ms total/9 %8.3f  %5s%%" % (n, r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/9e6, r['Percentage']))
PY
