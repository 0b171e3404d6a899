export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/prof_schnorr
python3 tools/bench_schnorr.py 2>/dev/null | tail -1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_schnorr -- python3 $R/tools/bench_schnorr.py > $O/prof_schnorr.log 2>&1)
python3 - <<PY
import csv, glob
f=glob.glob("$O/prof_schnorr/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:18]:
    n=r['Name'].replace('void ','').replace('cs::(anonymous namespace)::','').split('(')[0][:44]
    print("%-46s calls %4s avg %9.3f ms total/7 %8.3f  %5s%%" % (n, r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/7e6, r['Percentage']))
PY
