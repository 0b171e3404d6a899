#!/bin/bash
# HBM traffic of the hot-path kernels from PMC counters (separate passes, as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE do not fit one pass; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read stream).
export TMPDIR=/tmp
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$c.log 2>&1)
done
python3 - <<'PY'
import csv, glob, collections, os
R=os.getcwd()
tot=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(int)
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(R+"/gpurun_out/pmc_%s/*/*counter_collection.csv"%c)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]!=c: continue
        k=r["Kernel_Name"].replace("void ","").replace("cs::(anonymous namespace)::","")[:48]
        tot[k][c]+=float(r["Counter_Value"]); 
        if c=="FETCH_SIZE": cnt[k]+=1
print("kernel,dispatches,FETCH_SIZE_KB_per_dispatch,WRITE_SIZE_KB_per_dispatch,fetch_x2_plus_write_GB_per_dispatch")
for k in sorted(tot, key=lambda k:-tot[k]["FETCH_SIZE"]):
    n=max(cnt[k],1); f=tot[k]["FETCH_SIZE"]/n; w=tot[k]["WRITE_SIZE"]/n
    print("%s,%d,%.0f,%.0f,%.3f"%(k,n,f,w,(2*f+w)*1024/1e9))
PY
