# full measurement set of the build (tools/profile.sh) plus the option-set and in-flight benches
TAG=${1:-r03_v1}
bash tools/profile.sh $TAG || exit 1
python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --mode hotpath > gpurun_out/${TAG}_bench_hotpath.json 2>/dev/null
python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --inflight 2 > gpurun_out/${TAG}_bench_prove_inflight2.json 2>/dev/null
for fe in quadratic cubic; do python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --field-extension $fe > gpurun_out/${TAG}_bench_$fe.json 2>/dev/null; done
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --hash-fn sha3 > gpurun_out/${TAG}_bench_sha3.json 2>/dev/null
python3 tools/bench_small_airs.py > gpurun_out/${TAG}_small_air_proofs.txt 2>/dev/null
ls -la gpurun_out | grep $TAG | head -30
