R=$PWD; P=$R/certificate-stark_amd
g++ -std=c++17 -O2 -I $R/include $R/tools/cpp/bench_pool.cpp -o /tmp/bench_pool -pthread -L $P -lcstark_hip -Wl,-rpath,$P -Wl,-rpath,/opt/rocm/lib || exit 1
/tmp/bench_pool 24
