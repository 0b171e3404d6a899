mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r03_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
