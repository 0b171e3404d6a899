export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "merkle or Merkle or pinned or small_air or baseline or alternative or air" > $O/r03_merkle_sections_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r03_merkle_sections_pytest.log
[ $rc -eq 0 ] && for i in 1 2; do python3 tools/bench_merkle.py 15 2>/dev/null | tail -1; python3 tools/bench_merkle.py 31 2>/dev/null | tail -1; done
