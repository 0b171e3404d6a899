export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "merkle or Merkle or pinned or small_air or baseline" > $O/r03_merkle_rounds_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r03_merkle_rounds_pytest.log
[ $rc -eq 0 ] && for e in 1 0; do for d in 15 31; do echo "CSTARK_MERKLE_ROUNDS=$e"; CSTARK_MERKLE_ROUNDS=$e python3 tools/bench_merkle.py $d 2>/dev/null | tail -1; done; done
