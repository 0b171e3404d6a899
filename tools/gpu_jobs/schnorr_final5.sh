export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "schnorr or Schnorr or pinned or small_air or baseline or alternative" > $O/r03_schnorr_final5_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r03_schnorr_final5_pytest.log
[ $rc -eq 0 ] && for e in 1 0 1 0; do echo "CSTARK_SCHNORR_FINAL5=$e"; CSTARK_SCHNORR_FINAL5=$e python3 tools/bench_schnorr.py 2>/dev/null | tail -1; done
