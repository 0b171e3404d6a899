export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "merkle or Merkle or schnorr or Schnorr or range or pinned or small_air or baseline or air" > $O/r03_air_combine_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r03_air_combine_pytest.log
[ $rc -eq 0 ] && for e in 1 0; do echo "CSTARK_AIR_INV_TABLES=$e"; CSTARK_AIR_INV_TABLES=$e python3 tools/bench_merkle.py 15 2>/dev/null | tail -1; CSTARK_AIR_INV_TABLES=$e python3 tools/bench_schnorr.py 2>/dev/null | tail -1; done
