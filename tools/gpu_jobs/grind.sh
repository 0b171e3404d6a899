export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_prove.py -m gpu -x -q > $O/r03_grind_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r03_grind_pytest.log
[ $rc -eq 0 ] && for g in 16 20; do for e in 1 0; do CSTARK_GRIND_DEVICE=$e python3 - <<PY
import sys, time
sys.path.insert(0, "$R")
from certificate_stark_amd.backend import Backend
from certificate_stark_amd.prover import ProofOptions, TransactionMetadata, TransactionProver
b = Backend()
p = TransactionProver(ProofOptions(96, 8, $g, 0, 0, 4, 256), b)
p.load_witness(TransactionMetadata.build_random(1024, 15, seed=7))
p.prove(); p.prove()
t0 = time.perf_counter()
for _ in range(5): proof = p.prove()
print("grinding $g bits, CSTARK_GRIND_DEVICE=$e: %.2f ms per proof of 2^20 steps" % ((time.perf_counter() - t0) / 5 * 1e3))
PY
done; done
