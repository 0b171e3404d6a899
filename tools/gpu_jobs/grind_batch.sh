export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_pinned_proofs.py tests/test_gpu_prove.py tests/test_gpu_cpp_host.py -m gpu -x -q > $O/r03_grind_batch_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r03_grind_batch_pytest.log
[ $rc -eq 0 ] && for e in 1 0; do CSTARK_GRIND_DEVICE=$e python3 - <<PY
import sys, time
sys.path.insert(0, "$R")
from certificate_stark_amd.backend import Backend
from certificate_stark_amd.prover import ProofOptions
b = Backend()
nums = [(12345 + i) << 3 for i in range(1024)]
for g in (12, 16):
    opt = ProofOptions(42, 8, g, 0, 0, 4, 256)
    b.range_prove_batch(opt, nums[:64])
    t0 = time.perf_counter()
    p = b.range_prove_batch(opt, nums)
    print("1024 range proofs, grinding %d bits, CSTARK_GRIND_DEVICE=$e: %.1f ms" % (g, (time.perf_counter() - t0) * 1e3))
PY
done
