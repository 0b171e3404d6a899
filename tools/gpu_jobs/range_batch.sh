mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_pinned_proofs.py -m gpu -x -q -k "batched" 2>&1 | tail -15
timeout -k 10 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_range_batch.txt
import sys, time
sys.path.insert(0, '.')
import numpy as np
from certificate_stark_amd.backend import Backend
from certificate_stark_amd.prover import ProofOptions, RangeProofExample
b = Backend(); opt = ProofOptions(42, 8, 0, 0, 0, 4, 256)
nums = [(12345 + i) << 3 for i in range(1024)]
for B in (64, 256, 1024, 4096):
    ns = (nums * 4)[:B]
    b.range_prove_batch(opt, ns)
    t0 = time.perf_counter()
    for _ in range(5):
        pr = b.range_prove_batch(opt, ns)
    dt = (time.perf_counter() - t0) / 5
    print("batch %5d: %8.3f ms per call, %7.4f ms per proof, %9.0f proofs/s" % (B, dt * 1e3, dt * 1e3 / B, B / dt))
t0 = time.perf_counter()
for v in nums[:256]:
    RangeProofExample(opt, v, b).prove()
print("one by one: %.4f ms per proof" % ((time.perf_counter() - t0) / 256 * 1e3))
PY
