CSTARK_RB_PROF=1 timeout -k 10 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -24
import sys, time
sys.path.insert(0, '.')
from certificate_stark_amd.backend import Backend
from certificate_stark_amd.prover import ProofOptions
b = Backend(); opt = ProofOptions(42, 8, 0, 0, 0, 4, 256)
nums = [(12345 + i) << 3 for i in range(1024)]
b.range_prove_batch(opt, nums)
b.range_prove_batch(opt, nums)
sys.stderr.write("---- third call\n")
t0 = time.perf_counter(); b.range_prove_batch(opt, nums); print("python call %.3f ms" % ((time.perf_counter() - t0) * 1e3))
PY
