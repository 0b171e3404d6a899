export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
cat > /tmp/r1.py <<PY
import sys, time, numpy as np
sys.path.insert(0, "$R")
from certificate_stark_amd.backend import Backend
from certificate_stark_amd.prover import ProofOptions, RangeProofExample
b = Backend()
opt = ProofOptions(42, 8, 0, 0, 0, 4, 256)
def t(f, reps):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e3
one = RangeProofExample(opt, 12345 << 3, b)
print("fresh context: range 64 = %.3f ms" % t(one.prove, 50))
w = np.random.default_rng(16).integers(0, 2**64, size=1024, dtype=np.uint64); w[-1] &= np.uint64(2**63 - 1)
print("range 2^16 = %.3f ms" % t(lambda: b.range_prove_bits(opt, w, 16), 10))
print("after it: range 64 = %.3f ms" % t(one.prove, 50))
print("again: range 64 = %.3f ms" % t(one.prove, 50))
PY
python3 /tmp/r1.py 2>&1 | tail -5
CSTARK_RANGE_GENERIC=1 python3 /tmp/r1.py 2>&1 | tail -4
CSTARK_RB_PROF=1 python3 /tmp/r1.py 2>&1 | tail -40 | head -36
