"""Complete SchnorrAir proofs of 512 signatures (2^18 rows): time per proof and stage split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from certificate_stark_amd.backend import Backend
from certificate_stark_amd.prover import ProofOptions, SchnorrExample
b = Backend()
ex = SchnorrExample.build_random(ProofOptions(42, 8, 0, 0, 0, 4, 256), int(sys.argv[1]) if len(sys.argv) > 1 else 512, seed=1, backend=b)
ex.prove(); ex.prove()
t0 = time.perf_counter()
for _ in range(5):
    p = ex.prove()
dt = (time.perf_counter() - t0) / 5
print("schnorr: %.3f ms per proof, %d bytes, stages %s" % (dt * 1e3, len(p), {k: round(v, 2) for k, v in b.prove_stage_ms().items()}))
