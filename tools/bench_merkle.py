"""Complete MerkleAir proofs of 512 transfers (2^18 rows), depth 15 (argv[1] = depth): time per proof and stage split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from certificate_stark_amd.backend import Backend
from certificate_stark_amd.prover import MerkleExample, ProofOptions, TransactionMetadata
b = Backend()
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 15
meta = TransactionMetadata.build_random(512, depth, seed=31)
ex = MerkleExample(ProofOptions(42, 8, 0, 0, 0, 4, 256), meta, b)
ex.prove(); ex.prove()
t0 = time.perf_counter()
for _ in range(7):
    p = ex.prove()
dt = (time.perf_counter() - t0) / 7
print("merkle d%d: %.3f ms per proof, %d bytes, stages %s" % (depth, dt * 1e3, len(p), {k: round(v, 2) for k, v in b.prove_stage_ms().items()}))
