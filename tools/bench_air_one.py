"""One sub-AIR configuration of BASELINE.json proved N times -- the program the counter passes of tools/profile.sh run under rocprofv3:
    python tools/bench_air_one.py merkle_2_18|schnorr_2_18|range_2_16|rescue_2_12 [proofs]
Witnesses: the product's own seeded generators with the seeds of tools/proof_configs.py (the proofs tests/test_gpu_pinned_proofs.py pins)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from certificate_stark_amd.backend import Backend
from certificate_stark_amd.prover import MerkleExample, ProofOptions, RescueExample, SchnorrExample, TransactionMetadata

cfg = sys.argv[1]
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
b = Backend()
opt = ProofOptions(42, 8, 0, 0, 0, 4, 256)
if cfg == "merkle_2_18":
    full = TransactionMetadata.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "witness_1024_d15.npz"))
    m512 = TransactionMetadata(*[getattr(full, f) if f == "final_root" else getattr(full, f)[:512] for f in TransactionMetadata.FIELDS])
    m512.final_root = full.initial_roots[512].copy()
    prove = MerkleExample(opt, m512, b).prove
elif cfg == "schnorr_2_18":
    prove = SchnorrExample.build_random(opt, 512, seed=1, backend=b).prove
elif cfg == "range_2_16":
    words = np.random.default_rng(16).integers(0, 2**64, size=(1 << 16) // 64, dtype=np.uint64)
    words[-1] &= np.uint64(2**63 - 1)
    prove = lambda: b.range_prove_bits(opt, words, 16)
elif cfg == "rescue_2_12":
    prove = RescueExample(512, ProofOptions(42, 4, 0, 0, 0, 4, 256), b).prove  # benches/rescue.rs:370-378
else:
    raise SystemExit("unknown configuration")
t0 = time.perf_counter()
for _ in range(proofs):
    p = prove()
print("%s: %d proofs, %.3f ms each (first one included), %d bytes" % (cfg, proofs, (time.perf_counter() - t0) / proofs * 1e3, len(p)))
b.close()
