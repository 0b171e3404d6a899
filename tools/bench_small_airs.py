"""Times complete proofs of the standalone AIRs at BASELINE.json's sizes (configs 1-3): range (64 rows), merkle 512 transfers
= 2^18 rows (depth 15 and 31), schnorr 512 signatures = 2^18 rows.  Run on a GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from certificate_stark_amd.backend import Backend
from certificate_stark_amd.prover import (MerkleExample, ProofOptions, RangeProofExample, SchnorrExample, TransactionMetadata)

b = Backend()
opt = ProofOptions(42, 8, 0, 0, 0, 4, 256)


def timeit(name, f, reps=5):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(reps):
        p = f()
    dt = (time.perf_counter() - t0) / reps
    print("%-34s %8.3f ms/proof  %7.1f proofs/s  %d bytes   stages %s" % (
        name, dt * 1e3, 1 / dt, len(p), {k: round(v, 2) for k, v in b.prove_stage_ms().items()}))


timeit("range (64 rows)", RangeProofExample(opt, 12345 << 3, b).prove, reps=20)
full = TransactionMetadata.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "witness_1024_d15.npz"))
m512 = TransactionMetadata(*[getattr(full, f) if f == "final_root" else getattr(full, f)[:512] for f in TransactionMetadata.FIELDS])
m512.final_root = full.initial_roots[512].copy()
timeit("merkle 512 tx, depth 15 (2^18 rows)", MerkleExample(opt, m512, b).prove)
# depth 31 (the nearest legal value to BASELINE's "depth 32") is supported by the kernels (8*31+7 = 255 <= 511 rows) but the
# dense host witness generator stops at depth 24; its trace / constraint parity is covered at depths 3, 7, 15.
t0 = time.perf_counter()
ex = SchnorrExample.build_random(opt, 512, seed=1, backend=b)
print("schnorr witness synthesis 512 sigs: %.1f s (host, serial)" % (time.perf_counter() - t0))
timeit("schnorr 512 signatures (2^18 rows)", ex.prove)
