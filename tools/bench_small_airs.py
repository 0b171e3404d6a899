"""Times complete proofs of the standalone AIRs at BASELINE.json's sizes (configs 1-3): range (the reference's 64-row proof, the
synthetic 2^16-row accumulator, and 1024 64-row proofs back to back), merkle 512 transfers = 2^18 rows (depth 15 and depth 31, the
nearest legal value to BASELINE's "depth 32"), schnorr 512 signatures = 2^18 rows.  Run on a GPU box."""
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
words = np.random.default_rng(16).integers(0, 2**64, size=(1 << 16) // 64, dtype=np.uint64)
words[-1] &= np.uint64(2**63 - 1)
timeit("range, synthetic 2^16-row accumulator", lambda: b.range_prove_bits(opt, words, 16), reps=10)
t0 = time.perf_counter()
for i in range(1024):
    p = RangeProofExample(opt, (12345 + i) << 3, b).prove()
dt = time.perf_counter() - t0
print("%-34s %8.3f ms for 1024 proofs of 64 rows (2^16 rows in total)  %7.1f proofs/s" % ("range, 1024 x 64 rows", dt * 1e3, 1024 / dt))
# The same 1024 proofs from several host threads, each with its own context and stream (the calls release the interpreter lock).
# Measured on MI355X: 503 ms with one prover, 525 / 542 ms with 4 / 8 in flight -- a 64-row proof is ~150 tiny launches and host
# round trips, and the HIP runtime serialises them across threads: the small proofs are bound by host API time (0.49 ms each), not
# by the GPU.  A batched prover (one launch per stage for all proofs) is the way past that; see DESIGN.md 7.
from concurrent.futures import ThreadPoolExecutor
for workers in (4,):
    backs = [Backend() for _ in range(workers)]
    for bk in backs:
        RangeProofExample(opt, 12345 << 3, bk).prove()

    def run(wid, backs=backs, workers=workers):
        for i in range(wid, 1024, workers):
            RangeProofExample(opt, (12345 + i) << 3, backs[wid]).prove()

    t0 = time.perf_counter()
    with ThreadPoolExecutor(workers) as ex_:
        list(ex_.map(run, range(workers)))
    dt = time.perf_counter() - t0
    print("%-34s %8.3f ms for 1024 proofs of 64 rows, %d provers in flight on one GPU  %7.1f proofs/s" % (
        "range, 1024 x 64 rows", dt * 1e3, workers, 1024 / dt))
    del backs
full = TransactionMetadata.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "witness_1024_d15.npz"))
m512 = TransactionMetadata(*[getattr(full, f) if f == "final_root" else getattr(full, f)[:512] for f in TransactionMetadata.FIELDS])
m512.final_root = full.initial_roots[512].copy()
timeit("merkle 512 tx, depth 15 (2^18 rows)", MerkleExample(opt, m512, b).prove)
m31 = TransactionMetadata.build_random(512, 31, seed=31)  # sparse account tree
timeit("merkle 512 tx, depth 31 (2^18 rows)", MerkleExample(opt, m31, b).prove)
t0 = time.perf_counter()
ex = SchnorrExample.build_random(opt, 512, seed=1, backend=b)
print("schnorr witness synthesis 512 sigs: %.1f s (host, serial)" % (time.perf_counter() - t0))
timeit("schnorr 512 signatures (2^18 rows)", ex.prove)
