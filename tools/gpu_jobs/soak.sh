# one-off robustness run: many proofs of every configuration, every proof's digest must equal the first one's
export TMPDIR=/tmp; R=$PWD
python3 - <<PY
import sys, hashlib, time
sys.path.insert(0, "$R")
import numpy as np
from certificate_stark_amd.backend import Backend
from certificate_stark_amd.prover import (MerkleExample, ProofOptions, RangeProofExample, SchnorrExample, TransactionMetadata, TransactionProver)
b = Backend()
opt = ProofOptions(42, 8, 0, 0, 0, 4, 256)
def soak(name, prove, reps):
    want = hashlib.sha256(prove()).hexdigest()
    t0 = time.time()
    for i in range(reps):
        got = hashlib.sha256(prove()).hexdigest()
        assert got == want, (name, i)
    print("%-28s %4d proofs identical (%s...) in %.1f s" % (name, reps + 1, want[:16], time.time() - t0), flush=True)
meta = TransactionMetadata.load("$R/tests/golden/tx_metadata_1024.npz") if False else TransactionMetadata.build_random(1024, 15, seed=7)
p = TransactionProver(ProofOptions(num_queries=96), b); p.load_witness(meta)
soak("tx 2^20", p.prove, 300)
m = TransactionMetadata.build_random(512, 15, seed=31)
soak("merkle 2^18", MerkleExample(opt, m, b).prove, 300)
soak("schnorr 2^18", SchnorrExample.build_random(opt, 512, seed=1, backend=b).prove, 300)
soak("range 64", RangeProofExample(opt, 12345 << 3, b).prove, 1000)
nums = [(12345 + i) << 3 for i in range(1024)]
want = [hashlib.sha256(q).hexdigest() for q in b.range_prove_batch(opt, nums)]
for i in range(50):
    assert [hashlib.sha256(q).hexdigest() for q in b.range_prove_batch(opt, nums)] == want, i
print("range batch 1024 x 51 identical", flush=True)
b.close()
PY
