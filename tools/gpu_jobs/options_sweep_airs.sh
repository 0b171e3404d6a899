export TMPDIR=/tmp; R=$PWD
python3 - <<PY
import sys, time
sys.path.insert(0, "$R")
import numpy as np
from certificate_stark_amd.backend import Backend
from certificate_stark_amd.prover import MerkleExample, ProofOptions, RangeProofExample, SchnorrExample, TransactionMetadata
b = Backend()
meta = TransactionMetadata.build_random(512, 15, seed=31)
w16 = np.random.default_rng(16).integers(0, 2**64, size=1024, dtype=np.uint64); w16[-1] &= np.uint64(2**63 - 1)
for o in [(42, 8, 0, 0, 0, 4, 256), (42, 8, 16, 0, 0, 4, 256), (42, 8, 0, 1, 0, 4, 256), (42, 8, 16, 1, 0, 4, 256), (42, 8, 0, 0, 1, 4, 256), (42, 8, 0, 0, 2, 4, 256)]:
    opt = ProofOptions(*o)
    for name, ex in (("merkle 2^18", MerkleExample(opt, meta, b)), ("schnorr 2^18", SchnorrExample.build_random(opt, 512, seed=1, backend=b)),
                     ("range 64", RangeProofExample(opt, 12345 << 3, b))):
        ex.prove(); ex.prove()
        t0 = time.perf_counter()
        for _ in range(3): proof = ex.prove()
        print(o, "%-13s %8.3f ms %7d bytes" % (name, (time.perf_counter() - t0) / 3 * 1e3, len(proof)), flush=True)
    if o[4] == 0:
        b.range_prove_bits(opt, w16, 16); t0 = time.perf_counter()
        for _ in range(3): proof = b.range_prove_bits(opt, w16, 16)
        print(o, "%-13s %8.3f ms %7d bytes" % ("range 2^16", (time.perf_counter() - t0) / 3 * 1e3, len(proof)), flush=True)
PY
