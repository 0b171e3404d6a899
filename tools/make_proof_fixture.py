"""Writes tests/golden/proof_2tx_d3.npz: a proof produced on an MI355X by cstark_tx_prove for the seeded 2-transaction,
depth-3 witness (the shape of the reference's acceptance tests, src/tests.rs:11-16), together with its public inputs.
The CPU suite replays the restated verifier on it.  Run on a GPU box:  python tools/make_proof_fixture.py gpurun_out/"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle import verifier as V  # noqa: E402
from certificate_stark_amd.prover import ProofOptions, TransactionExample, TransactionMetadata  # noqa: E402

out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden")
w = O.TxWitness.generate(2, 3, seed=0x5EED)
meta = TransactionMetadata(*[getattr(w, f) for f in TransactionMetadata.FIELDS])
tx = TransactionExample(ProofOptions(42, 8, 0, 0, 0, 4, 256), meta)
proof = tx.prove()
assert V.verify(proof, *tx.pub_inputs())
np.savez_compressed(os.path.join(out_dir, "proof_2tx_d3.npz"), proof=np.frombuffer(proof, np.uint8),
                    initial_root=meta.initial_roots[0], final_root=meta.final_root, seed=np.uint64(0x5EED))
print("proof bytes:", len(proof))
