"""Generates the committed fixtures under tests/golden/ with the CPU oracle (self-consistency vectors: the
reference has no golden vectors of its own).  Run in the build container:
    python tools/make_fixtures.py
  witness_1024_d15.npz   synthetic witness for the BASELINE workload (1024 transfers, depth-15 account tree),
                         SplitMix64 seed 0x5EED; input of bench.py
  tx2_d3_golden.npz      2 transfers at depth 3: witness + SHA-256 of the oracle trace, the trace-commitment
                         root and samples of the combined constraint evaluations
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)

w = O.TxWitness.generate(1024, 15, seed=0x5EED)
w.save(os.path.join(G, "witness_1024_d15.npz"))
print("witness_1024_d15: final_root[0] =", hex(int(w.final_root[0])))

w2 = O.TxWitness.generate(2, 3, seed=0x5EED)
trace = O.tx_build_trace(w2)
co = O.interpolate_columns(trace)
lde = O.lde_columns(co, 3)
leaves = O.hash_rows(lde, 3)
nodes = O.merkle_build(leaves)
cf = O.make_coeffs(17)
pub = np.concatenate([w2.initial_roots[0][:2], w2.final_root[:2]])
comb = O.tx_evaluate_constraints(lde, cf, pub, 3, 3)
np.savez_compressed(
    os.path.join(G, "tx2_d3_golden.npz"),
    **{f: getattr(w2, f) for f in w2.FIELDS},
    trace_sha256=np.frombuffer(hashlib.sha256(trace.tobytes()).digest(), np.uint8),
    lde_sha256=np.frombuffer(hashlib.sha256(lde.tobytes()).digest(), np.uint8),
    trace_root=nodes[1],
    coeff_seed=np.array([17]),
    combined_sha256=np.frombuffer(hashlib.sha256(comb.tobytes()).digest(), np.uint8),
    combined_samples=comb[:, ::257].copy(),
)
print("tx2_d3: trace root =", nodes[1].tobytes().hex())
