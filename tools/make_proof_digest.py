"""Writes tests/golden/proof_1024tx_d15_q96.json: SHA-256 digests of the proof the CPU restatement of the prover
(oracle/prover.py) produces for BASELINE.json's headline configuration -- the committed 1024-transaction witness
(tests/golden/witness_1024_d15.npz), 2^20 steps, blowup 8, 96 queries, Blake3_256, no field extension.  The GPU suite compares
the MI355X proof of the same witness against these digests (whole proof and per section), so the benchmarked configuration is
pinned to the oracle bit for bit, not only accepted by the verifier.

Self-consistency vector: produced by this repository's CPU oracle, not by the reference (which cannot be built here).
Run in the build container (about two minutes on 8 cores, ~12 GB):  python tools/make_proof_digest.py
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle import prover as OP  # noqa: E402

OPTIONS = (96, 8, 0, 0, 0, 4, 256)


def section_digests(proof, n_queries, width=94, n_comp=8):
    """SHA-256 of the proof's sections (layout: include/cstark.h), so that a mismatch names the stage it comes from."""
    import struct
    log_n = struct.unpack_from("<I", proof, 16)[0]
    log_N = log_n + 3
    off = 4 + 4 + 16 + 28
    out = {}

    def take(name, size):
        nonlocal off
        out[name] = hashlib.sha256(proof[off:off + size]).hexdigest()
        off += size
    take("trace_root", 32)
    take("constraint_root", 32)
    n_layers = struct.unpack_from("<I", proof, off)[0]
    off += 4
    take("fri_layer_roots", 32 * n_layers)
    take("remainder_commitment", 32)
    take("ood_trace", 2 * width * 8)
    take("ood_composition", n_comp * 8)
    take("pow_nonce", 8)
    take("trace_rows", n_queries * width * 8)
    take("trace_paths", n_queries * log_N * 32)
    take("composition_rows", n_queries * n_comp * 8)
    take("composition_paths", n_queries * log_N * 32)
    out["fri_openings_and_remainder"] = hashlib.sha256(proof[off:]).hexdigest()
    return out


if __name__ == "__main__":
    w = O.TxWitness.load(os.path.join(ROOT, "tests", "golden", "witness_1024_d15.npz"))
    t0 = time.perf_counter()
    proof = OP.prove(w, OPTIONS)
    dt = time.perf_counter() - t0
    doc = {"witness": "tests/golden/witness_1024_d15.npz", "options": list(OPTIONS), "proof_bytes": len(proof),
           "sha256": hashlib.sha256(proof).hexdigest(), "sections": section_digests(proof, OPTIONS[0]),
           "generated_by": "tools/make_proof_digest.py (oracle/prover.py, CPU restatement; %d threads, %.1f s)" % (O.num_threads(), dt)}
    path = os.path.join(ROOT, "tests", "golden", "proof_1024tx_d15_q96.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    print(json.dumps(doc, indent=1))
