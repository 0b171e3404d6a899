"""Writes the full-size proof digests under tests/golden/: SHA-256 (whole proof and per section) of the proofs the CPU restatement
of the prover (oracle/prover.py) produces for BASELINE.json's configurations, so that every configuration the GPU is timed on is
pinned to the oracle bit for bit, not only accepted by the verifier:

  proof_1024tx_d15_q96.json   the headline: the committed 1024-transaction witness (tests/golden/witness_1024_d15.npz), 2^20 steps,
                              blowup 8, 96 queries, Blake3_256, no field extension
  proof_<name>.json           the configurations of tools/proof_configs.py: range 2^16 / 64 rows, merkle 2^18 at depth 15 / 31,
                              schnorr 2^18, and the headline under the quadratic / cubic extension and Sha3_256

Self-consistency vectors: produced by this repository's CPU oracle, not by the reference (which cannot be built here).
Run in the build container:  python tools/make_proof_digest.py [name ...]     (no names: the headline only; `all`: everything;
about two minutes and ~12 GB per 2^20 proof on 8 cores, three to six for the extension proofs)
"""
import hashlib
import json
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

OPTIONS = (96, 8, 0, 0, 0, 4, 256)


def section_digests(proof, n_queries, width=94, n_comp=8):
    """SHA-256 of the proof's sections (layout: include/cstark.h), so that a mismatch names the stage it comes from.  The words per
    drawn-field element (1, 2 or 3) are read from the header's field_extension."""
    log_n = struct.unpack_from("<I", proof, 16)[0]
    m = struct.unpack_from("<I", proof, 24 + 16)[0] + 1
    log_N = log_n + struct.unpack_from("<I", proof, 24 + 4)[0].bit_length() - 1  # the LDE domain: blowup_factor of the header
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
    take("ood_trace", 2 * width * 8 * m)
    take("ood_composition", n_comp * 8 * m)
    take("pow_nonce", 8)
    take("trace_rows", n_queries * width * 8)
    take("trace_paths", n_queries * log_N * 32)
    take("composition_rows", n_queries * n_comp * 8 * m)
    take("composition_paths", n_queries * log_N * 32)
    out["fri_openings_and_remainder"] = hashlib.sha256(proof[off:]).hexdigest()
    return out


def write(path, doc):
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    print(json.dumps(doc, indent=1), flush=True)


def headline():
    from oracle import oracle as O
    from oracle import prover as OP
    w = O.TxWitness.load(os.path.join(ROOT, "tests", "golden", "witness_1024_d15.npz"))
    t0 = time.perf_counter()
    proof = OP.prove(w, OPTIONS)
    dt = time.perf_counter() - t0
    write(os.path.join(ROOT, "tests", "golden", "proof_1024tx_d15_q96.json"),
          {"witness": "tests/golden/witness_1024_d15.npz", "options": list(OPTIONS), "proof_bytes": len(proof),
           "sha256": hashlib.sha256(proof).hexdigest(), "sections": section_digests(proof, OPTIONS[0]),
           "generated_by": "tools/make_proof_digest.py (oracle/prover.py, CPU restatement; %d threads, %.1f s)" % (O.num_threads(), dt)})


def config(name):
    from oracle import oracle as O
    from oracle import prover as OP
    from tools.proof_configs import configs, golden_path
    cfg = configs(O)[name]
    t0 = time.perf_counter()
    proof = OP.prove_air(cfg["air"], cfg["witness"](), cfg["options"], log_n=cfg.get("log_n", 6))
    dt = time.perf_counter() - t0
    write(golden_path(name),
          {"config": name, "options": list(cfg["options"]), "proof_bytes": len(proof), "sha256": hashlib.sha256(proof).hexdigest(),
           "sections": section_digests(proof, cfg["options"][0], cfg["width"], cfg["n_comp"]),
           "generated_by": "tools/make_proof_digest.py %s (oracle/prover.py prove_air, CPU restatement; %d threads, %.1f s)" % (name, O.num_threads(), dt)})


if __name__ == "__main__":
    names = sys.argv[1:]
    if not names:
        headline()
    for nm in names:
        if nm == "all":
            from oracle import oracle as O
            from tools.proof_configs import configs
            headline()
            for k in configs(O):
                config(k)
        elif nm == "headline":
            headline()
        else:
            config(nm)
