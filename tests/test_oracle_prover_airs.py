"""CPU: the oracle's restatement of the standalone AIRs' prove() (oracle/prover.py::prove_air -- src/merkle/update/mod.rs:81-106,
src/schnorr/mod.rs:143-172, src/range/mod.rs:75-100) writes proofs its own verifier accepts, rejects under the reference's negative
cases, and the full-size digests the GPU suite compares against are present and well formed."""
import json
import os

import numpy as np
import pytest

OPTS = (42, 8, 0, 0, 0, 4, 256)


def test_merkle_prove_air_round_trip(oracle):
    from oracle import prover as OP
    from oracle import verifier as V
    w = oracle.TxWitness.generate(2, 3, seed=5)
    trace = oracle.merkle_build_trace(w)
    r0, r1 = trace[58:65, 0], trace[58:65, -1]
    for opts in (OPTS, (28, 8, 4, 1, 0, 4, 128), (42, 8, 0, 0, 1, 4, 256), (42, 8, 0, 0, 2, 4, 256)):
        proof = OP.prove_air(oracle.AIR_MERKLE, w, opts)
        d = V.parse(proof)
        assert d["air"] == 1 and d["depth"] == 3 and d["log_n"] == 10
        assert V.verify_merkle(proof, r0, r1, options=list(opts))
        with pytest.raises(V.VerifierError):   # verify_with_wrong_inputs (src/merkle/update/mod.rs:129-138)
            V.verify_merkle(proof, r0, np.full(7, r1[0], np.uint64))
    assert proof == OP.prove_air(oracle.AIR_MERKLE, w, opts)


def test_range_prove_air_round_trip(oracle):
    from oracle import prover as OP
    from oracle import verifier as V
    number = int(oracle.to_mont([42])[0])
    for opts in (OPTS, (42, 8, 0, 1, 2, 4, 256)):
        proof = OP.prove_air(oracle.AIR_RANGE, number, opts)
        assert V.verify_range(proof, number, options=list(opts))
        with pytest.raises(V.VerifierError):   # src/range/tests.rs: another number
            V.verify_range(proof, int(oracle.to_mont([43])[0]))
    # the synthetic long accumulator; with 64 rows it IS the reference-shaped proof
    words = np.array([0x1234_5678_9ABC_DEF0 >> 1], np.uint64)
    assert OP.prove_air(oracle.AIR_RANGE, words, OPTS, log_n=6) == OP.prove_air(oracle.AIR_RANGE, int(oracle.to_mont(words)[0]), OPTS)
    from tools.proof_configs import range_words
    words = range_words(10, 3)
    proof = OP.prove_air(oracle.AIR_RANGE, words, OPTS, log_n=10)
    _, v = oracle.range_build_trace_bits(words, 10)
    assert V.parse(proof)["log_n"] == 10 and V.verify_range(proof, v, options=list(OPTS))


def test_schnorr_prove_air_round_trip(oracle):
    from oracle import prover as OP
    from oracle import verifier as V
    w = oracle.SchnorrWitness.generate(2, seed=7)
    for opts in (OPTS, (42, 8, 0, 0, 1, 4, 256)):
        proof = OP.prove_air(oracle.AIR_SCHNORR, w, opts)
        d = V.parse(proof)
        assert d["air"] == 2 and d["depth"] == 2
        assert V.verify_schnorr(proof, w, options=list(opts))
        w2 = oracle.SchnorrWitness(2)
        w2.messages[...], w2.sig_rx[...], w2.sig_s[...] = w.messages, w.sig_rx, w.sig_s
        w2.messages[0, 20] ^= np.uint64(1)
        with pytest.raises(V.VerifierError):   # wrong message (src/schnorr/mod.rs verify_with_wrong_inputs)
            V.verify_schnorr(proof, w2)


def test_full_size_digests_are_committed(oracle):
    """One digest file per BASELINE-size configuration (tools/proof_configs.py), written by tools/make_proof_digest.py."""
    from tools.proof_configs import configs, golden_path
    for name, cfg in configs(oracle).items():
        doc = json.load(open(golden_path(name)))
        assert doc["config"] == name and doc["options"] == list(cfg["options"])
        assert len(doc["sha256"]) == 64 and doc["proof_bytes"] > 0 and len(doc["sections"]) == 12
    # the two small ones are cheap enough to regenerate here
    from oracle import prover as OP
    import hashlib
    for name in ("range_64", "range_2_16"):
        cfg = configs(oracle)[name]
        proof = OP.prove_air(cfg["air"], cfg["witness"](), cfg["options"], log_n=cfg["log_n"])
        assert hashlib.sha256(proof).hexdigest() == json.load(open(golden_path(name)))["sha256"]
