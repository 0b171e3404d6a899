"""The restated verifier on a committed proof (tests/golden/proof_2tx_d3.npz, written on an MI355X by
tools/make_proof_fixture.py): acceptance, the reference's wrong-public-input case (src/lib.rs:152-161) and tamper
rejection, all on the CPU."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "proof_2tx_d3.npz")


@pytest.fixture(scope="module")
def fixture():
    z = np.load(GOLDEN)
    return bytes(z["proof"].tobytes()), z["initial_root"], z["final_root"]


def test_golden_proof_verifies(fixture):
    from oracle import verifier as V
    proof, r0, r1 = fixture
    assert V.verify(proof, r0, r1, options=[42, 8, 0, 0, 0, 4, 256])


def test_public_inputs_match_the_seeded_witness(fixture, oracle):
    proof, r0, r1 = fixture
    w = oracle.TxWitness.generate(2, 3, seed=0x5EED)
    assert np.array_equal(w.initial_roots[0], r0) and np.array_equal(w.final_root, r1)


def test_wrong_public_inputs_are_rejected(fixture):
    from oracle import verifier as V
    proof, r0, r1 = fixture
    with pytest.raises(V.VerifierError):
        V.verify(proof, r0, np.full(7, r1[0], np.uint64))
    with pytest.raises(V.VerifierError):
        V.verify(proof, r0, r1, options=[96, 8, 0, 0, 0, 4, 256])


def test_tampered_proofs_are_rejected(fixture):
    from oracle import verifier as V
    proof, r0, r1 = fixture
    rng = np.random.default_rng(3)
    rejected = 0
    for off in rng.integers(52, len(proof), 24):
        bad = bytearray(proof)
        bad[int(off)] ^= 0x10
        try:
            V.verify(bytes(bad), r0, r1)
        except V.VerifierError:
            rejected += 1
    assert rejected == 24


def test_channel_matches_trace_commitment(fixture, oracle):
    """The trace root in the proof is the root the oracle computes for the same witness (trace -> LDE -> Blake3 tree)."""
    from oracle import verifier as V
    proof, _, _ = fixture
    d = V.parse(proof)
    w = oracle.TxWitness.generate(2, 3, seed=0x5EED)
    trace = oracle.tx_build_trace(w)
    lde = oracle.lde_columns(oracle.interpolate_columns(trace.copy()), 3)
    nodes = oracle.merkle_build(oracle.hash_rows(lde, 3))
    assert nodes[1].tobytes() == d["trace_root"]


def test_oracle_prover_reproduces_the_gpu_proof_bytes(fixture, oracle):
    """Whole-pipeline parity: the CPU restatement of prove() gives byte for byte the proof the MI355X wrote."""
    from oracle import prover as OP
    proof, _, _ = fixture
    w = oracle.TxWitness.generate(2, 3, seed=0x5EED)
    mine = OP.prove(w, (42, 8, 0, 0, 0, 4, 256))
    assert len(mine) == len(proof)
    assert mine == proof
