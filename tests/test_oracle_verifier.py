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


@pytest.mark.parametrize("ext,hash_fn", [(1, 0), (2, 0), (2, 1)])
def test_extension_field_proofs_on_the_cpu(oracle, ext, hash_fn):
    """FieldExtension::Quadratic / Cubic end to end in the restatement (src/tests.rs:18-30): the verifier's out-of-domain check
    evaluates the 115 constraints over the extension by sampling the base-field evaluator along a curve and interpolating, so an
    accepted proof ties the m-component constraint merge, the composition polynomial and every extension-field stage together."""
    from oracle import prover as OP
    from oracle import verifier as V
    w = oracle.TxWitness.generate(2, 3, seed=5 + ext)
    opts = (42, 8, 0, hash_fn, ext, 4, 256)
    proof = OP.prove(w, opts)
    assert V.verify(proof, w.initial_roots[0], w.final_root, options=list(opts))
    with pytest.raises(V.VerifierError):
        V.verify(proof, w.initial_roots[0], np.full(7, w.final_root[0], np.uint64))
    rng = np.random.default_rng(ext)
    for off in rng.integers(52, len(proof), 8):
        bad = bytearray(proof)
        bad[int(off)] ^= 0x20
        with pytest.raises(V.VerifierError):
            V.verify(bytes(bad), w.initial_roots[0], w.final_root)


def test_extension_arithmetic(oracle):
    """E2 = F_p[u]/(u^2 - 2u - 2) and E3 = F_p[v]/(v^3 + v + 1): inverses, associativity, the defining relations, and agreement of
    the C code (oracle/ext.c) with the Python tuples through a polynomial evaluation."""
    from oracle import verifier as V
    rng = np.random.default_rng(8)
    for m in (2, 3):
        g = V.e_gen(m)
        if m == 2:
            assert V.e_mul(g, g) == V.e_add(V.e_scale(g, 2), V.e_base(2, m))
        else:
            assert V.e_mul(V.e_mul(g, g), g) == V.e_sub(V.e_base(0, m), V.e_add(g, V.e_base(1, m)))
        for _ in range(20):
            x, y, z = [tuple(int(v) for v in rng.integers(0, V.P, m)) for _ in range(3)]
            assert V.e_mul(x, V.e_inv(x)) == V.e_base(1, m)
            assert V.e_mul(V.e_mul(x, y), z) == V.e_mul(x, V.e_mul(y, z))
            assert V.e_mul(x, V.e_add(y, z)) == V.e_add(V.e_mul(x, y), V.e_mul(x, z))
        co = rng.integers(0, V.P, size=(3, 64), dtype=np.uint64)
        zpt = tuple(int(v) for v in rng.integers(0, V.P, m))
        got = oracle.evaluate_polys_at_ext(oracle.to_mont(co), V.e_mont(zpt))
        for c in range(3):
            acc = V.e_base(0, m)
            for k in reversed(range(64)):
                acc = V.e_add(V.e_mul(acc, zpt), V.e_base(int(co[c, k]), m))
            assert tuple(V.from_mont(v) for v in got[c]) == acc
