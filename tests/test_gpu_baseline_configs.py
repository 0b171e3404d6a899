"""GPU parity for the BASELINE.json configurations that have no full-size counterpart in the reference's own harness:

* "range-proof AIR, 2^16 steps": the reference's range trace is fixed at 64 rows (src/range/mod.rs:34), so the 2^16-step case is the
  SYNTHETIC long accumulator of cstark_range_build_trace_bits / cstark_range_prove_bits (same RangeProofAir, src/range/air.rs:60-105,
  over an (n-1)-bit value) -- trace, both transition constraints and the merged evaluations bit-exact against the oracle, one proof
  accepted by the restated verifier -- plus 1024 independent 64-row proofs of the reference's own shape.
* "Merkle AIR, depth 32": depth + 1 must be a power of two (src/lib.rs:102-105), the nearest legal depth is 31 (8 * 31 + 7 = 255 rows
  of the 512-row cycle, src/merkle/constants.rs:27-29): 512 transfers = 2^18 rows from the sparse witness generator -- trace and all
  106 transition constraints bit-exact against the oracle, one proof accepted by the verifier, and the composite TransactionAir at
  the same depth.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 2**62 + 2**56 + 2**55 + 1
OPTS = (42, 8, 0, 0, 0, 4, 256)


@pytest.fixture(scope="module")
def backend():
    from certificate_stark_amd.backend import Backend
    b = Backend()
    yield b
    b.close()


def options(opts=OPTS):
    from certificate_stark_amd.prover import ProofOptions
    return ProofOptions(*opts)


def _words(log_n, seed):
    rng = np.random.default_rng(seed)
    w = rng.integers(0, 2**64, size=(1 << log_n) // 64, dtype=np.uint64)
    w[-1] &= np.uint64(2**63 - 1)
    return w


@pytest.mark.parametrize("log_n", [6, 7, 12, 16])
def test_long_range_trace_and_constraints(oracle, backend, log_n):
    from certificate_stark_amd.backend import to_numpy_u64
    words = _words(log_n, 1000 + log_n)
    ref, number = oracle.range_build_trace_bits(words, log_n)
    d_trace, got_number = backend.range_build_trace_bits(words, log_n)
    assert (to_numpy_u64(d_trace) == ref).all() and got_number == number
    value = sum(int(w) << (64 * i) for i, w in enumerate(words))
    assert int(oracle.from_mont([number])[0]) == value % P == int(oracle.from_mont(ref[1, -1:])[0])
    if log_n == 6:  # the reference's shape: identical to RangeProver::build_trace
        assert (ref == oracle.range_build_trace(int(words[0]))).all()
    # both transition constraints on one coset of the extension, and the merged evaluations on the constraint-evaluation domain
    lde = backend.lde_columns(backend.interpolate_columns(d_trace), 3)
    ref_lde = oracle.lde_columns(oracle.interpolate_columns(ref), 3)
    assert (to_numpy_u64(lde) == ref_lde).all()
    ev = backend.air_evaluate_transitions(backend.AIR_RANGE, lde, 0, 3)
    ref_ev = oracle.air_evaluate_transitions(oracle.AIR_RANGE, ref_lde, None, 2)
    assert (to_numpy_u64(ev) == ref_ev).all()
    desc = oracle.range_desc(int(oracle.from_mont([number])[0]))
    ta, tb, ba, bb = (oracle.random_elements(2, s) for s in (1, 2, 3, 4))
    ref_comb = oracle.air_combine(desc, ref_lde, ref_ev, ta, tb, ba, bb, 3)
    comb = backend.air_combine(backend.AIR_RANGE, lde, ev, ta, tb, ba, bb, [0, number], 3)
    assert (to_numpy_u64(comb) == ref_comb).all()


@pytest.mark.parametrize("log_n,opts", [(16, OPTS), (16, (96, 8, 0, 0, 2, 4, 256)), (10, (28, 8, 4, 1, 0, 4, 128))])
def test_long_range_proof_verifies(oracle, backend, log_n, opts):
    """BASELINE 'range-proof AIR, 2^16 steps, blowup 8' (synthetic): prove -> restated verifier; wrong value / flipped byte -> error."""
    from oracle import verifier as V
    words = _words(log_n, 7 + log_n)
    _, number = oracle.range_build_trace_bits(words, log_n)
    proof = backend.range_prove_bits(options(opts), words, log_n)
    assert V.parse(proof)["log_n"] == log_n
    assert V.verify_range(proof, number, options=list(opts))
    with pytest.raises(V.VerifierError):
        V.verify_range(proof, int(oracle.fp_add(np.array([number], np.uint64), oracle.to_mont([1]))[0]))
    bad = bytearray(proof)
    bad[len(bad) // 3] ^= 16
    with pytest.raises(V.VerifierError):
        V.verify_range(bytes(bad), number)
    assert proof == backend.range_prove_bits(options(opts), words, log_n)


def test_long_range_with_64_rows_is_the_reference_range_proof(oracle, backend):
    from certificate_stark_amd.prover import RangeProofExample
    value = 0x1234_5678_9ABC_DEF0 >> 1
    number = int(oracle.to_mont([value])[0])
    assert backend.range_prove_bits(options(), np.array([value], np.uint64), 6) == RangeProofExample(options(), number, backend).prove()


def test_1024_independent_range_proofs(oracle, backend):
    """1024 proofs of the reference's own shape (64 rows each = 2^16 rows in total), every one verified."""
    from oracle import verifier as V
    from certificate_stark_amd.prover import RangeProofExample
    rng = np.random.default_rng(64)
    values = rng.integers(0, 2**63, size=1024, dtype=np.uint64)
    numbers = oracle.to_mont(values % np.uint64(P))
    for i in range(1024):
        proof = RangeProofExample(options(), int(numbers[i]), backend).prove()
        if i % 32 == 0:  # the verifier is Python big-integer code: a sample is verified, every proof is parsed
            assert V.verify_range(proof, int(numbers[i]), options=list(OPTS))
        else:
            assert V.parse(proof)["log_n"] == 6


def test_merkle_air_depth_31_2_18(oracle, backend):
    """BASELINE 'Merkle AIR, depth 32, 2^18 steps' at the nearest legal depth 31: 512 transfers."""
    from oracle import verifier as V
    from certificate_stark_amd.backend import to_numpy_u64
    from certificate_stark_amd.prover import MerkleExample, TransactionMetadata
    w = oracle.TxWitness.generate(512, 31, seed=31)
    meta = TransactionMetadata.build_random(512, 31, seed=31)  # the product's own sparse generator: byte-identical witness
    for f in TransactionMetadata.FIELDS:
        assert (getattr(meta, f) == getattr(w, f)).all(), f
    assert int(w.s_indices.max()) >= 2**24  # genuinely beyond what a dense tree could hold
    ref = oracle.merkle_build_trace(w)
    assert ref.shape == (65, 1 << 18)
    backend.upload_witness(w)
    d_trace = backend.merkle_build_trace()
    assert (to_numpy_u64(d_trace) == ref).all()
    assert (ref[58:65, 0] == w.initial_roots[0]).all() and (ref[58:65, -1] == w.final_root).all()
    # all 106 transition constraints on two cosets of the extension
    ptab = oracle.periodic_table(oracle.merkle_periodic_columns(31), 18, 3)
    coeffs = backend.interpolate_columns(d_trace)
    ref_coeffs = oracle.interpolate_columns(ref)
    for k0 in (0, 5):
        lde1 = backend.lde_columns(coeffs, 3, k0=k0, nk=1)
        ref_lde1 = oracle.lde_columns(ref_coeffs, 3, k0=k0, nk=1)
        assert (to_numpy_u64(lde1) == ref_lde1).all()
        got_ev = to_numpy_u64(backend.air_evaluate_transitions(backend.AIR_MERKLE, lde1, 31, 3, k0=k0))
        assert (got_ev == oracle.air_evaluate_transitions(oracle.AIR_MERKLE, ref_lde1, ptab, 106, k0=k0)).all()
    ex = MerkleExample(options(), meta, backend)
    proof = ex.prove()
    d = V.parse(proof)
    assert d["log_n"] == 18 and d["depth"] == 31
    assert V.verify_merkle(proof, *ex.pub_inputs(), options=list(OPTS))
    with pytest.raises(V.VerifierError):
        V.verify_merkle(proof, ex.pub_inputs()[0], np.full(7, ex.pub_inputs()[1][0], np.uint64))


def test_transaction_air_depth_31(oracle, backend):
    """The composite TransactionAir at Merkle depth 31: trace, fused evaluations and proof bytes against the CPU restatement."""
    from oracle import prover as OP
    from oracle import verifier as V
    from certificate_stark_amd.backend import to_numpy_u64
    w = oracle.TxWitness.generate(4, 31, seed=131)
    ref = oracle.tx_build_trace(w)
    assert oracle.tx_check_trace(ref, 4, 31) == -1
    backend.upload_witness(w)
    assert (to_numpy_u64(backend.build_trace()) == ref).all()
    proof = backend.prove(options())
    assert proof == OP.prove(w, OPTS)
    assert V.verify(proof, w.initial_roots[0], w.final_root, options=list(OPTS))
