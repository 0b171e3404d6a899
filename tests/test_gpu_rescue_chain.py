"""GPU: RescueAir, the hash-chain AIR of the reference's rescue bench (benches/rescue.rs:128-360; BASELINE.json config 0: 2^12 trace
steps at blowup 4) through cstark_rescue_prove: trace, transition values and whole proofs against the CPU restatement; the restated
verifier accepts them and rejects wrong public inputs (_verify_with_wrong_inputs, :96-102)."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
BENCH_OPTS = (42, 4, 0, 0, 0, 4, 256)  # benches/rescue.rs:370-378


@pytest.fixture(scope="module")
def backend():
    from certificate_stark_amd.backend import Backend
    b = Backend()
    yield b
    b.close()


def seed(oracle, first=42):
    return oracle.to_mont(np.arange(first, first + 7, dtype=np.uint64))  # RescueExample::new :38-46


@pytest.mark.parametrize("chain", [8, 64, 512])
def test_trace_and_transitions(oracle, backend, chain):
    from certificate_stark_amd.backend import Backend, to_numpy_u64
    sd = seed(oracle)
    ref = oracle.rescue_chain_build_trace(sd, chain)
    trace = backend.rescue_chain_build_trace(sd, chain)
    assert (to_numpy_u64(trace) == ref).all()
    log_n = ref.shape[1].bit_length() - 1
    lde_ref = oracle.lde_columns(oracle.interpolate_columns(ref.copy()), 2)
    lde = backend.lde_columns(backend.interpolate_columns(trace), 2)
    ptab = oracle.periodic_table(oracle.rescue_chain_periodic_columns(), log_n, 2)
    got = backend.air_evaluate_transitions(Backend.AIR_RESCUE_CHAIN, lde, 0, 2)
    assert (to_numpy_u64(got) == oracle.air_evaluate_transitions(oracle.AIR_RESCUE_CHAIN, lde_ref, ptab, 14)).all()


@pytest.mark.parametrize("chain,opts", [(8, BENCH_OPTS), (128, BENCH_OPTS), (512, BENCH_OPTS), (16, (42, 8, 0, 0, 0, 4, 256)), (32, (28, 16, 5, 1, 0, 8, 128)),
                                        (16, (42, 4, 0, 0, 1, 4, 256)), (16, (42, 4, 0, 0, 2, 16, 128))])
def test_proof_bytes_and_verification(oracle, backend, chain, opts):
    from oracle import prover as OP
    from oracle import verifier as V
    from certificate_stark_amd.prover import ProofOptions, RescueExample
    ex = RescueExample(chain, ProofOptions(*opts), backend)
    assert (ex.seed == seed(oracle)).all()
    proof = ex.prove()
    assert proof == OP.prove_air(oracle.AIR_RESCUE_CHAIN, (ex.seed, chain), opts)
    trace = oracle.rescue_chain_build_trace(ex.seed, chain)
    result = trace[:7, -1].copy()
    assert V.verify_rescue(proof, ex.seed, result, options=list(opts))
    with pytest.raises(V.VerifierError):                     # _verify_with_wrong_inputs: result = [result[0]; 7]
        V.verify_rescue(proof, ex.seed, np.full(7, result[0], np.uint64))
    with pytest.raises(V.VerifierError):
        V.verify_rescue(proof, seed(oracle, 43), result)


def test_bench_sizes(oracle, backend):
    """the bench's chain lengths 128 .. 1024 (benches/rescue.rs:23): 2^10 .. 2^13 trace steps; every proof verifies"""
    from oracle import verifier as V
    from certificate_stark_amd.prover import ProofOptions, RescueExample
    for chain in (128, 256, 512, 1024):
        ex = RescueExample(chain, ProofOptions(*BENCH_OPTS), backend)
        proof = ex.prove()
        result = oracle.rescue_chain_build_trace(ex.seed, chain)[:7, -1].copy()
        assert V.parse(proof)["log_n"] == 3 + chain.bit_length() - 1
        assert V.verify_rescue(proof, ex.seed, result, options=list(BENCH_OPTS))


def test_bad_arguments_are_refused(oracle, backend):
    from certificate_stark_amd import CstarkError
    from certificate_stark_amd.prover import ProofOptions
    for chain in (0, 4, 12):
        with pytest.raises(CstarkError):
            backend.rescue_prove(ProofOptions(*BENCH_OPTS), seed(oracle), chain)
    with pytest.raises(CstarkError):                         # blowup 2 is below the AIR's constraint-evaluation blowup (degree 3 + a cycle)
        backend.rescue_prove(ProofOptions(42, 2, 0, 0, 0, 4, 256), seed(oracle), 16)
    with pytest.raises(CstarkError):
        backend.rescue_prove(ProofOptions(*BENCH_OPTS), np.full(7, 2**64 - 1, np.uint64), 16)
