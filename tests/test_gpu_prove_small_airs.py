"""Complete proofs of the standalone AIRs through cstark_air_prove, accepted by the restated verifier.  Mirrors the reference's
acceptance tests: src/range/tests.rs (17, 42, max input, too-large input, wrong public input), src/schnorr/tests.rs,
src/merkle/update/tests.rs (prove -> verify; verify with wrong inputs -> error)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 2**62 + 2**56 + 2**55 + 1
OPTS = (42, 8, 0, 0, 0, 4, 256)  # build_options(1) of src/schnorr/tests.rs:40-54 and src/tests.rs (blowup 8); the MerkleAir / RangeProofAir tests use blowup 4: tests/test_gpu_options.py mirrors them at their own options


@pytest.fixture(scope="module")
def backend():
    from certificate_stark_amd.backend import Backend
    b = Backend()
    yield b
    b.close()


def options():
    from certificate_stark_amd.prover import ProofOptions
    return ProofOptions(*OPTS)


@pytest.mark.parametrize("value", [17, 42, 2**63 - 1, 0])
def test_range_proof_verification(oracle, backend, value):
    from oracle import verifier as V
    from certificate_stark_amd.prover import RangeProofExample
    number = int(oracle.to_mont([value % P])[0])      # BaseElement::from reduces
    proof = RangeProofExample(options(), number, backend).prove()
    assert V.verify_range(proof, number, options=list(OPTS))
    with pytest.raises(V.VerifierError):              # range_test_basic_proof_verification_fail: another number
        V.verify_range(proof, int(oracle.to_mont([(value + 1) % P])[0]))
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 4
    with pytest.raises(V.VerifierError):
        V.verify_range(bytes(bad), number)


def test_range_proof_input_too_large(backend):
    from certificate_stark_amd import CstarkError
    from certificate_stark_amd.prover import RangeProofExample
    with pytest.raises(CstarkError):                  # src/range/tests.rs:54-62 (should_panic): raw M is not an element
        RangeProofExample(options(), P, backend).prove()


@pytest.mark.parametrize("n_tx,depth", [(2, 3), (8, 15)])
def test_merkle_proof_verification(oracle, backend, n_tx, depth):
    from oracle import verifier as V
    from certificate_stark_amd.prover import MerkleExample, TransactionMetadata
    meta = TransactionMetadata.build_random(n_tx, depth, seed=31 + n_tx)
    ex = MerkleExample(options(), meta, backend)
    proof = ex.prove()
    assert V.verify_merkle(proof, *ex.pub_inputs(), options=list(OPTS))
    r0, r1 = ex.pub_inputs()
    with pytest.raises(V.VerifierError):              # verify_with_wrong_inputs (src/merkle/update/mod.rs:129-138)
        V.verify_merkle(proof, r0, np.full(7, r1[0], np.uint64))
    with pytest.raises(V.VerifierError):
        V.verify(proof, r0, r1)                       # not a TransactionAir proof


@pytest.mark.parametrize("n_sig", [1, 2, 8])
def test_schnorr_proof_verification(oracle, backend, n_sig):
    from oracle import verifier as V
    from certificate_stark_amd.prover import SchnorrExample
    ex = SchnorrExample.build_random(options(), n_sig, seed=500 + n_sig, backend=backend)
    proof = ex.prove()
    w = oracle.SchnorrWitness(n_sig)
    w.messages[...], w.sig_rx[...], w.sig_s[...] = ex.messages, ex.sig_rx, ex.sig_s
    assert V.verify_schnorr(proof, w, options=list(OPTS))
    w2 = oracle.SchnorrWitness(n_sig)                 # wrong message (src/schnorr/mod.rs verify_with_wrong_inputs)
    w2.messages[...], w2.sig_rx[...], w2.sig_s[...] = ex.messages, ex.sig_rx, ex.sig_s
    w2.messages[0, 20] ^= np.uint64(1)
    with pytest.raises(V.VerifierError):
        V.verify_schnorr(proof, w2)


def test_invalid_signature_gives_unverifiable_proof(oracle, backend):
    from oracle import verifier as V
    from certificate_stark_amd.prover import SchnorrExample
    ex = SchnorrExample.build_random(options(), 2, seed=9, backend=backend)
    ex.sig_s[1, 3] ^= 1
    proof = ex.prove()
    w = oracle.SchnorrWitness(2)
    w.messages[...], w.sig_rx[...], w.sig_s[...] = ex.messages, ex.sig_rx, ex.sig_s
    with pytest.raises(V.VerifierError):
        V.verify_schnorr(proof, w)


@pytest.mark.parametrize("ext", [1, 2])
def test_sub_air_proofs_over_extension_fields(oracle, backend, ext):
    """The quadratic / cubic variants of the sub-AIR acceptance tests (src/range/tests.rs:25-43, src/schnorr/tests.rs:19-31,
    src/merkle/update/tests.rs:19-31): one generic extension prover serves every AIR; the restated verifier evaluates each AIR over
    the extension by interpolation."""
    from oracle import verifier as V
    from certificate_stark_amd.prover import MerkleExample, ProofOptions, RangeProofExample, SchnorrExample, TransactionMetadata
    opts = (42, 8, 0, 0, ext, 4, 256)
    po = ProofOptions(*opts)
    number = int(oracle.to_mont([42])[0])
    proof = RangeProofExample(po, number, backend).prove()
    assert V.verify_range(proof, number, options=list(opts))
    with pytest.raises(V.VerifierError):
        V.verify_range(proof, int(oracle.to_mont([43])[0]))
    meta = TransactionMetadata.build_random(4, 7, seed=70 + ext)
    mex = MerkleExample(po, meta, backend)
    proof = mex.prove()
    assert V.verify_merkle(proof, *mex.pub_inputs(), options=list(opts))
    r0, r1 = mex.pub_inputs()
    with pytest.raises(V.VerifierError):
        V.verify_merkle(proof, r0, np.full(7, r1[0], np.uint64))
    sex = SchnorrExample.build_random(po, 2, seed=80 + ext, backend=backend)
    proof = sex.prove()
    w = oracle.SchnorrWitness(2)
    w.messages[...], w.sig_rx[...], w.sig_s[...] = sex.messages, sex.sig_rx, sex.sig_s
    assert V.verify_schnorr(proof, w, options=list(opts))
    w.messages[1, 15] ^= np.uint64(2)
    with pytest.raises(V.VerifierError):
        V.verify_schnorr(proof, w)


def _switch_digests():
    """sha256 of one proof per sub-AIR (and one with an extension field), on a fresh backend: what the child process of the test below
    prints and what the test itself computes with the default paths."""
    import hashlib
    from certificate_stark_amd.backend import Backend
    from certificate_stark_amd.prover import MerkleExample, ProofOptions, RangeProofExample, SchnorrExample, TransactionMetadata
    b = Backend()
    try:
        opt = ProofOptions(*OPTS)
        out = [hashlib.sha256(MerkleExample(opt, TransactionMetadata.build_random(8, 15, seed=39), b).prove()).hexdigest(),
               hashlib.sha256(SchnorrExample.build_random(opt, 8, seed=508, backend=b).prove()).hexdigest(),
               hashlib.sha256(SchnorrExample.build_random(ProofOptions(42, 8, 0, 0, 1, 4, 256), 8, seed=509, backend=b).prove()).hexdigest(),
               hashlib.sha256(RangeProofExample(opt, 12345 << 3, b).prove()).hexdigest()]
    finally:
        b.close()
    return out


def test_alternative_sub_air_paths_give_the_same_bytes():
    """The sub-AIR provers choose their kernels by switches read once per process: SchnorrAir's degree split and overlapped ladders,
    MerkleAir's folded round gadgets, the four-lane Merkle tails, the cached assertion divisors, the 8-lane host coin, the device-side coin of the FRI layers, a single
    range proof as a batch of one.  A child process with every switch on its other setting must produce the same proofs, byte for byte."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_gpu_prove_small_airs import _switch_digests\n"
            "print(' '.join(_switch_digests()))\n") % (root, os.path.join(root, "tests"))
    env = dict(os.environ, CSTARK_SCHNORR_SPLIT="0", CSTARK_SCHNORR_OVERLAP="0", CSTARK_MERKLE_ROUNDS="0", CSTARK_MERKLE_QUAD="0",  # (without the
               # split the folded round of SchnorrAir's hash is not used either; CSTARK_SCHNORR_ROUNDS=0 alone: second child below)
               CSTARK_AIR_INV_TABLES="0", CSTARK_COIN_SCALAR="1", CSTARK_RANGE_VIA_BATCH="1", CSTARK_FRI_DEVICE_COIN="0")
    got = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert got.returncode == 0, got.stderr[-2000:]
    want = _switch_digests()
    assert got.stdout.strip().splitlines()[-1].split() == want
    got = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CSTARK_SCHNORR_ROUNDS="0", CSTARK_SCHNORR_FINAL5="0"), capture_output=True, text=True, timeout=600)
    assert got.returncode == 0, got.stderr[-2000:]
    assert got.stdout.strip().splitlines()[-1].split() == want
