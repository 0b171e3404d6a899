"""BASELINE.json's configurations as (witness, options) pairs, shared by the digest generator (tools/make_proof_digest.py, CPU
oracle), the GPU parity tests (tests/test_gpu_pinned_proofs.py) and -- through the same seeds and the product's own witness
generators -- bench.py's `other_configs`.  TEST INFRASTRUCTURE: imports the oracle for the witness generators.

Every entry: name -> dict(air, options, witness() -> object the oracle prover takes, plus what section_digests needs)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
OPTS42 = (42, 8, 0, 0, 0, 4, 256)  # the reference's get_example options (src/lib.rs:78-86 and the sub-AIRs' equivalents)
OPTS42_B4 = (42, 4, 0, 0, 0, 4, 256)  # build_options of the reference's MerkleAir / RangeProofAir tests (src/merkle/update/tests.rs:41-52, src/range/tests.rs:87-98)


def range_words(log_n=16, seed=16):
    w = np.random.default_rng(seed).integers(0, 2**64, size=(1 << log_n) // 64, dtype=np.uint64)
    w[-1] &= np.uint64(2**63 - 1)
    return w


def _golden_witness(O, n_tx=1024):
    """the committed 1024-transaction witness, or its first n_tx transactions (final root = the root before transaction n_tx)"""
    w = O.TxWitness.load(os.path.join(GOLDEN, "witness_1024_d15.npz"))
    if n_tx == w.n_tx:
        return w
    s = O.TxWitness(n_tx, w.depth)
    for f in s.FIELDS:
        if f != "final_root":
            getattr(s, f)[...] = getattr(w, f)[:n_tx]
    s.final_root[...] = w.initial_roots[n_tx]
    return s


def configs(O):
    """O: the oracle module (oracle.oracle)"""
    return {
        # BASELINE config 1: range-proof AIR, 2^16 steps (synthetic long accumulator) and the reference's own 64-row shape
        "range_2_16": dict(air=O.AIR_RANGE, options=OPTS42, witness=lambda: range_words(16, 16), log_n=16, width=2, n_comp=2),
        "range_64": dict(air=O.AIR_RANGE, options=OPTS42, witness=lambda: int(O.to_mont([12345 << 3])[0]), log_n=6, width=2, n_comp=2),
        # BASELINE config 2: Merkle AIR, 2^18 steps: depth 15 (the reference's constant) and 31 (nearest legal to "depth 32")
        "merkle_2_18_d15": dict(air=O.AIR_MERKLE, options=OPTS42, witness=lambda: _golden_witness(O, 512), width=65, n_comp=4),
        "merkle_2_18_d31": dict(air=O.AIR_MERKLE, options=OPTS42, witness=lambda: O.TxWitness.generate(512, 31, seed=31), width=65, n_comp=4),
        # the same two AIRs at the blowup factor the reference's own tests prove them at (4), and the long accumulator there
        "merkle_2_18_d15_b4": dict(air=O.AIR_MERKLE, options=OPTS42_B4, witness=lambda: _golden_witness(O, 512), width=65, n_comp=4),
        "range_2_16_b4": dict(air=O.AIR_RANGE, options=OPTS42_B4, witness=lambda: range_words(16, 16), log_n=16, width=2, n_comp=2),
        # BASELINE config 3: Schnorr AIR, 2^18 steps = 512 signatures
        "schnorr_2_18": dict(air=O.AIR_SCHNORR, options=OPTS42, witness=lambda: O.SchnorrWitness.generate(512, seed=1), width=56, n_comp=8),
        # BASELINE config 4 (headline) under the other option sets the reference's tests and CLI use (src/tests.rs:40-54,
        # examples/state-transition.rs:62-71): quadratic / cubic extension, Sha3_256
        "tx_2_20_quadratic": dict(air=O.AIR_STATE_TRANSITION, options=(96, 8, 0, 0, 1, 4, 256), witness=lambda: _golden_witness(O), width=94, n_comp=8),
        "tx_2_20_cubic": dict(air=O.AIR_STATE_TRANSITION, options=(96, 8, 0, 0, 2, 4, 256), witness=lambda: _golden_witness(O), width=94, n_comp=8),
        # the command line's -b / -f (examples/state-transition.rs:33-34, :46-47): blowup 16 with FRI folding 8, 64 transfers = 2^16 steps
        "tx_2_16_b16_f8": dict(air=O.AIR_STATE_TRANSITION, options=(96, 16, 0, 0, 0, 8, 256), witness=lambda: _golden_witness(O, 64), width=94, n_comp=8),
        "tx_2_20_sha3": dict(air=O.AIR_STATE_TRANSITION, options=(96, 8, 0, 1, 0, 4, 256), witness=lambda: _golden_witness(O), width=94, n_comp=8),
    }


def golden_path(name):
    return os.path.join(GOLDEN, "proof_%s.json" % name)
