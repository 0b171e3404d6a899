"""GPU: every proof the MI355X is timed on is pinned to the CPU restatement of the prover byte for byte.

* small sizes: cstark_air_prove / cstark_range_prove_bits == oracle/prover.py::prove_air for MerkleAir, SchnorrAir, RangeProofAir
  (src/merkle/update/mod.rs:81-106, src/schnorr/mod.rs:143-172, src/range/mod.rs:75-100), base field, both extensions, both hashes;
* BASELINE sizes (tools/proof_configs.py): SHA-256 of the whole proof and of every section against tests/golden/proof_<name>.json,
  written in the build container by tools/make_proof_digest.py from the CPU prover: range 2^16 / 64 rows, merkle 2^18 at depth 15
  and 31, schnorr 2^18, and the 2^20 TransactionAir proof under the quadratic / cubic extension and Sha3_256;
* the fused sub-AIR evaluators at 2^18 rows (k_schnorr_fused / k_merkle_fused index far beyond the 2^12 rows of the stage tests):
  == the materialising evaluator + the generic merge on one coset, and the whole proofs accepted by the restated verifier.
"""
import hashlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
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


def gpu_prove(backend, oracle, cfg, witness):
    from certificate_stark_amd.backend import Backend
    air, opts = cfg["air"], options(cfg["options"])
    if air == oracle.AIR_STATE_TRANSITION:
        backend.upload_witness(witness)
        return backend.prove(opts)
    if air == oracle.AIR_MERKLE:
        backend.upload_witness(witness)
        return backend.air_prove(Backend.AIR_MERKLE, opts)
    if air == oracle.AIR_SCHNORR:
        backend.upload_schnorr_witness(witness.messages, witness.sig_rx, witness.sig_s)
        return backend.air_prove(Backend.AIR_SCHNORR, opts)
    if isinstance(witness, int):
        return backend.air_prove(Backend.AIR_RANGE, opts, witness)
    return backend.range_prove_bits(opts, witness, cfg["log_n"])


# ---- byte parity at sizes the CPU prover finishes in seconds ------------------------------------------------------------------------
@pytest.mark.parametrize("n_tx,depth,opts", [(2, 3, OPTS), (8, 15, OPTS), (4, 7, (28, 8, 4, 1, 0, 4, 128)), (2, 3, (42, 8, 0, 0, 1, 4, 256)),
                                             (4, 7, (42, 8, 0, 1, 2, 4, 512))])
def test_merkle_proof_bytes(oracle, backend, n_tx, depth, opts):
    from oracle import prover as OP
    w = oracle.TxWitness.generate(n_tx, depth, seed=40 + n_tx)
    got = gpu_prove(backend, oracle, dict(air=oracle.AIR_MERKLE, options=opts), w)
    assert got == OP.prove_air(oracle.AIR_MERKLE, w, opts)


@pytest.mark.parametrize("n_sig,opts", [(1, OPTS), (2, OPTS), (8, OPTS), (2, (28, 8, 4, 1, 0, 4, 128)), (2, (42, 8, 0, 0, 1, 4, 256)),
                                        (4, (42, 8, 0, 0, 2, 4, 256))])
def test_schnorr_proof_bytes(oracle, backend, n_sig, opts):
    from oracle import prover as OP
    w = oracle.SchnorrWitness.generate(n_sig, seed=600 + n_sig)
    got = gpu_prove(backend, oracle, dict(air=oracle.AIR_SCHNORR, options=opts), w)
    assert got == OP.prove_air(oracle.AIR_SCHNORR, w, opts)


@pytest.mark.parametrize("value,opts", [(17, OPTS), (2**63 - 1, OPTS), (0, OPTS), (42, (42, 8, 0, 1, 2, 4, 256)), (42, (30, 8, 3, 0, 1, 4, 128))])
def test_range_proof_bytes(oracle, backend, value, opts):
    from oracle import prover as OP
    number = int(oracle.to_mont([value % oracle.P])[0])
    got = gpu_prove(backend, oracle, dict(air=oracle.AIR_RANGE, options=opts), number)
    assert got == OP.prove_air(oracle.AIR_RANGE, number, opts)


@pytest.mark.parametrize("log_n,opts", [(7, OPTS), (12, OPTS), (10, (28, 8, 4, 1, 0, 4, 128)), (12, (42, 8, 0, 0, 2, 4, 256))])
def test_long_range_proof_bytes(oracle, backend, log_n, opts):
    from oracle import prover as OP
    from tools.proof_configs import range_words
    words = range_words(log_n, 100 + log_n)
    got = gpu_prove(backend, oracle, dict(air=oracle.AIR_RANGE, options=opts, log_n=log_n), words)
    assert got == OP.prove_air(oracle.AIR_RANGE, words, opts, log_n=log_n)


# ---- BASELINE sizes: digests of the CPU prover's proofs -----------------------------------------------------------------------------
def _check_digest(name, cfg, proof):
    from tools.make_proof_digest import section_digests
    from tools.proof_configs import golden_path
    gold = json.load(open(golden_path(name)))
    assert gold["options"] == list(cfg["options"])
    assert len(proof) == gold["proof_bytes"]
    got = section_digests(proof, cfg["options"][0], cfg["width"], cfg["n_comp"])
    for sec, digest in gold["sections"].items():
        assert got[sec] == digest, "%s: proof section differs from the CPU prover's: %s" % (name, sec)
    assert hashlib.sha256(proof).hexdigest() == gold["sha256"]


@pytest.mark.parametrize("cfg", ["merkle_2_18", "schnorr_2_18"])
def test_sub_air_proofs_are_the_same_without_the_matrix_cores(cfg):
    """The sub-AIRs' folded round gadgets run on the matrix cores (k_merkle_rounds_mfma, csrc/rounds_mfma.hip) -- the path the digest
    tests below pin; CSTARK_ROUNDS_MFMA=0 (read once per process) keeps the vector-ALU kernel k_merkle_rounds.  A child process per
    setting proves the configuration of tools/bench_air_one.py: the proofs must be byte-identical."""
    import subprocess
    import sys
    code = ("import sys, os, hashlib; sys.path.insert(0, %r); sys.argv = ['bench_air_one.py', %r, '1']\n"
            "import runpy, builtins\n"
            "g = runpy.run_path(os.path.join(%r, 'tools', 'bench_air_one.py'))\n"
            "print(hashlib.sha256(g['p']).hexdigest())\n") % (ROOT, cfg, ROOT)
    digests = []
    for v in ("1", "0"):
        got = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CSTARK_ROUNDS_MFMA=v), capture_output=True, text=True, timeout=600)
        assert got.returncode == 0, got.stderr[-2000:]
        digests.append(got.stdout.strip().splitlines()[-1])
    assert len(digests[0]) == 64 and digests[0] == digests[1]


@pytest.mark.parametrize("name", ["range_64", "range_2_16", "merkle_2_18_d15", "merkle_2_18_d31", "schnorr_2_18"])
def test_sub_air_proofs_at_baseline_size_equal_the_cpu_provers(oracle, backend, name):
    from oracle import verifier as V
    from tools.proof_configs import configs
    cfg = configs(oracle)[name]
    w = cfg["witness"]()
    proof = gpu_prove(backend, oracle, cfg, w)
    _check_digest(name, cfg, proof)
    # ... and the restated verifier accepts them (the reference's acceptance tests at BASELINE's sizes)
    opts = list(cfg["options"])
    if cfg["air"] == oracle.AIR_MERKLE:
        assert V.parse(proof)["log_n"] == 18
        assert V.verify_merkle(proof, w.initial_roots[0], w.final_root, options=opts)
        with pytest.raises(V.VerifierError):
            V.verify_merkle(proof, w.initial_roots[0], np.full(7, w.final_root[0], np.uint64))
    elif cfg["air"] == oracle.AIR_SCHNORR:
        assert V.parse(proof)["log_n"] == 18 and V.parse(proof)["depth"] == 512
        assert V.verify_schnorr(proof, w, options=opts)
        w2 = oracle.SchnorrWitness(512)
        w2.messages[...], w2.sig_rx[...], w2.sig_s[...] = w.messages, w.sig_rx, w.sig_s
        w2.messages[300, 20] ^= np.uint64(1)
        with pytest.raises(V.VerifierError):
            V.verify_schnorr(proof, w2)
    elif isinstance(w, int):
        assert V.verify_range(proof, w, options=opts)
    else:
        assert V.verify_range(proof, oracle.range_build_trace_bits(w, cfg["log_n"])[1], options=opts)


@pytest.mark.parametrize("name", ["tx_2_20_quadratic", "tx_2_20_cubic", "tx_2_20_sha3"])
def test_state_transition_2_20_option_sets_equal_the_cpu_provers(oracle, backend, name):
    """The headline witness under FieldExtension::Quadratic / Cubic and Sha3_256 (src/tests.rs:40-54, examples/state-transition.rs:62-71):
    the proofs `profiles/*bench_{quadratic,cubic,sha3}.json` time."""
    from tools.proof_configs import configs
    cfg = configs(oracle)[name]
    proof = gpu_prove(backend, oracle, cfg, cfg["witness"]())
    _check_digest(name, cfg, proof)


def test_product_witness_generators_give_the_pinned_witnesses(oracle):
    """bench.py's other_configs build their witnesses with the product's generators and the seeds of tools/proof_configs.py."""
    from certificate_stark_amd.prover import ProofOptions, SchnorrExample, TransactionMetadata
    from tools.proof_configs import configs
    cf = configs(oracle)
    w = cf["schnorr_2_18"]["witness"]()
    ex = SchnorrExample.build_random(ProofOptions(), 512, seed=1, backend=object())
    assert (ex.messages == w.messages).all() and (ex.sig_rx == w.sig_rx).all() and (ex.sig_s == w.sig_s).all()
    w = cf["merkle_2_18_d31"]["witness"]()
    meta = TransactionMetadata.build_random(512, 31, seed=31)
    for f in TransactionMetadata.FIELDS:
        assert (getattr(meta, f) == getattr(w, f)).all(), f


# ---- the fused sub-AIR evaluators at 2^18 rows ------------------------------------------------------------------------------------------
def test_schnorr_fused_evaluator_at_2_18(oracle, backend):
    """cstark_schnorr_evaluate_constraints (k_schnorr_fused) == cstark_schnorr_evaluate_transitions + cstark_air_combine at 512
    signatures, on two cosets (a whole 8-coset table of materialised values would be 0.94 GB; two cosets exercise the same indexing)."""
    import torch
    from certificate_stark_amd.backend import to_numpy_u64
    n_sig, log_n, log_b = 512, 18, 3
    w = oracle.SchnorrWitness.generate(n_sig, seed=1)
    backend.upload_schnorr_witness(w.messages, w.sig_rx, w.sig_s)
    co = backend.interpolate_columns(backend.schnorr_build_trace())
    aux_co = backend.interpolate_columns(backend.schnorr_aux_columns())
    av_co = backend.schnorr_assertion_polys(log_n)
    ta, tb = oracle.random_elements(56, 11), oracle.random_elements(56, 12)
    ba, bb = oracle.random_elements(61, 13), oracle.random_elements(61, 14)
    for k0 in (0, 5):
        lde = backend.lde_columns(co, log_b, k0=k0, nk=1)
        aux = backend.lde_columns(aux_co, log_b, k0=k0, nk=1)
        av = backend.lde_columns(av_co, log_b, k0=k0, nk=1)
        ev = backend.schnorr_evaluate_transitions(lde, aux, log_b, k0=k0)
        ref = backend.air_combine(backend.AIR_SCHNORR, lde, ev, ta, tb, ba, bb, None, log_b, k0=k0, n_items=n_sig, avals_lde=av)
        fused = backend.schnorr_evaluate_constraints(lde, aux, ta, tb, ba, bb, av, log_b, k0=k0, n_sig=n_sig)
        assert torch.equal(ref, fused)
        if k0 == 5:  # the materialising side itself against the oracle, on a window of one coset's values
            ptab = oracle.periodic_table(oracle.schnorr_mask_columns(), log_n, log_b)
            ref_ev = oracle.schnorr_evaluate_transitions(to_numpy_u64(lde), to_numpy_u64(aux), ptab, k0=k0)
            assert (to_numpy_u64(ev) == ref_ev).all()
            desc = oracle.schnorr_desc(w)
            ref_comb = oracle.air_combine(desc, to_numpy_u64(lde), ref_ev, ta, tb, ba, bb, log_b, k0=k0, avals=to_numpy_u64(av))
            assert (to_numpy_u64(fused) == ref_comb).all()


def test_merkle_fused_evaluator_at_2_18_depth_15(oracle, backend):
    """cstark_merkle_evaluate_constraints (k_merkle_fused) == the materialising evaluator + merge == the oracle, 512 transfers at the
    reference's depth 15, on one coset of the constraint-evaluation domain and one outside it."""
    import torch
    from certificate_stark_amd.backend import to_numpy_u64
    from tools.proof_configs import configs
    w = configs(oracle)["merkle_2_18_d15"]["witness"]()
    backend.upload_witness(w)
    d_trace = backend.merkle_build_trace()
    ref_trace = oracle.merkle_build_trace(w)
    assert (to_numpy_u64(d_trace) == ref_trace).all()
    desc = oracle.merkle_desc(ref_trace)
    co = backend.interpolate_columns(d_trace)
    ta, tb = oracle.random_elements(106, 21), oracle.random_elements(106, 22)
    ba, bb = oracle.random_elements(14, 23), oracle.random_elements(14, 24)
    ptab = oracle.periodic_table(oracle.merkle_periodic_columns(15), 18, 3)
    for k0 in (4, 3):
        lde = backend.lde_columns(co, 3, k0=k0, nk=1)
        ev = backend.air_evaluate_transitions(backend.AIR_MERKLE, lde, 15, 3, k0=k0)
        ref = backend.air_combine(backend.AIR_MERKLE, lde, ev, ta, tb, ba, bb, desc.a_value, 3, k0=k0)
        fused = backend.merkle_evaluate_constraints(lde, 15, ta, tb, ba, bb, desc.a_value, 3, k0=k0)
        assert torch.equal(ref, fused)
        ref_ev = oracle.air_evaluate_transitions(oracle.AIR_MERKLE, to_numpy_u64(lde), ptab, 106, k0=k0)
        assert (to_numpy_u64(ev) == ref_ev).all()
        assert (to_numpy_u64(fused) == oracle.air_combine(desc, to_numpy_u64(lde), ref_ev, ta, tb, ba, bb, 3, k0=k0)).all()
        assert bool(fused.any()) == (k0 % 2 == 0)   # odd cosets are outside the 4n-point evaluation domain


# ---- the batched range prover ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("opts", [OPTS, (30, 8, 3, 1, 0, 4, 128), (42, 8, 0, 0, 0, 4, 512), (96, 8, 0, 0, 0, 4, 1024),
                                  (20, 8, 13, 0, 0, 4, 256), (20, 8, 12, 1, 0, 4, 256)])   # the last two: proof of work by the device-side search, all proofs at once
def test_batched_range_proofs_equal_single_proofs(oracle, backend, opts):
    """cstark_range_prove_batch: B reference-shaped proofs in one call (benches/range.rs:15-37 proves them one by one); every proof
    must equal cstark_air_prove's bytes -- and, for a sample, the CPU prover's -- and verify."""
    from oracle import prover as OP
    from oracle import verifier as V
    from certificate_stark_amd.backend import Backend
    rng = np.random.default_rng(5)
    values = [0, 1, 17, 2**63 - 1, 2**62, oracle.P - 1] + [int(v) for v in rng.integers(0, 2**63, size=94, dtype=np.uint64)]
    numbers = oracle.to_mont(np.array([v % oracle.P for v in values], np.uint64))
    proofs = backend.range_prove_batch(options(opts), numbers)
    assert len(proofs) == len(values)
    for i, proof in enumerate(proofs):
        assert proof == backend.air_prove(Backend.AIR_RANGE, options(opts), int(numbers[i])), i
        if i < 24:   # cstark_range_prove_bits at 64 rows: the generic prover whatever CSTARK_RANGE_VIA_BATCH says
            assert proof == backend.range_prove_bits(options(opts), np.array([values[i] % oracle.P], np.uint64), 6), i
        if i < 8:
            assert proof == OP.prove_air(oracle.AIR_RANGE, int(numbers[i]), opts)
            assert V.verify_range(proof, int(numbers[i]), options=list(opts))
    assert backend.range_prove_batch(options(opts), numbers[:3]) == proofs[:3]   # a smaller batch on the same context


def test_batched_range_prover_at_baseline_size_and_its_errors(oracle, backend):
    """BASELINE 'range, 2^16 steps' in the reference's own shape: 1024 proofs of 64 rows in ONE call; every proof parsed, a sample
    compared with the single-proof path and verified.  Non-elements and values of 64 bits are refused as by the single prover."""
    from oracle import verifier as V
    from certificate_stark_amd import CstarkError
    from certificate_stark_amd.backend import Backend
    from certificate_stark_amd.prover import ProofOptions
    values = np.random.default_rng(64).integers(0, 2**63, size=1024, dtype=np.uint64)
    numbers = oracle.to_mont(values % np.uint64(oracle.P))
    proofs = backend.range_prove_batch(options(), numbers)
    assert len(proofs) == 1024
    for i in range(0, 1024, 97):
        assert proofs[i] == backend.air_prove(Backend.AIR_RANGE, options(), int(numbers[i]))
        assert proofs[i] == backend.range_prove_bits(options(), np.array([int(values[i]) % oracle.P], np.uint64), 6)   # generic path
        assert V.verify_range(proofs[i], int(numbers[i]), options=list(OPTS))
    for p in proofs:
        assert V.parse(p)["log_n"] == 6
    with pytest.raises(CstarkError):
        backend.range_prove_batch(options(), np.array([1, oracle.P], np.uint64))            # raw M is not an element (src/range/tests.rs:54-62)
    with pytest.raises(CstarkError):
        backend.range_prove_batch(ProofOptions(42, 8, 0, 0, 1, 4, 256), numbers[:2])        # extension fields: cstark_air_prove
    # a slot too small for a proof is refused (the required size is what cstark_tx_proof_size_bound(1, opt) returns)
    import ctypes as C
    from certificate_stark_amd import _lib
    o = backend._options_struct(options())
    nums = np.ascontiguousarray(numbers[:2])
    buf, lens = np.zeros(2 * 1000, np.uint8), np.zeros(2, np.uint64)
    rc = backend.lib.cstark_range_prove_batch(backend.ctx, C.byref(o), nums.ctypes.data_as(_lib.u64p), C.c_uint32(2), buf.ctypes.data_as(_lib.u8p),
                                              C.c_size_t(1000), lens.ctypes.data_as(C.POINTER(C.c_size_t)))
    assert rc == -1 and not buf.any()
    assert backend.lib.cstark_range_prove_batch(backend.ctx, C.byref(o), nums.ctypes.data_as(_lib.u64p), C.c_uint32(0), buf.ctypes.data_as(_lib.u8p),
                                                C.c_size_t(1000), lens.ctypes.data_as(C.POINTER(C.c_size_t))) == -1


@pytest.mark.parametrize("n_sig", [8, 64, 512])
def test_schnorr_degree_split_equals_direct_evaluation(oracle, backend, n_sig):
    """cstark_schnorr_evaluate_constraints_lde (the doubling / addition gadgets on the even cosets only, their eight merged polynomials
    extended to the odd cosets: what cstark_air_prove uses) == cstark_schnorr_evaluate_constraints (every point directly) at every
    point of the real extension; at 8 signatures also == the oracle's materialise-and-merge."""
    import torch
    from certificate_stark_amd.backend import to_numpy_u64
    log_b = 3
    w = oracle.SchnorrWitness.generate(n_sig, seed=900 + n_sig)
    backend.upload_schnorr_witness(w.messages, w.sig_rx, w.sig_s)
    log_n = 9 + n_sig.bit_length() - 1
    lde = backend.lde_columns(backend.interpolate_columns(backend.schnorr_build_trace()), log_b)
    aux = backend.lde_columns(backend.interpolate_columns(backend.schnorr_aux_columns()), log_b)
    av = backend.lde_columns(backend.schnorr_assertion_polys(log_n), log_b)
    ta, tb = oracle.random_elements(56, 31), oracle.random_elements(56, 32)
    ba, bb = oracle.random_elements(61, 33), oracle.random_elements(61, 34)
    direct = backend.schnorr_evaluate_constraints(lde, aux, ta, tb, ba, bb, av, log_b, n_sig=n_sig)
    split = backend.schnorr_evaluate_constraints_lde(lde, aux, ta, tb, ba, bb, av, n_sig=n_sig)
    assert torch.equal(direct, split)
    if n_sig == 8:
        ptab = oracle.periodic_table(oracle.schnorr_mask_columns(), log_n, log_b)
        ev = oracle.schnorr_evaluate_transitions(to_numpy_u64(lde), to_numpy_u64(aux), ptab)
        ref = oracle.air_combine(oracle.schnorr_desc(w), to_numpy_u64(lde), ev, ta, tb, ba, bb, log_b, avals=to_numpy_u64(av))
        assert (to_numpy_u64(split) == ref).all()
