"""End-to-end proofs through cstark_tx_prove, checked by the restated verifier (oracle/verifier.py).
Mirrors the reference's acceptance tests /root/reference/src/tests.rs:11-38 (prove -> verify ok; wrong public inputs ->
error) with the cfg(test) tree depth 3 (src/merkle/constants.rs:22) and the production depth 15."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def build_options(queries=42):
    from certificate_stark_amd.prover import ProofOptions
    return ProofOptions(queries, 8, 0, ProofOptions.BLAKE3_256, ProofOptions.EXT_NONE, 4, 256)  # src/tests.rs:40-55


def example(n_tx, depth, seed=0x5EED, options=None):
    from oracle import oracle as O
    from certificate_stark_amd.prover import TransactionExample, TransactionMetadata
    w = O.TxWitness.generate(n_tx, depth, seed=seed)
    meta = TransactionMetadata(*[getattr(w, f) for f in TransactionMetadata.FIELDS])
    return TransactionExample(options or build_options(), meta)


def test_transaction_basic_proof_verification():
    from oracle import verifier as V
    tx = example(2, 3)
    proof = tx.prove()
    assert V.verify(proof, *tx.pub_inputs(), options=[42, 8, 0, 0, 0, 4, 256])


def test_transaction_basic_proof_verification_fail():
    """verify_with_wrong_inputs, src/lib.rs:152-161: final_root replaced by [final_root[0]; 7]."""
    from oracle import verifier as V
    tx = example(2, 3)
    proof = tx.prove()
    initial_root, final_root = tx.pub_inputs()
    with pytest.raises(V.VerifierError):
        V.verify(proof, initial_root, np.full(7, final_root[0], np.uint64))
    with pytest.raises(V.VerifierError):
        V.verify(proof, final_root, final_root)


def test_proof_is_deterministic_and_binds_every_section():
    from oracle import verifier as V
    tx = example(2, 3, seed=11)
    proof = tx.prove()
    assert proof == tx.prove()
    pub = tx.pub_inputs()
    d = V.parse(proof)
    nq, log_N = 42, d["log_n"] + 3
    hdr = 4 + 4 + 16 + 28
    sections = {
        "trace_root": hdr + 3, "cons_root": hdr + 32 + 5, "layer_root": hdr + 64 + 4 + 7,
        "ood_trace": hdr + 64 + 4 + 32 * len(d["layer_roots"]) + 32 + 8 * 10 + 1,
        "ood_comp": hdr + 64 + 4 + 32 * len(d["layer_roots"]) + 32 + 8 * 188 + 9,
    }
    base = hdr + 64 + 4 + 32 * len(d["layer_roots"]) + 32 + 8 * 196 + 8
    sections["trace_row"] = base + 8 * 94 * 5 + 16
    sections["trace_path"] = base + nq * 94 * 8 + 32 * log_N * 3 + 40
    sections["cons_row"] = base + nq * 94 * 8 + nq * log_N * 32 + 8 * 8 * 7 + 3
    sections["remainder"] = len(proof) - 8 * 17
    for name, off in sections.items():
        bad = bytearray(proof)
        bad[off] ^= 0x01
        with pytest.raises(V.VerifierError):
            V.verify(bytes(bad), *pub)
    with pytest.raises(V.VerifierError):
        V.verify(proof[:-8], *pub)


def test_invalid_witness_gives_unverifiable_proof():
    """A transfer whose signature is wrong still produces a proof, but not one that verifies (src/lib.rs:212-218 note)."""
    from oracle import oracle as O
    from oracle import verifier as V
    from certificate_stark_amd.prover import TransactionExample, TransactionMetadata
    w = O.TxWitness.generate(2, 3, seed=5)
    w.deltas[1] ^= np.uint64(1 << 20)
    meta = TransactionMetadata(*[getattr(w, f) for f in TransactionMetadata.FIELDS])
    tx = TransactionExample(build_options(), meta)
    with pytest.raises(V.VerifierError):
        V.verify(tx.prove(), *tx.pub_inputs())


@pytest.mark.parametrize("n_tx,depth,queries", [(16, 15, 42), (64, 15, 96)])
def test_production_depth_proofs(n_tx, depth, queries):
    from oracle import verifier as V
    tx = example(n_tx, depth, options=build_options(queries))
    proof = tx.prove()
    assert V.verify(proof, *tx.pub_inputs())
    stages = tx.prover.backend.prove_stage_ms()
    assert set(stages) == set(tx.prover.backend.PROVE_STAGES) and all(v >= 0 for v in stages.values())


def test_unsupported_options_are_refused():
    from certificate_stark_amd._lib import CstarkError
    from certificate_stark_amd.prover import ProofOptions
    for opt in (ProofOptions(42, 4), ProofOptions(42, 32), ProofOptions(42, 8, 0, 2), ProofOptions(42, 8, 0, 0, 3), ProofOptions(42, 8, 0, 0, 0, 2),
                ProofOptions(42, 8, 0, 0, 0, 32)):  # (blowup 16, folding 8 / 16: supported, tests/test_gpu_options.py)
        tx = example(2, 3, options=opt)
        with pytest.raises(CstarkError):
            tx.prove()


@pytest.mark.parametrize("n_tx,depth,opts", [(2, 3, (42, 8, 0, 0, 0, 4, 256)), (8, 15, (28, 8, 0, 0, 0, 4, 128)),
                                             (4, 7, (16, 8, 8, 0, 0, 4, 1024)), (64, 15, (96, 8, 0, 0, 0, 4, 256)),
                                             (2, 3, (20, 8, 15, 0, 0, 4, 256)), (2, 3, (20, 8, 13, 0, 1, 4, 256)),
                                             (2, 3, (20, 8, 13, 1, 0, 4, 256))])
def test_proof_bytes_equal_the_cpu_restatement(n_tx, depth, opts):
    """Bit-exact whole-pipeline parity (trace -> LDE -> commitments -> constraints -> composition -> DEEP -> FRI -> openings),
    including a proof-of-work nonce search (grinding 8 on the host; 15 and, with the quadratic extension, 13 bits by the device-side
    search, the last case with the Sha3 coin) and the other remainder sizes."""
    from oracle import oracle as O
    from oracle import prover as OP
    from oracle import verifier as V
    from certificate_stark_amd.prover import ProofOptions, TransactionExample, TransactionMetadata
    w = O.TxWitness.generate(n_tx, depth, seed=77 + n_tx)
    meta = TransactionMetadata(*[getattr(w, f) for f in TransactionMetadata.FIELDS])
    tx = TransactionExample(ProofOptions(*opts), meta)
    proof = tx.prove()
    assert proof == OP.prove(w, opts)
    assert V.verify(proof, *tx.pub_inputs(), options=list(opts))


def test_one_context_across_options_reuses_its_arena_safely():
    """One Backend (one cstark_ctx, one arena) proving under changing options: a base-field proof with 96 queries, a cubic proof
    with 42 queries, a cubic proof with 96 queries, a base-field proof again.  The arena's per-option buffers must grow with the
    request (the openings buffer of the extension path was once sized by the first extension proof's query count); every proof
    equals the CPU restatement's bytes and verifies."""
    from oracle import oracle as O
    from oracle import prover as OP
    from oracle import verifier as V
    from certificate_stark_amd.backend import Backend
    from certificate_stark_amd.prover import ProofOptions, TransactionMetadata
    w = O.TxWitness.generate(2, 3, seed=31)
    meta = TransactionMetadata(*[getattr(w, f) for f in TransactionMetadata.FIELDS])
    b = Backend()
    b.upload_witness(meta)
    for opts in ((96, 8, 0, 0, 0, 4, 256), (42, 8, 0, 0, 2, 4, 256), (96, 8, 0, 0, 2, 4, 256), (128, 8, 0, 0, 1, 4, 256), (96, 8, 0, 0, 0, 4, 256)):
        proof = b.prove(ProofOptions(*opts))
        assert proof == OP.prove(w, opts), opts
        assert V.verify(proof, w.initial_roots[0], w.final_root, options=list(opts))
    b.close()


def test_smallest_and_largest_traces():
    """One transfer (2^10 rows, 2^13-point domain) and 2048 transfers (2^21 rows: the 2^24-point domain is the library's maximum)."""
    import torch
    from oracle import verifier as V
    from certificate_stark_amd.prover import TransactionExample, TransactionMetadata
    tx = example(1, 3, seed=123)
    proof = tx.prove()
    assert V.verify(proof, *tx.pub_inputs())
    tx.prover.backend.close()
    meta = TransactionMetadata.build_random(2048, 15, seed=2048)
    big = TransactionExample(build_options(), meta)
    proof = big.prove()
    d = V.parse(proof)
    assert d["log_n"] == 21 and len(d["remainder"]) == 256
    assert V.verify(proof, *big.pub_inputs())
    big.prover.backend.close()
    torch.cuda.empty_cache()


def test_transaction_count_must_be_a_power_of_two():
    from certificate_stark_amd._lib import CstarkError
    tx = example(3, 3)
    with pytest.raises(ValueError):            # the host mirror checks first
        tx.prove()
    b = tx.prover.backend
    b.upload_witness(tx.tx_metadata)
    with pytest.raises(CstarkError):           # and so does the library
        b.prove(build_options())


def test_two_contexts_prove_concurrently():
    """The library is re-entrant per context (include/cstark.h): two host threads, each with its own context and stream, prove
    different witnesses at the same time; both proofs verify and equal the proofs made alone."""
    import threading
    import torch
    from oracle import verifier as V
    from certificate_stark_amd.backend import Backend
    from certificate_stark_amd.prover import TransactionExample
    jobs = [example(4, 7, seed=41), example(8, 15, seed=42)]
    alone = [tx.prove() for tx in jobs]
    for tx in jobs:
        tx.prover.backend.close()
    results = [None, None]

    def work(i):
        with torch.cuda.stream(torch.cuda.Stream()):
            tx = TransactionExample(build_options(), jobs[i].tx_metadata, Backend())
            for _ in range(3):
                results[i] = tx.prove()
            tx.prover.backend.close()

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i, tx in enumerate(jobs):
        assert results[i] == alone[i]
        assert V.verify(results[i], *tx.pub_inputs())


def test_proof_buffer_too_small_reports_required_size():
    import ctypes as C
    from certificate_stark_amd import _lib
    tx = example(2, 3)
    b = tx.prover.backend
    b.upload_witness(tx.tx_metadata)
    o = _lib.OptionsStruct(42, 8, 0, 0, 0, 4, 256)
    need = C.c_size_t(0)
    small = (C.c_uint8 * 16)()
    rc = b.lib.cstark_tx_prove(b.ctx, C.byref(o), small, C.c_size_t(16), C.byref(need))
    assert rc == -1 and need.value > 100000
    buf = (C.c_uint8 * need.value)()
    n2 = C.c_size_t(0)
    assert b.lib.cstark_tx_prove(b.ctx, C.byref(o), buf, C.c_size_t(need.value), C.byref(n2)) == 0 and n2.value == need.value


@pytest.mark.parametrize("n_tx,depth", [(2, 3), (8, 15)])
def test_sha3_proofs(n_tx, depth):
    """HashFunction::Sha3_256 (the reference's other ProofOptions hash, examples/state-transition.rs:67-71): commitments, channel
    and openings all switch; bytes equal the CPU restatement, the verifier accepts, a Blake3 reading of the same proof fails."""
    from oracle import oracle as O
    from oracle import prover as OP
    from oracle import verifier as V
    from certificate_stark_amd.prover import ProofOptions, TransactionExample, TransactionMetadata
    opts = (42, 8, 0, 1, 0, 4, 256)
    w = O.TxWitness.generate(n_tx, depth, seed=300 + n_tx)
    meta = TransactionMetadata(*[getattr(w, f) for f in TransactionMetadata.FIELDS])
    tx = TransactionExample(ProofOptions(*opts), meta)
    proof = tx.prove()
    assert proof == OP.prove(w, opts)
    assert V.verify(proof, *tx.pub_inputs(), options=list(opts))
    as_blake = bytearray(proof)
    as_blake[4 + 4 + 16 + 12] = 0   # the hash_fn word of the header
    with pytest.raises(V.VerifierError):
        V.verify(bytes(as_blake), *tx.pub_inputs())


@pytest.mark.parametrize("n_tx,depth,hash_fn,ext", [(2, 3, 0, 1), (8, 15, 0, 1), (4, 7, 1, 1), (2, 3, 0, 2), (8, 15, 0, 2), (4, 7, 1, 2)])
def test_extension_field_proofs(n_tx, depth, hash_fn, ext):
    """transaction_test_basic_proof_verification_quadratic_extension / _cubic_extension (src/tests.rs:18-30): FieldExtension::
    Quadratic and ::Cubic.  The proof bytes equal the CPU restatement's, the restated verifier accepts them (evaluating the AIR over
    the extension by interpolation), wrong public inputs and tampering are rejected."""
    from oracle import oracle as O
    from oracle import prover as OP
    from oracle import verifier as V
    from certificate_stark_amd.prover import ProofOptions, TransactionExample, TransactionMetadata
    opts = (42, 8, 0, hash_fn, ext, 4, 256)
    w = O.TxWitness.generate(n_tx, depth, seed=600 + n_tx)
    meta = TransactionMetadata(*[getattr(w, f) for f in TransactionMetadata.FIELDS])
    tx = TransactionExample(ProofOptions(*opts), meta)
    proof = tx.prove()
    ref = OP.prove(w, opts)
    assert len(proof) == len(ref)
    assert proof == ref
    assert V.verify(proof, *tx.pub_inputs(), options=list(opts))
    r0, r1 = tx.pub_inputs()
    with pytest.raises(V.VerifierError):
        V.verify(proof, r0, np.full(7, r1[0], np.uint64))
    bad = bytearray(proof)
    bad[len(bad) - 40] ^= 2
    with pytest.raises(V.VerifierError):
        V.verify(bytes(bad), r0, r1)


def test_repeated_proofs_are_identical():
    """The prover overlaps its trace recurrences with the transforms on side streams, polls its stream at the channel's wait points and
    gathers every opening in one launch: forty proofs of one witness on one context (buffers reused each time) must be the same bytes."""
    import hashlib
    tx = example(16, 15, seed=4242)
    want = hashlib.sha256(tx.prove()).hexdigest()
    for _ in range(40):
        assert hashlib.sha256(tx.prove()).hexdigest() == want


def test_direct_and_split_prover_paths_give_the_same_bytes():
    """Inside cstark_tx_prove the constraints run as the degree-split evaluation and the trace is committed in overlapped column
    batches; CSTARK_ROUNDS_SPLIT=0 / CSTARK_TRACE_OVERLAP=0 select the direct forms (read once per process, hence the child
    process).  Both must produce the default path's proof, byte for byte."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_gpu_prove import example\n"
            "print(hashlib.sha256(example(4, 15, seed=77).prove()).hexdigest())\n") % (root, os.path.join(root, "tests"))
    want = hashlib.sha256(example(4, 15, seed=77).prove()).hexdigest()
    env = dict(os.environ, CSTARK_ROUNDS_SPLIT="0", CSTARK_TRACE_OVERLAP="0", CSTARK_SYNC_BLOCK="1",  # also: blocking channel waits,
               CSTARK_MERKLE_QUAD="0", CSTARK_FRI_DEVICE_COIN="0", CSTARK_COIN_SCALAR="1")  # lane-per-node tree tails, host-side FRI and scalar coin
    got = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert got.returncode == 0, got.stderr[-2000:]
    assert got.stdout.strip().splitlines()[-1] == want


def _channel_digests():
    """sha256 of TransactionAir proofs the device-side channel covers (Blake3 coin, base field, no proof of work) at several sizes and
    option sets, on a fresh backend: printed by the child process of the test below, computed in-process by the test itself."""
    import hashlib
    from certificate_stark_amd.prover import ProofOptions, TransactionMetadata
    out = []
    for n_tx, depth, opts in [(2, 3, (42, 8, 0, 0, 0, 4, 256)), (8, 15, (96, 8, 0, 0, 0, 4, 128)), (4, 7, (28, 16, 0, 0, 0, 8, 1024)), (16, 15, (128, 8, 0, 0, 0, 16, 256))]:
        tx = example(n_tx, depth, options=ProofOptions(*opts), seed=300 + n_tx)
        out.append(hashlib.sha256(tx.prove()).hexdigest())
    return out


def test_host_and_device_channel_give_the_same_bytes():
    """Round 4: the Fiat-Shamir channel of these proofs runs on the device (csrc/channel.hip: seed, reseeds, draws, query positions and
    their folds); CSTARK_HOST_CHANNEL=1 -- read once per process -- keeps the C++ host channel.  Both must write the same proofs (and
    test_proof_bytes_equal_the_cpu_restatement compares the device channel with the CPU prover)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_gpu_prove import _channel_digests\n"
            "print(' '.join(_channel_digests()))\n") % (root, os.path.join(root, "tests"))
    got = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CSTARK_HOST_CHANNEL="1"), capture_output=True, text=True, timeout=600)
    assert got.returncode == 0, got.stderr[-2000:]
    assert got.stdout.strip().splitlines()[-1].split() == _channel_digests()
