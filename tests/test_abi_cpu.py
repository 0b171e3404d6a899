"""CPU-only checks of the product library: it loads, exports every symbol include/cstark.h declares, refuses to
compute without a GPU (no fallback), and its host-side AIR description equals the oracle's restatement."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from certificate_stark_amd import _lib
    return _lib.load()


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "cstark.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(cstark_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 20
    lib = _lib()
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_debug_companion_is_separate_from_the_product_library():
    """include/cstark_debug.h is served by libcstark_debug.so (test-only); the product library exports none of it."""
    hdr = open(os.path.join(ROOT, "include", "cstark_debug.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(cstark_debug_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) == 4
    from certificate_stark_amd import _lib as L
    dbg = L.load_debug()
    lib = _lib()
    assert all(hasattr(dbg, n) for n in names)
    assert not any(hasattr(lib, n) for n in names)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    lib = _lib()
    ctx = C.c_void_p()
    rc = lib.cstark_ctx_create(C.c_int(-1), None, C.byref(ctx))
    assert rc == -2  # CSTARK_ERR_NO_DEVICE
    assert b"no HIP device" in lib.cstark_last_error()
    from certificate_stark_amd.backend import Backend
    from certificate_stark_amd import CstarkError
    try:
        Backend()
        assert False, "Backend() must fail without a GPU"
    except CstarkError as e:
        assert e.code == -2


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import, link or call it."""
    pkg = os.path.join(ROOT, "certificate-stark_amd")
    pat = re.compile(r"(import\s+oracle|from\s+oracle|oracle/|libcs_oracle|\bcso_)")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cuh", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not pat.search(txt), (dirpath, f)


def test_host_air_description_matches_oracle(oracle):
    lib = _lib()
    base, cyc = oracle.tx_constraint_degrees()
    for i in range(115):
        b, c = C.c_uint32(), C.c_uint32()
        assert lib.cstark_tx_constraint_degree(C.c_uint32(i), C.byref(b), C.byref(c)) == 0
        assert (b.value, c.value) == (int(base[i]), int(cyc[i])), i
    assert lib.cstark_tx_constraint_degree(C.c_uint32(115), C.byref(b), C.byref(c)) == -1
    for depth in (3, 7, 15, 31):
        out = np.zeros((48, 1024), np.uint64)
        assert lib.cstark_tx_periodic_columns(C.c_uint32(depth), out.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
        assert (out == oracle.tx_periodic_columns(depth)).all()
    assert lib.cstark_tx_periodic_columns(C.c_uint32(64), out.ctypes.data_as(C.POINTER(C.c_uint64))) == -1
    lib.cstark_field_generator.restype = C.c_uint64
    lib.cstark_field_root_of_unity.restype = C.c_uint64
    lib.cstark_field_lde_offset.restype = C.c_uint64
    assert lib.cstark_field_lde_offset() == oracle.generator()   # the oracle's "generator" is its domain offset
    from oracle import verifier as V
    assert lib.cstark_field_generator() == int(oracle.to_mont([V.CONV["generator"]])[0])
    for k in (1, 10, 20, 23):
        assert lib.cstark_field_root_of_unity(C.c_uint32(k)) == oracle.root_of_unity(k)


def test_small_air_host_descriptions_match_oracle(oracle):
    lib = _lib()
    base = np.zeros(106, np.uint32); cyc = np.zeros(106, np.uint32)
    oracle.lib().cso_merkle_constraint_degrees(base.ctypes.data_as(C.POINTER(C.c_uint32)), cyc.ctypes.data_as(C.POINTER(C.c_uint32)))
    for i in range(106):
        b, c = C.c_uint32(), C.c_uint32()
        assert lib.cstark_air_constraint_degree(C.c_int(1), C.c_uint32(2), C.c_uint32(i), C.byref(b), C.byref(c)) == 0
        assert (b.value, c.value) == (int(base[i]), int(cyc[i]))
    for depth in (3, 15, 31):
        out = np.zeros((33, 512), np.uint64)
        assert lib.cstark_merkle_periodic_columns(C.c_uint32(depth), out.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
        assert (out == oracle.merkle_periodic_columns(depth)).all()
    w, nc, na, lce = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    assert lib.cstark_air_shape(C.c_int(3), C.c_uint32(1), C.byref(w), C.byref(nc), C.byref(na), C.byref(lce)) == 0
    assert (w.value, nc.value, na.value, lce.value) == (2, 2, 2, 1)
    assert lib.cstark_air_shape(C.c_int(2), C.c_uint32(4), C.byref(w), C.byref(nc), C.byref(na), C.byref(lce)) == 0
    assert (w.value, nc.value, na.value, lce.value) == (56, 56, 61, 3)
    for n_sig in (1, 2):
        sb, sc = oracle.schnorr_constraint_degrees(n_sig)
        for i in range(56):
            b, c = C.c_uint32(), C.c_uint32()
            assert lib.cstark_air_constraint_degree(C.c_int(2), C.c_uint32(n_sig), C.c_uint32(i), C.byref(b), C.byref(c)) == 0
            assert (b.value, c.value) == (int(sb[i]), int(sc[i]))
    assert lib.cstark_air_shape(C.c_int(4), C.c_uint32(1), C.byref(w), C.byref(nc), C.byref(na), C.byref(lce)) == 0    # RescueAir, benches/rescue.rs
    assert (w.value, nc.value, na.value, lce.value) == (14, 14, 14, 2)
    out = np.zeros((29, 8), np.uint64)
    assert lib.cstark_rescue_chain_periodic_columns(out.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    assert (out == oracle.rescue_chain_periodic_columns()).all()
    assert lib.cstark_air_shape(C.c_int(5), C.c_uint32(1), C.byref(w), C.byref(nc), C.byref(na), C.byref(lce)) == -5   # no such AIR
    out = np.zeros((36, 512), np.uint64)
    assert lib.cstark_schnorr_mask_columns(out.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    assert (out == oracle.schnorr_mask_columns()).all()


@pytest.mark.parametrize("n_tx,depth,seed", [(4, 3, 7), (8, 7, 0x5EED), (4, 31, 99), (16, 15, 1)])
def test_host_witness_generator_matches_oracle(oracle, n_tx, depth, seed):
    """cstark_tx_witness_generate (product, host C++) against the oracle's generator: same arrays, and the trace built from
    them satisfies all 115 constraints on every row (known answers 6-9 of SURVEY 8(c))."""
    from certificate_stark_amd.prover import TransactionMetadata
    meta = TransactionMetadata.build_random(n_tx, depth, seed)
    w = oracle.TxWitness.generate(n_tx, depth, seed=seed)
    for f in TransactionMetadata.FIELDS:
        assert np.array_equal(getattr(meta, f), getattr(w, f)), f
    assert oracle.tx_check_trace(oracle.tx_build_trace(w), n_tx, depth) == -1  # -1: no violated row


def test_host_schnorr_witness_generator_matches_oracle(oracle):
    lib = _lib()
    n = 3
    msg, rx, s = np.zeros((n, 28), np.uint64), np.zeros((n, 6), np.uint64), np.zeros((n, 32), np.uint8)
    u64p, u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)
    assert lib.cstark_schnorr_witness_generate(C.c_uint32(n), C.c_uint64(99), msg.ctypes.data_as(u64p), rx.ctypes.data_as(u64p),
                                               s.ctypes.data_as(u8p)) == 0
    w = oracle.SchnorrWitness.generate(n, seed=99)
    assert np.array_equal(msg, w.messages) and np.array_equal(rx, w.sig_rx) and np.array_equal(s, w.sig_s)


def test_vectorised_coin_candidates_equal_the_scalar_hash(tmp_path):
    """hostblake3.h::coin_candidates_x8 (eight BLAKE3(seed || counter) per pass on 8-lane vectors; the prover draws its coefficients
    through it) against the scalar implementation: compiled with g++ from the product's header, host code only."""
    import subprocess
    exe = str(tmp_path / "host_coin_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "host_coin_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr


def test_resident_witness_is_read_only_until_invalidated():
    """MerkleExample / SchnorrExample keep their witness in device memory between prove() calls.  A host-side edit must never
    be proved from the stale device copy: the uploaded arrays are read-only while resident, invalidate() or a replaced array uploads anew."""
    import numpy as np
    import pytest
    from certificate_stark_amd.prover import ProofOptions, SchnorrExample

    class FakeBackend:
        resident = None
        uploads = 0

        def upload_schnorr_witness(self, *a):
            self.uploads += 1

        def air_prove(self, air, options, number=0):
            return b"proof"

    b = FakeBackend()
    ex = SchnorrExample(ProofOptions(), np.zeros((2, 28), np.uint64), np.zeros((2, 6), np.uint64), np.zeros((2, 32), np.uint8), backend=b)
    ex.prove(); ex.prove()
    assert b.uploads == 1
    with pytest.raises(ValueError):          # numpy: assignment destination is read-only
        ex.sig_s[1, 3] ^= 1
    ex.invalidate()
    ex.sig_s[1, 3] ^= 1
    ex.prove()
    assert b.uploads == 2
    ex.messages = ex.messages.copy()         # a replaced array is a new witness too
    ex.prove()
    assert b.uploads == 3
    other = SchnorrExample(ProofOptions(), np.zeros((2, 28), np.uint64), np.zeros((2, 6), np.uint64), np.zeros((2, 32), np.uint8), backend=b)
    other.prove(); ex.prove()                # another example took the device copy
    assert b.uploads == 5
