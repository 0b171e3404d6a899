"""GPU parity for the standalone sub-AIRs: MerkleAir and RangeProofAir traces, all transition
constraints and the generic merged constraint evaluations (constraint-evaluation blowup below the LDE blowup),
bit-exact against the oracle through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def backend():
    from certificate_stark_amd.backend import Backend
    b = Backend()
    yield b
    b.close()


@pytest.mark.parametrize("n_tx,depth", [(1, 3), (2, 15), (8, 7)])
def test_merkle_air(oracle, backend, n_tx, depth):
    from certificate_stark_amd.backend import to_numpy_u64
    w = oracle.TxWitness.generate(n_tx, depth, seed=900 + n_tx)
    ref_trace = oracle.merkle_build_trace(w)
    backend.upload_witness(w)
    d_trace = backend.merkle_build_trace()
    assert (to_numpy_u64(d_trace) == ref_trace).all()
    log_b, log_n = 3, ref_trace.shape[1].bit_length() - 1
    co = oracle.interpolate_columns(ref_trace)
    lde = oracle.lde_columns(co, log_b)
    d_lde = backend.lde_columns(backend.interpolate_columns(d_trace), log_b)
    assert (to_numpy_u64(d_lde) == lde).all()
    ptab = oracle.periodic_table(oracle.merkle_periodic_columns(depth), log_n, log_b)
    ref_ev = oracle.air_evaluate_transitions(oracle.AIR_MERKLE, lde, ptab, 106)
    d_ev = backend.air_evaluate_transitions(backend.AIR_MERKLE, d_lde, depth, log_b)
    got = to_numpy_u64(d_ev)
    if not (got == ref_ev).all():
        raise AssertionError("constraints differ: %s" % sorted(set(np.argwhere(got != ref_ev)[:, 1].tolist())))
    desc = oracle.merkle_desc(ref_trace)
    assert backend.air_shape(backend.AIR_MERKLE) == (65, 106, 14, 2) and desc.log_ce == 2
    ta, tb = oracle.random_elements(106, 1), oracle.random_elements(106, 2)
    ba, bb = oracle.random_elements(14, 3), oracle.random_elements(14, 4)
    ref = oracle.air_combine(desc, lde, ref_ev, ta, tb, ba, bb, log_b)
    out = to_numpy_u64(backend.air_combine(backend.AIR_MERKLE, d_lde, d_ev, ta, tb, ba, bb, desc.a_value, log_b))
    assert (out == ref).all()
    assert not out[1::2].any() and out[0::2].any()      # odd cosets are outside the 4n-point evaluation domain
    # the fused evaluator (no materialised transition values): same merged evaluations, whole table and a coset window
    fused = backend.merkle_evaluate_constraints(d_lde, depth, ta, tb, ba, bb, desc.a_value, log_b)
    assert (to_numpy_u64(fused) == ref).all()
    part = backend.merkle_evaluate_constraints(d_lde[3:7].contiguous(), depth, ta, tb, ba, bb, desc.a_value, log_b, k0=3)
    assert (to_numpy_u64(part) == ref[3:7]).all()


@pytest.mark.parametrize("number", [0, 1, 2**63 - 1, 0x0123456789ABCDEF])
def test_range_air(oracle, backend, number):
    from certificate_stark_amd.backend import to_numpy_u64
    number %= 2**62 + 2**56 + 2**55 + 1   # BaseElement::from reduces (the reference's "max input" 2^63 - 1 wraps mod p)
    nm = int(oracle.to_mont([number])[0])
    ref_trace = oracle.range_build_trace(number)
    d_trace = backend.range_build_trace(nm)
    assert (to_numpy_u64(d_trace) == ref_trace).all()
    log_b = 3
    lde = oracle.lde_columns(oracle.interpolate_columns(ref_trace), log_b)
    d_lde = backend.lde_columns(backend.interpolate_columns(d_trace), log_b)
    assert (to_numpy_u64(d_lde) == lde).all()
    ref_ev = oracle.air_evaluate_transitions(oracle.AIR_RANGE, lde, None, 2)
    d_ev = backend.air_evaluate_transitions(backend.AIR_RANGE, d_lde, 0, log_b)
    assert (to_numpy_u64(d_ev) == ref_ev).all()
    desc = oracle.range_desc(number)
    ta, tb = oracle.random_elements(2, 5), oracle.random_elements(2, 6)
    ba, bb = oracle.random_elements(2, 7), oracle.random_elements(2, 8)
    ref = oracle.air_combine(desc, lde, ref_ev, ta, tb, ba, bb, log_b)
    out = to_numpy_u64(backend.air_combine(backend.AIR_RANGE, d_lde, d_ev, ta, tb, ba, bb, desc.a_value, log_b))
    assert (out == ref).all()


def test_range_rejects_non_field_elements(oracle, backend):
    from certificate_stark_amd import CstarkError
    with pytest.raises(CstarkError):
        backend.range_build_trace(2**62 + 2**56 + 2**55 + 1)   # raw value M is not a field element (src/range/tests.rs:54-62: should_panic)


@pytest.mark.parametrize("n_sig", [1, 2, 8])
def test_schnorr_air_trace_and_transitions(oracle, backend, n_sig):
    """SchnorrAir (src/schnorr): trace, public-input columns and all 56 transition constraints over the LDE."""
    from certificate_stark_amd.backend import to_numpy_u64
    w = oracle.SchnorrWitness.generate(n_sig, seed=4000 + n_sig)
    ref_trace = oracle.schnorr_build_trace(w)
    backend.upload_schnorr_witness(w.messages, w.sig_rx, w.sig_s)
    d_trace = backend.schnorr_build_trace()
    got = to_numpy_u64(d_trace)
    if not (got == ref_trace).all():
        raise AssertionError("trace columns differ: %s" % sorted(set(np.argwhere(got != ref_trace)[:, 0].tolist())))
    for t in range(n_sig):   # the assertion of src/schnorr/air.rs:217-224: x(s*G + h*P) == R.x at step 511 of each block
        assert (got[0:6, 512 * t + 511] == w.sig_rx[t]).all()
    ref_aux = oracle.schnorr_aux_columns(w)
    d_aux = backend.schnorr_aux_columns()
    assert (to_numpy_u64(d_aux) == ref_aux).all()
    log_b, log_n = 3, ref_trace.shape[1].bit_length() - 1
    lde = oracle.lde_columns(oracle.interpolate_columns(ref_trace), log_b)
    aux_lde = oracle.lde_columns(oracle.interpolate_columns(ref_aux), log_b)
    d_lde = backend.lde_columns(backend.interpolate_columns(d_trace), log_b)
    d_aux_lde = backend.lde_columns(backend.interpolate_columns(d_aux), log_b)
    assert (to_numpy_u64(d_lde) == lde).all() and (to_numpy_u64(d_aux_lde) == aux_lde).all()
    ptab = oracle.periodic_table(oracle.schnorr_mask_columns(), log_n, log_b)
    ref_ev = oracle.schnorr_evaluate_transitions(lde, aux_lde, ptab)
    got_ev = to_numpy_u64(backend.schnorr_evaluate_transitions(d_lde, d_aux_lde, log_b))
    if not (got_ev == ref_ev).all():
        raise AssertionError("constraints differ: %s" % sorted(set(np.argwhere(got_ev != ref_ev)[:, 1].tolist())))
    part = to_numpy_u64(backend.schnorr_evaluate_transitions(d_lde[3:5].contiguous(), d_aux_lde[3:5].contiguous(), log_b, k0=3))
    assert (part == ref_ev[3:5]).all()


@pytest.mark.parametrize("n_sig", [1, 4])
def test_schnorr_air_merged_evaluations(oracle, backend, n_sig):
    """SchnorrAir end to end: transitions merged with the 61 periodic / sequence assertions (src/schnorr/air.rs:111-226)."""
    from certificate_stark_amd.backend import to_numpy_u64
    w = oracle.SchnorrWitness.generate(n_sig, seed=77 + n_sig)
    trace = oracle.schnorr_build_trace(w)
    log_b, log_n = 3, trace.shape[1].bit_length() - 1
    lde = oracle.lde_columns(oracle.interpolate_columns(trace), log_b)
    aux_lde = oracle.lde_columns(oracle.interpolate_columns(oracle.schnorr_aux_columns(w)), log_b)
    ev = oracle.schnorr_evaluate_transitions(lde, aux_lde, oracle.periodic_table(oracle.schnorr_mask_columns(), log_n, log_b))
    desc = oracle.schnorr_desc(w)
    polys = oracle.schnorr_assertion_polys(w, log_n)
    avals = oracle.lde_columns(polys, log_b)
    ta, tb = oracle.random_elements(56, 1), oracle.random_elements(56, 2)
    ba, bb = oracle.random_elements(61, 3), oracle.random_elements(61, 4)
    ref = oracle.air_combine(desc, lde, ev, ta, tb, ba, bb, log_b, avals=avals)
    backend.upload_schnorr_witness(w.messages, w.sig_rx, w.sig_s)
    assert backend.air_shape(backend.AIR_SCHNORR, n_sig) == (56, 56, 61, 3)
    d_polys = backend.schnorr_assertion_polys(log_n)
    assert (to_numpy_u64(d_polys) == polys).all()
    d_avals = backend.lde_columns(d_polys, log_b)
    out = backend.air_combine(backend.AIR_SCHNORR, backend.from_numpy_u64(lde), backend.from_numpy_u64(ev), ta, tb, ba, bb, None, log_b,
                              n_items=n_sig, avals_lde=d_avals)
    assert (to_numpy_u64(out) == ref).all()
    # the fused evaluator (no materialised transition values): same merged evaluations, whole table and a coset window
    d_lde, d_aux = backend.from_numpy_u64(lde), backend.from_numpy_u64(aux_lde)
    fused = backend.schnorr_evaluate_constraints(d_lde, d_aux, ta, tb, ba, bb, d_avals, log_b, n_sig=n_sig)
    assert (to_numpy_u64(fused) == ref).all()
    part = backend.schnorr_evaluate_constraints(d_lde[2:5].contiguous(), d_aux[2:5].contiguous(), ta, tb, ba, bb, d_avals[2:5].contiguous(), log_b,
                                                k0=2, n_sig=n_sig)
    assert (to_numpy_u64(part) == ref[2:5]).all()
