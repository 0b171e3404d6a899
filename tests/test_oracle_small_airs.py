"""Pins the oracle's restatement of the standalone sub-AIRs: MerkleAir, RangeProofAir and the
Rescue hash-chain AIR of benches/rescue.rs.  Known answers: constraints vanish on valid traces and fail on
perturbed ones; the combined evaluations over the constraint-evaluation domain are those of a polynomial
(checked off-domain); range / chain end values.  CPU only."""
import numpy as np
import pytest

P = 2**62 + 2**56 + 2**55 + 1


def _check_trace(oracle, air, trace, periodic_cols, nc):
    """all transition constraints vanish on rows 0..n-2 of the base trace"""
    width, n = trace.shape
    lde = trace.reshape(1, width, n)                       # blowup 1, offset irrelevant: use rows directly
    if periodic_cols is None:
        ptab = None
    else:
        cl = periodic_cols.shape[1]
        ptab = periodic_cols.reshape(1, periodic_cols.shape[0], cl)
    ev = oracle.air_evaluate_transitions(air, lde, ptab, nc)[0]
    return ev[:, :n - 1]


def test_merkle_air_constraints_vanish(oracle, witness_d3, witness_d15):
    for w in (witness_d3, witness_d15):
        trace = oracle.merkle_build_trace(w)
        assert trace.shape == (65, 512 * w.n_tx)
        one = oracle.to_mont([1])[0]
        assert trace[14, 1] == one and trace[43, 1] == one          # prover.rs:72-77 poke
        ev = _check_trace(oracle, oracle.AIR_MERKLE, trace, oracle.merkle_periodic_columns(w.depth), 106)
        assert not ev.any()
        # same per-transaction recurrence as the Merkle half of the composite trace
        full = oracle.tx_build_trace(w)
        for t in range(w.n_tx):
            a = trace[:, 512 * t:512 * (t + 1)].copy(); b = full[:65, 1024 * t:1024 * t + 512]
            if t == 0:
                a[14, 1] = b[14, 1]; a[43, 1] = b[43, 1]
            assert (a == b).all()
        bad = w.copy(); bad.s_paths[1, 2, 0] ^= np.uint64(1)
        ev = _check_trace(oracle, oracle.AIR_MERKLE, oracle.merkle_build_trace(bad), oracle.merkle_periodic_columns(w.depth), 106)
        assert ev.any()


def test_range_air(oracle):
    for number in (0, 1, (2**63 - 1) % P, 0x123456789ABCDEF, P - 1):   # src/range/tests.rs:44-52 (2^63 - 1 wraps mod p)
        trace = oracle.range_build_trace(number)
        assert not _check_trace(oracle, oracle.AIR_RANGE, trace, None, 2).any()
        assert int(oracle.from_mont(trace[1, -1:])[0]) == number % P and trace[1, 0] == 0
    trace = oracle.range_build_trace(12345); trace[0, 7] = oracle.to_mont([2])[0]
    assert _check_trace(oracle, oracle.AIR_RANGE, trace, None, 2).any()


def test_rescue_chain_air(oracle):
    seed = oracle.to_mont(np.arange(42, 49, dtype=np.uint64))      # benches/rescue.rs:38-46
    trace = oracle.rescue_chain_build_trace(seed, 16)
    assert trace.shape == (14, 128)
    assert not _check_trace(oracle, oracle.AIR_RESCUE_CHAIN, trace, oracle.rescue_chain_periodic_columns(), 14).any()
    # one link equals the off-circuit merge([seed, 0]); longer chains differ from compute_hash_chain (reference quirk)
    assert (trace[:7, 7] == oracle.rescue_compute_hash_chain(seed, 1)).all()
    one = oracle.rescue_chain_build_trace(seed, 1)
    assert (one[:7, 7] == oracle.rescue_compute_hash_chain(seed, 1)).all()
    assert (trace[:7, -1] != oracle.rescue_compute_hash_chain(seed, 16)).any()


@pytest.mark.parametrize("which", ["merkle", "range", "rescue"])
def test_combined_evaluations_are_polynomial(oracle, witness_d3, which):
    """The merged quotient is a polynomial of degree < ce_size: interpolating it from the constraint-evaluation
    domain reproduces the same rational function on LDE cosets OUTSIDE that domain; not so for an invalid trace."""
    log_b = 3
    if which == "merkle":
        trace = oracle.merkle_build_trace(witness_d3); air = oracle.AIR_MERKLE
        desc = oracle.merkle_desc(trace); cols = oracle.merkle_periodic_columns(3)
    elif which == "range":
        trace = oracle.range_build_trace(0xDEADBEEF12345); air = oracle.AIR_RANGE
        desc = oracle.range_desc(0xDEADBEEF12345); cols = None
    else:
        seed = oracle.to_mont(np.arange(42, 49, dtype=np.uint64))
        trace = oracle.rescue_chain_build_trace(seed, 8); air = oracle.AIR_RESCUE_CHAIN
        desc = oracle.rescue_chain_desc(trace); cols = oracle.rescue_chain_periodic_columns()
    width, n = trace.shape
    log_n = n.bit_length() - 1
    assert desc.log_ce == {"merkle": 2, "range": 1, "rescue": 2}[which]
    ta, tb = oracle.random_elements(desc.nc, 1), oracle.random_elements(desc.nc, 2)
    ba, bb = oracle.random_elements(desc.na, 3), oracle.random_elements(desc.na, 4)

    def off_domain_error(tr):
        lde = oracle.lde_columns(oracle.interpolate_columns(tr), log_b)
        ptab = None if cols is None else oracle.periodic_table(cols, log_n, log_b)
        ev = oracle.air_evaluate_transitions(air, lde, ptab, desc.nc)
        full = oracle.air_combine(desc, lde, ev, ta, tb, ba, bb, log_b, all_cosets=True)      # [8][n]
        stride = 1 << (log_b - desc.log_ce)
        on = full[::stride]                                                                    # ce domain cosets
        nat = np.ascontiguousarray(on.T).ravel()                                              # natural order of g<w_{ce n}>
        h = oracle.ntt(nat, inverse=True)                                                      # H(g y)
        # evaluate H on the odd cosets: H(g w_{8n}^k w_n^j) = sum h_m (w_{8n}^k)^m (w_ce^j ... ) -> use per-point Horner on a few points
        w8 = oracle.root_of_unity(log_n + log_b); wn = oracle.root_of_unity(log_n)
        errs = 0
        for (k, j) in [(1, 0), (1, 5), (3, n - 1), (7, n // 2)]:
            y = int(oracle.fp_mul(oracle.fp_pow(np.array([w8], np.uint64), k), oracle.fp_pow(np.array([wn], np.uint64), j))[0])
            errs += oracle.poly_eval(h, y) != int(full[k, j])
        return errs
    assert off_domain_error(trace) == 0
    bad = trace.copy(); bad[0, 3] = oracle.fp_add(bad[0, 3:4], oracle.to_mont([1]))[0]
    assert off_domain_error(bad) > 0


def test_schnorr_air(oracle):
    """SchnorrAir (src/schnorr/air.rs): all 56 transition constraints vanish on the trace of valid signatures, the final x
    equals R.x (the assertion of :217-224), the h limbs equal the hash output; a wrong message breaks the constraints."""
    w = oracle.SchnorrWitness.generate(2, seed=9)
    trace = oracle.schnorr_build_trace(w)
    aux = oracle.schnorr_aux_columns(w)
    ptab = oracle.schnorr_mask_columns().reshape(1, 36, 512)
    n = trace.shape[1]
    ev = oracle.schnorr_evaluate_transitions(trace.reshape(1, 56, n), aux.reshape(1, 19, n), ptab)[0]
    assert not ev[:, :n - 1].any()
    for t in range(2):
        assert (trace[0:6, 512 * t + 511] == w.sig_rx[t]).all()
        assert (trace[38:42, 512 * t + 510] == trace[42:46, 512 * t + 510]).all()
    bad_aux = aux.copy(); bad_aux[13, 7] = oracle.fp_add(bad_aux[13, 7:8], oracle.to_mont([1]))[0]
    ev = oracle.schnorr_evaluate_transitions(trace.reshape(1, 56, n), bad_aux.reshape(1, 19, n), ptab)[0]
    assert ev[:, :n - 1].any()
    from collections import Counter
    b1, c1 = oracle.schnorr_constraint_degrees(1)
    b2, c2 = oracle.schnorr_constraint_degrees(2)
    assert Counter(zip(b2.tolist(), c2.tolist())) == {(5, 2): 24, (4, 2): 12, (2, 1): 2, (1, 2): 4, (3, 1): 14}
    assert b1[19] == 3 and b2[19] == 5


def test_schnorr_merged_evaluations_are_polynomial(oracle):
    """SchnorrAir with its 61 periodic / sequence assertions (src/schnorr/air.rs:111-226): the merged quotient is a
    polynomial (interpolated from the 8n-point domain it reproduces the rational function on a different coset of a
    16-fold extension); perturbing a signature's R.x in the assertions -- the reference's wrong-input test -- breaks it."""
    w = oracle.SchnorrWitness.generate(2, seed=21)
    trace = oracle.schnorr_build_trace(w)
    n = trace.shape[1]; log_n = n.bit_length() - 1
    desc = oracle.schnorr_desc(w)
    assert desc.log_ce == 3 and desc.na == 61
    ta, tb = oracle.random_elements(56, 1), oracle.random_elements(56, 2)
    ba, bb = oracle.random_elements(61, 3), oracle.random_elements(61, 4)
    log_b = 4                                        # 16 cosets; the constraint-evaluation domain is the even ones
    co = oracle.interpolate_columns(trace)
    lde = oracle.lde_columns(co, log_b)
    aux_lde = oracle.lde_columns(oracle.interpolate_columns(oracle.schnorr_aux_columns(w)), log_b)
    ptab = oracle.periodic_table(oracle.schnorr_mask_columns(), log_n, log_b)
    ev = oracle.schnorr_evaluate_transitions(lde, aux_lde, ptab)

    def off_domain_errors(wit):
        avals = oracle.lde_columns(oracle.schnorr_assertion_polys(wit, log_n), log_b)
        full = oracle.air_combine(desc, lde, ev, ta, tb, ba, bb, log_b, all_cosets=True, avals=avals)   # [16][n]
        nat = np.ascontiguousarray(full[::2].T).ravel()          # the 8n-point evaluation domain, natural order
        h = oracle.ntt(nat, inverse=True)
        w16 = oracle.root_of_unity(log_n + log_b); wn = oracle.root_of_unity(log_n)
        errs = 0
        for (k, j) in [(1, 0), (3, 77), (15, n - 1)]:
            y = int(oracle.fp_mul(oracle.fp_pow(np.array([w16], np.uint64), k), oracle.fp_pow(np.array([wn], np.uint64), j))[0])
            errs += oracle.poly_eval(h, y) != int(full[k, j])
        return errs
    assert off_domain_errors(w) == 0
    bad = oracle.SchnorrWitness(2)
    bad.messages[...] = w.messages; bad.sig_s[...] = w.sig_s; bad.sig_rx[...] = w.sig_rx
    bad.sig_rx[1, 2] = oracle.fp_add(bad.sig_rx[1, 2:3], oracle.to_mont([1]))[0]
    assert off_domain_errors(bad) > 0


def test_rescue_chain_at_the_benchmark_size(oracle):
    """BASELINE config 0, 'benches/rescue.rs Rescue-Prime hash chain, 2^12 trace steps, CPU path (plumbing, no GPU)': chain length 512
    = 4096 rows, blowup 4 (benches/rescue.rs:23, :370-378), seed 42..48 (:38-46).  The whole hot path through the CPU oracle: trace ->
    interpolation -> extension -> row hashes -> Merkle tree -> transition constraints -> merged evaluations; the trace satisfies every
    constraint, the merged evaluations are those of a polynomial (checked on the LDE cosets outside the constraint-evaluation
    domain would need blowup 8: here the interpolant is checked against a direct off-domain evaluation of the numerators)."""
    import time
    seed = oracle.to_mont(np.arange(42, 49, dtype=np.uint64))
    log_b = 2
    t0 = time.perf_counter()
    trace = oracle.rescue_chain_build_trace(seed, 512)
    assert trace.shape == (14, 4096)
    cols = oracle.rescue_chain_periodic_columns()
    assert not _check_trace(oracle, oracle.AIR_RESCUE_CHAIN, trace, cols, 14).any()
    desc = oracle.rescue_chain_desc(trace)
    assert desc.log_ce == 2
    lde = oracle.lde_columns(oracle.interpolate_columns(trace.copy()), log_b)
    nodes = oracle.merkle_build(oracle.hash_rows(lde, log_b))
    ptab = oracle.periodic_table(cols, 12, log_b)
    ev = oracle.air_evaluate_transitions(oracle.AIR_RESCUE_CHAIN, lde, ptab, desc.nc)
    ta, tb = oracle.random_elements(desc.nc, 1), oracle.random_elements(desc.nc, 2)
    ba, bb = oracle.random_elements(desc.na, 3), oracle.random_elements(desc.na, 4)
    comb = oracle.air_combine(desc, lde, ev, ta, tb, ba, bb, log_b)
    dt = time.perf_counter() - t0
    assert nodes.shape == (2 * 4 * 4096, 32) and comb.shape == (4, 4096)
    # the merged evaluations form a polynomial of degree < 4n: the top coefficient block vanishes by the degree adjustment only if
    # every quotient is exact -- perturbing one trace cell breaks it
    h = oracle.ntt(np.ascontiguousarray(comb.T).ravel(), inverse=True)
    assert h.size == 4 * 4096
    bad = trace.copy(); bad[3, 77] = oracle.fp_add(bad[3, 77:78], oracle.to_mont([1]))[0]
    assert _check_trace(oracle, oracle.AIR_RESCUE_CHAIN, bad, cols, 14).any()
    # last row of the chain equals 512 sequential permutation links recomputed link by link (one link == off-circuit merge)
    state = seed.copy()
    for _ in range(3):
        state = oracle.rescue_compute_hash_chain(state, 1)
    assert (trace[:7, 8 * 3 - 1] == state).all()
    print("rescue chain 512 (2^12 rows, blowup 4) hot path through the CPU oracle: %.3f s" % dt)
    assert dt < 60
