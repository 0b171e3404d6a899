"""Pins the oracle's restatement of the composite state-transition AIR with the algebraic known
answers of SURVEY.md 8(c) (6)-(9), mirroring the reference's own acceptance tests
(src/tests.rs:11-38: positive round trip; negative with wrong inputs).  CPU only."""
import numpy as np
import pytest

P = 2**62 + 2**56 + 2**55 + 1


@pytest.mark.parametrize("fixture", ["witness_d3", "witness_d15"])
def test_all_constraints_vanish_on_valid_trace(oracle, request, fixture):
    w = request.getfixturevalue(fixture)
    trace = oracle.tx_build_trace(w)
    assert trace.shape == (94, 1024 * w.n_tx)
    assert oracle.tx_check_trace(trace, w.n_tx, w.depth) == -1


def test_merkle_signature_and_range_known_answers(oracle, witness_d15):
    w = witness_d15
    tr = oracle.tx_build_trace(w)
    one = oracle.to_mont([1])[0]
    for t in range(w.n_tx):
        base = 1024 * t
        row = 8 * w.depth + 7  # TRANSACTION_HASH_LENGTH: state after step 126
        # (7) roots: old sender path -> this tx's initial root; new sender == old receiver (intermediate);
        #     new receiver -> next initial root / final root
        assert (tr[0:7, base + row] == w.initial_roots[t]).all()
        assert (tr[15:22, base + row] == tr[29:36, base + row]).all()
        nxt = w.initial_roots[t + 1] if t + 1 < w.n_tx else w.final_root
        assert (tr[44:51, base + row] == nxt).all()
        assert (tr[58:65, base + row] == nxt).all()          # copied at step 126, src/merkle/update/trace.rs:87-93
        assert (tr[58:65, base] == w.initial_roots[t]).all()
        # (8) Schnorr: affine x of s*G + h*P equals R.x at the last row; limbs equal hash output at row 1022
        assert (tr[0:6, base + 1023] == w.sig_rx[t]).all()
        assert (tr[38:42, base + 1022] == tr[42:46, base + 1022]).all()
        # (9) range accumulators after 64 steps
        assert tr[57, base + 576] == tr[89, base + 576] == w.deltas[t]
        assert tr[93, base + 576] == tr[90, base + 576]
        # copies
        assert (tr[65:77, base + 1] == w.s_old_values[t, :12]).all()
        assert (tr[77:89, base + 5] == w.r_old_values[t, :12]).all()
        # Schnorr init row (src/schnorr/trace.rs:18-30)
        r512 = tr[:56, base + 512]
        expect = np.zeros(56, np.uint64); expect[6] = one; expect[25] = one; expect[42:48] = w.sig_rx[t]
        assert (r512 == expect).all()
    # public inputs as get_pub_inputs reads them (src/prover.rs:106-129)
    assert (tr[58:65, 0] == w.initial_roots[0]).all()
    assert (tr[58:65, -1] == w.final_root).all()


@pytest.mark.parametrize("field,idx", [("deltas", 0), ("s_paths", (1, 2, 0)), ("r_paths", (0, 1, 3)),
                                       ("s_old_values", (1, 12)), ("r_old_values", (0, 3)), ("initial_roots", (1, 0))])
def test_perturbed_witness_violates_some_constraint(oracle, witness_d3, field, idx):
    """Mirror of the reference's negative tests: any witness element changed -> an invalid trace."""
    w = witness_d3.copy()
    arr = getattr(w, field)
    if arr.dtype == np.uint8:
        arr[idx] ^= 1
    else:
        arr[idx] = oracle.fp_add(np.array([arr[idx]], np.uint64), oracle.to_mont([1]))[0]
    trace = oracle.tx_build_trace(w)
    assert oracle.tx_check_trace(trace, w.n_tx, w.depth) >= 0


@pytest.mark.parametrize("field,idx", [("sig_rx", (0, 0)), ("sig_s", (1, 5))])
def test_signature_is_only_bound_by_the_final_x_coordinate(oracle, witness_d3, field, idx):
    """Reference quirk, reproduced: TransactionAir has only the 4 root assertions (src/air.rs:175-184), so a
    wrong signature still satisfies every *transition* constraint; what it breaks is the row-1023 x == R.x
    relation that the standalone SchnorrAir asserts (src/schnorr/air.rs:217-224)."""
    w = witness_d3.copy()
    arr = getattr(w, field)
    if arr.dtype == np.uint8:
        arr[idx] ^= 1
    else:
        arr[idx] = oracle.fp_add(np.array([arr[idx]], np.uint64), oracle.to_mont([1]))[0]
    trace = oracle.tx_build_trace(w)
    assert oracle.tx_check_trace(trace, w.n_tx, w.depth) == -1
    t = idx[0]
    assert (trace[0:6, 1024 * t + 1023] != w.sig_rx[t]).any()


def test_swapped_index_bit_violates(oracle, witness_d3):
    w = witness_d3.copy()
    w.s_indices[0] ^= 1
    assert oracle.tx_check_trace(oracle.tx_build_trace(w), w.n_tx, w.depth) >= 0


def test_periodic_columns_shape(oracle):
    """48 columns (src/air.rs:194-380): 19 of period 1024, HASH_INPUT and the 28 ARK columns of period 8."""
    pc = oracle.tx_periodic_columns(15)
    one = oracle.to_mont([1])[0]
    assert pc.shape == (48, 1024)
    short = [c for c in range(48) if (pc[c].reshape(128, 8) == pc[c, :8]).all()]
    assert short == [2] + list(range(20, 48))
    assert pc[0, 0] == one and not pc[0, 1:].any()                      # SETUP
    assert (pc[1, :127] == one).all() and not pc[1, 127:].any()        # MERKLE: 127 ones
    assert pc[3, 126] == one and pc[3].astype(bool).sum() == 1         # FINISH one-hot at 126
    assert (pc[5, 512:1023] == one).all() and pc[5, 1023] == 0 and not pc[5, :512].any()   # SCHNORR: 511 ones
    assert (pc[6, 512:1022] == one).all() and not pc[6, 1022:].any()   # SCALAR_MULT: 510 ones
    for k in range(4):                                                  # message-chunk one-hots (:301-318)
        assert pc[13 + k, 512 + 8 * (k + 1) - 1] == one and pc[13 + k].astype(bool).sum() == 1
    assert (pc[17, 512:576] == one).all() and pc[17].astype(bool).sum() == 64
    assert pc[18, 575] == one and pc[18].astype(bool).sum() == 1
    assert pc[19, 0] == 0 and (pc[19, 1:576] == one).all() and not pc[19, 576:].any()      # VALUE_COPY
    assert not pc[20:48, 7].any()                                       # 8th ARK row is zero (rescue.rs:995)


def test_constraint_degree_multiset(oracle):
    """24 x (5;2 cycles), 12 x (4;2), 22 x (3;1), 1 x (2;1), 56 x (1;1)."""
    base, cyc = oracle.tx_constraint_degrees()
    from collections import Counter
    assert Counter(zip(base.tolist(), cyc.tolist())) == {(5, 2): 24, (4, 2): 12, (3, 1): 22, (2, 1): 1, (1, 1): 56}
    assert max(b + c for b, c in zip(base, cyc)) == 7  # -> constraint-evaluation blowup 8


def test_witness_is_deterministic_and_signatures_are_valid(oracle):
    a = oracle.TxWitness.generate(4, 3, seed=11)
    b = oracle.TxWitness.generate(4, 3, seed=11)
    c = oracle.TxWitness.generate(4, 3, seed=12)
    for f in a.FIELDS:
        assert (getattr(a, f) == getattr(b, f)).all()
    assert any((getattr(a, f) != getattr(c, f)).any() for f in a.FIELDS)
    assert (a.s_indices != a.r_indices).all()
    assert (a.sig_s[:, 31] < 128).all()          # s < 2^255
    for t in range(a.n_tx):                       # public keys are curve points
        assert oracle.lib().cso_ecc_on_curve_affine(oracle._p(np.ascontiguousarray(a.s_old_values[t, :12]))) == 1
