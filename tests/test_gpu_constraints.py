"""GPU parity: K6 constraint evaluation (all 115 transition constraints, and the fused combined evaluations with
boundary terms) vs the CPU oracle, bit-exact, on the LDE of oracle-built traces."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 2**62 + 2**56 + 2**55 + 1


@pytest.fixture(scope="module")
def backend():
    from certificate_stark_amd.backend import Backend
    b = Backend()
    yield b
    b.close()


def _lde(oracle, w, log_b=3):
    trace = oracle.tx_build_trace(w)
    return oracle.lde_columns(oracle.interpolate_columns(trace), log_b)


@pytest.mark.parametrize("n_tx,depth", [(1, 3), (2, 15)])
def test_all_transition_constraints_match_oracle(oracle, backend, n_tx, depth):
    from certificate_stark_amd.backend import to_numpy_u64
    w = oracle.TxWitness.generate(n_tx, depth, seed=31 + n_tx)
    lde = _lde(oracle, w)
    ref = oracle.tx_evaluate_transitions(lde, depth, 3)
    got = to_numpy_u64(backend.evaluate_transitions(backend.from_numpy_u64(lde), depth))
    if not (got == ref).all():
        bad = sorted(set(np.argwhere(got != ref)[:, 1].tolist()))
        raise AssertionError("constraints differ: %s" % bad)
    # subset of cosets
    part = to_numpy_u64(backend.evaluate_transitions(backend.from_numpy_u64(lde[5:7]), depth, k0=5))
    assert (part == ref[5:7]).all()


@pytest.mark.parametrize("n_tx,depth", [(1, 3), (2, 15), (4, 7)])
def test_fused_combined_evaluations_match_oracle(oracle, backend, n_tx, depth):
    from certificate_stark_amd.backend import to_numpy_u64
    w = oracle.TxWitness.generate(n_tx, depth, seed=41 + n_tx)
    lde = _lde(oracle, w)
    cf = oracle.make_coeffs(17)
    pub = np.concatenate([w.initial_roots[0][:2], w.final_root[:2]])
    ref = oracle.tx_evaluate_constraints(lde, cf, pub, depth, 3)
    d_lde = backend.from_numpy_u64(lde)
    got = to_numpy_u64(backend.evaluate_constraints(d_lde, cf, pub, depth))
    assert (got == ref).all()
    part = to_numpy_u64(backend.evaluate_constraints(d_lde[2:3].contiguous(), cf, pub, depth, k0=2))
    assert (part == ref[2:3]).all()
    # an invalid trace (perturbed LDE cell) still matches the oracle bit for bit: parity is not validity
    lde2 = lde.copy(); lde2[3, 17, 5] ^= np.uint64(1)
    ref2 = oracle.tx_evaluate_constraints(lde2, cf, pub, depth, 3)
    got2 = to_numpy_u64(backend.evaluate_constraints(backend.from_numpy_u64(lde2), cf, pub, depth))
    assert (got2 == ref2).all()


@pytest.mark.parametrize("m", [1, 2, 3])
def test_merged_evaluations_for_several_coefficient_sets(oracle, backend, m):
    """cstark_tx_evaluate_constraints_ext: the components of an extension-field proof.  Each coefficient set must give exactly
    what a single-set evaluation (and the oracle) gives for it; also on a sub-range of cosets."""
    from certificate_stark_amd.backend import to_numpy_u64
    n_tx, depth = 2, 15
    w = oracle.TxWitness.generate(n_tx, depth, seed=77)
    lde = _lde(oracle, w)
    sets = [oracle.make_coeffs(100 + q) for q in range(m)]
    pub = np.concatenate([w.initial_roots[0][:2], w.final_root[:2]])
    d_lde = backend.from_numpy_u64(lde)
    got = to_numpy_u64(backend.evaluate_constraints_ext(d_lde, sets, pub, depth))
    assert got.shape == (m, 8, lde.shape[2])
    for q in range(m):
        assert (got[q] == oracle.tx_evaluate_constraints(lde, sets[q], pub, depth, 3)).all(), "coefficient set %d" % q
    part = to_numpy_u64(backend.evaluate_constraints_ext(d_lde[4:6].contiguous(), sets, pub, depth, k0=4))
    assert (part == got[:, 4:6]).all()


def test_coefficient_set_count_is_checked(backend):
    from certificate_stark_amd._lib import CstarkError
    import torch
    lde = backend.empty_u64(1, 94, 1024)
    lde.zero_()
    from oracle import oracle as O
    with pytest.raises(CstarkError):
        backend.evaluate_constraints_ext(lde, [O.make_coeffs(1)] * 4, [0, 0, 0, 0], 3)


def test_misaligned_lde_is_refused(backend):
    """The Rescue-window kernel moves 16-byte pieces by LDS-DMA: an LDE base that is only 8-byte aligned is an argument error."""
    import ctypes as C
    from certificate_stark_amd._lib import CstarkError
    from oracle import oracle as O
    buf = backend.empty_u64(94 * 1024 + 1)
    buf.zero_()
    lde = buf[1:].view(1, 94, 1024)
    assert lde.data_ptr() % 16 == 8
    with pytest.raises(CstarkError):
        backend.evaluate_constraints(lde, O.make_coeffs(1), [0, 0, 0, 0], 3)


@pytest.mark.parametrize("n_tx,depth", [(1, 3), (4, 15)])
def test_degree_split_evaluation_matches_on_a_genuine_extension(oracle, backend, n_tx, depth):
    """cstark_tx_evaluate_constraints_lde: on the extension of a real trace the degree-split evaluation (even cosets, extension of
    the merged polynomials, recombination) gives the oracle's values bit for bit -- also for a trace that violates the constraints
    (a perturbed TRACE cell: still columns of degree < n, which is all the split relies on)."""
    from certificate_stark_amd.backend import to_numpy_u64
    w = oracle.TxWitness.generate(n_tx, depth, seed=91 + n_tx)
    trace = oracle.tx_build_trace(w)
    cf = oracle.make_coeffs(23)
    pub = np.concatenate([w.initial_roots[0][:2], w.final_root[:2]])
    for perturb in (False, True):
        t = trace.copy()
        if perturb:
            t[17, 5] ^= np.uint64(1)
            t[70, 900] ^= np.uint64(3)
        lde = oracle.lde_columns(oracle.interpolate_columns(t), 3)
        ref = oracle.tx_evaluate_constraints(lde, cf, pub, depth, 3)
        got = to_numpy_u64(backend.evaluate_constraints(backend.from_numpy_u64(lde), cf, pub, depth, input_is_lde=True))
        assert (got == ref).all(), "perturbed" if perturb else "valid"
