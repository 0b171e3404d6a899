"""GPU parity: K1 trace generation through the C ABI vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def backend():
    from certificate_stark_amd.backend import Backend
    b = Backend()
    yield b
    b.close()


def _compare(oracle, backend, w):
    from certificate_stark_amd.backend import to_numpy_u64
    expect = oracle.tx_build_trace(w)
    backend.upload_witness(w)
    got = to_numpy_u64(backend.build_trace())
    backend.synchronize()
    if not (got == expect).all():
        bad = np.argwhere(got != expect)
        cols = sorted(set(bad[:, 0].tolist()))
        first = bad[0]
        raise AssertionError("trace mismatch: %d cells, columns %s, first at (col %d, row %d): got %x want %x" % (
            len(bad), cols[:20], first[0], first[1], got[first[0], first[1]], expect[first[0], first[1]]))
    return got


@pytest.mark.parametrize("n_tx,depth", [(1, 3), (2, 3), (2, 15), (4, 7)])
def test_trace_matches_oracle(oracle, backend, n_tx, depth):
    w = oracle.TxWitness.generate(n_tx, depth, seed=100 + n_tx + depth)
    _compare(oracle, backend, w)


def test_trace_16_tx_depth15_and_constraints(oracle, backend):
    w = oracle.TxWitness.generate(16, 15, seed=0x5EED)
    got = _compare(oracle, backend, w)
    assert oracle.tx_check_trace(got, 16, 15) == -1
