"""GPU parity: K2 interpolation and K3 coset LDE through the C ABI vs the CPU oracle, bit-exact; plus the
size-independent round-trip property at the full 2^20 x 8 size."""
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


def _rand(oracle, shape, seed):
    rng = np.random.default_rng(seed)
    return oracle.to_mont(rng.integers(0, P, size=int(np.prod(shape)), dtype=np.uint64)).reshape(shape)


@pytest.mark.parametrize("log_n,width", [(6, 1), (7, 3), (10, 5), (11, 2), (13, 3), (14, 2), (16, 2), (18, 1)])
def test_interpolate_and_lde_match_oracle(oracle, backend, log_n, width):
    from certificate_stark_amd.backend import to_numpy_u64
    ev = _rand(oracle, (width, 1 << log_n), log_n)
    co_ref = oracle.interpolate_columns(ev)
    co = backend.interpolate_columns(backend.from_numpy_u64(ev))
    assert (to_numpy_u64(co) == co_ref).all()
    lde_ref = oracle.lde_columns(co_ref, 3)
    lde = backend.lde_columns(co, 3)
    assert (to_numpy_u64(lde) == lde_ref).all()
    part = backend.lde_columns(co, 3, k0=5, nk=2)
    assert (to_numpy_u64(part) == lde_ref[5:7]).all()


def test_custom_offset_and_blowup(oracle, backend):
    from certificate_stark_amd.backend import to_numpy_u64
    ev = _rand(oracle, (4, 1 << 10), 77)
    co_ref = oracle.interpolate_columns(ev)
    off = int(oracle.fp_pow(np.array([oracle.generator()], np.uint64), 8)[0])
    ref = oracle.lde_columns(co_ref, 2, offset=off)
    got = backend.lde_columns(backend.from_numpy_u64(co_ref), 2, offset=off)
    assert (to_numpy_u64(got) == ref).all()
    # offset 1, blowup 1: evaluating the interpolant on the trace domain returns the input
    one = int(oracle.to_mont([1])[0])
    back = backend.lde_columns(backend.from_numpy_u64(co_ref), 0, offset=one)
    assert (to_numpy_u64(back)[0] == ev).all()


def test_full_size_roundtrip_2_20(oracle, backend):
    """BASELINE size: 2^20-point columns.  interpolate -> evaluate on the trace domain (offset 1) is the identity,
    and coset 0 of the g-offset LDE agrees with the oracle on one column."""
    import torch
    from certificate_stark_amd.backend import to_numpy_u64
    log_n, width = 20, 3
    ev = _rand(oracle, (width, 1 << log_n), 2020)
    d_ev = backend.from_numpy_u64(ev)
    co = backend.interpolate_columns(d_ev.clone())
    one = int(oracle.to_mont([1])[0])
    back = backend.lde_columns(co, 0, offset=one)
    assert torch.equal(back[0], d_ev)
    lde0 = to_numpy_u64(backend.lde_columns(co, 3, k0=3, nk=1))
    co_ref = oracle.interpolate_columns(ev[:1])
    assert (to_numpy_u64(co[0]) == co_ref[0]).all()
    ref = oracle.lde_columns(co_ref, 3, k0=3, nk=1)
    assert (lde0[0, 0] == ref[0, 0]).all()
