"""GPU parity for the device field / tower primitives (fp.cuh, tower.cuh) against the oracle, bit-exact."""
import ctypes as C
import random

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


def _run(backend, name, a, b, n, op, out_len):
    from certificate_stark_amd.backend import to_numpy_u64
    da = backend.from_numpy_u64(a)
    db = backend.from_numpy_u64(b) if b is not None else None
    out = backend.empty_u64(out_len)
    from certificate_stark_amd import _lib
    fn = getattr(_lib.load_debug(), name)
    rc = fn(C.c_void_p(backend.stream.cuda_stream), backend._ptr(da), backend._ptr(db) if db is not None else None,
            backend._ptr(out), C.c_size_t(n), C.c_int(op))
    assert rc == 0
    backend.synchronize()
    return to_numpy_u64(out)


def test_fp_ops(oracle, backend):
    rng = random.Random(9)
    edge = [0, 1, 2, P - 1, P - 2, 2**32, 2**32 - 1, 2**62, (P - 1) // 2]
    xs = np.array(edge + [rng.randrange(P) for _ in range(4096 - len(edge))], np.uint64)
    ys = np.array(list(reversed(edge)) + [rng.randrange(P) for _ in range(4096 - len(edge))], np.uint64)
    a, b = oracle.to_mont(xs), oracle.to_mont(ys)
    n = len(xs)
    assert (_run(backend, "cstark_debug_fp_op", a, b, n, 0, n) == oracle.fp_mul(a, b)).all()
    assert (_run(backend, "cstark_debug_fp_op", a, b, n, 1, n) == oracle.fp_add(a, b)).all()
    assert (_run(backend, "cstark_debug_fp_op", a, b, n, 2, n) == oracle.fp_sub(a, b)).all()
    assert (_run(backend, "cstark_debug_fp_op", a, None, n, 3, n) == oracle.fp_inv(a)).all()
    assert (_run(backend, "cstark_debug_fp_op", a, None, n, 4, n) == oracle.fp_pow(a, 3146514939656186539)).all()
    raw = np.array([2**64 - 1, P, P + 1, 0, 5] + [rng.randrange(2**64) for _ in range(100)], np.uint64)
    assert (_run(backend, "cstark_debug_fp_op", raw, None, len(raw), 5, len(raw)) == oracle.to_mont(raw)).all()
    assert (_run(backend, "cstark_debug_fp_op", a, None, n, 6, n) == xs).all()
    assert (_run(backend, "cstark_debug_fp_op", a, None, n, 7, n) == oracle.fp_sub(np.zeros_like(a), a)).all()
    assert (_run(backend, "cstark_debug_fp_op", a, None, n, 8, n) == oracle.fp_add(a, a)).all()
    # unreduced first factor of a product (fp_sub_lazy, the transform butterflies): x - y + p up to 2p - 1, times every extreme y
    ext = [0, 1, P - 1, P - 2, 2**32 - 1, 2**32, 2**62, P - 2**32, (P - 1) // 2]
    lx = np.array([u for u in ext for _ in ext] + [rng.randrange(P) for _ in range(1000)], np.uint64)
    ly = np.array([v for _ in ext for v in ext] + [rng.randrange(P) for _ in range(1000)], np.uint64)
    for raw in (False, True):  # canonical values through the Montgomery map, and the same bit patterns taken as memory-form words
        la, lb = (lx, ly) if raw else (oracle.to_mont(lx), oracle.to_mont(ly))
        assert (_run(backend, "cstark_debug_fp_op", la, lb, len(la), 10, len(la)) == oracle.fp_mul(oracle.fp_sub(la, lb), lb)).all()
        assert (_run(backend, "cstark_debug_fp_op", la, lb, len(la), 0, len(la)) == oracle.fp_mul(la, lb)).all()


def test_wide_accumulator_reductions_at_their_bounds(backend):
    """acc_reduce_below_p (one conditional subtraction) is sound for a high word up to p - 2^32 -- not for every hi < p, as its
    comment once said -- and acc_reduce for hi < 2p.  Drives the boundary with the low words that maximise the unreduced result
    (low 32 bits of the first REDC step zero, its high part as large as it gets)."""
    rng = random.Random(77)
    P1, K, M32 = 0x41800000, 0x41800001, 2**32 - 1
    lows = [0, 1, 2**64 - 1, 2**32 - 1, 2**32, 2**63]
    for lo0 in (0, 1, 2, 0x80000000, M32, 12345):      # craft lo1 so that v = ~lo0 * P1 + lo1 + K has a zero low word
        lo1 = (-(((~lo0) & M32) * P1 + K)) & M32
        lows.append((lo1 << 32) | lo0)
    lows += [rng.randrange(2**64) for _ in range(500)]
    r_inv = pow(2**64, -1, P)

    def check(op, his):
        lo = np.array([l for l in lows for _ in his], np.uint64)
        hi = np.array([h for _ in lows for h in his], np.uint64)
        got = _run(backend, "cstark_debug_fp_op", lo, hi, len(lo), op, len(lo))
        want = np.array([((int(h) << 64) + int(l)) * r_inv % P for l, h in zip(lo, hi)], np.uint64)
        assert (got == want).all()
    below = [0, 1, P - 2**32, P - 2**32 - 1, (3 * P * P) >> 64, 2**62, 2**32] + [rng.randrange(P - 2**32 + 1) for _ in range(50)]
    check(11, below)
    check(12, below + [P - 1, P, P + 1, 2 * P - 1, 2 * P - 2**32] + [rng.randrange(2 * P) for _ in range(50)])


def test_fp6_ops(oracle, backend):
    rng = random.Random(10)
    n = 512
    a = oracle.to_mont(np.array([rng.randrange(P) for _ in range(6 * n)], np.uint64))
    b = oracle.to_mont(np.array([rng.randrange(P) for _ in range(6 * n)], np.uint64))
    # memory-form words at the ends of the range in every coefficient (carry bounds of the wide products and their 32-bit carry sums)
    hi = [P - 1, P - 2, P - 2**32, 2**62, 2**32 - 1, 0]
    for k in range(16):
        a[6 * k:6 * k + 6] = [hi[(k + i) % 3 if k < 8 else (k * i) % 6] for i in range(6)]
        b[6 * k:6 * k + 6] = [hi[(k // 2 + 2 * i) % 3 if k < 8 else (k + i) % 6] for i in range(6)]
    a[0:6] = P - 1
    b[0:6] = P - 1
    L = oracle.lib()
    exp_mul = np.zeros(6 * n, np.uint64); exp_sqr = np.zeros(6 * n, np.uint64); exp_inv = np.zeros(6 * n, np.uint64)
    for i in range(n):
        sl = slice(6 * i, 6 * i + 6)
        ai, bi = a[sl].copy(), b[sl].copy()
        o = np.zeros(6, np.uint64)
        L.cso_fp6_mul(oracle._p(ai), oracle._p(bi), oracle._p(o)); exp_mul[sl] = o
        L.cso_fp6_sqr(oracle._p(ai), oracle._p(o)); exp_sqr[sl] = o
        L.cso_fp6_inv(oracle._p(ai), oracle._p(o)); exp_inv[sl] = o
    assert (_run(backend, "cstark_debug_fp6_op", a, b, n, 0, 6 * n) == exp_mul).all()
    assert (_run(backend, "cstark_debug_fp6_op", a, None, n, 1, 6 * n) == exp_sqr).all()
    assert (_run(backend, "cstark_debug_fp6_op", a, None, n, 2, 6 * n) == exp_inv).all()


def test_matrix_core_mds_product_is_exact(backend):
    from certificate_stark_amd import _lib
    """mds_mfma.cuh: INV_MDS times a vector through the int8 matrix cores (byte-diagonal GEMM with signed digits and offsets)
    equals the 128-bit carry-propagating scalar code bit for bit, including the extreme operand patterns."""
    import ctypes as C
    from certificate_stark_amd.backend import to_numpy_u64
    npts = 1 << 14
    rng = np.random.default_rng(99)
    x = rng.integers(0, P, size=(14, npts), dtype=np.uint64)
    x[:, 0] = 0
    x[:, 1] = P - 1
    x[:, 2] = 0x0080808080808080
    x[:, 3] = 0x417fffffffffffff
    x[:, 4] = 0x007f7f7f7f7f7f7f
    x[:, 5] = np.arange(14, dtype=np.uint64)
    d_in = backend.from_numpy_u64(x)
    outs = []
    for use_mfma in (0, 1):
        d_out = backend.empty_u64(14, npts)
        ms = C.c_float()
        assert _lib.load_debug().cstark_debug_mds(C.c_void_p(backend.stream.cuda_stream), backend._ptr(d_in), backend._ptr(d_out), C.c_size_t(npts),
                                            use_mfma, 1, C.byref(ms)) == 0
        outs.append(to_numpy_u64(d_out))
    assert (outs[0] == outs[1]).all()


def test_wave_next_shift(backend):
    """fp.cuh wave_next: lane l receives lane l+1's value, the last lane of the wave its fallback operand."""
    n = 256
    a = np.arange(1000, 1000 + n, dtype=np.uint64) * np.uint64(0x100000001)
    b = np.arange(5000, 5000 + n, dtype=np.uint64) * np.uint64(0x300000007)
    got = _run(backend, "cstark_debug_fp_op", a, b, n, 9, n)
    exp = np.where(np.arange(n) % 64 == 63, b, np.roll(a, -1))
    assert (got == exp).all()
