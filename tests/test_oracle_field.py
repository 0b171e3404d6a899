"""Pins the oracle's field / tower / gadget arithmetic against Python big-integer arithmetic and the
known answers of SURVEY.md 8(c) (1)-(5).  CPU only."""
import random

import numpy as np

P = 2**62 + 2**56 + 2**55 + 1
R = 2**64


def mont(x): return x * R % P
def unmont(x): return x * pow(R, -1, P) % P


def test_modulus_facts():
    assert P == 0x4180000000000001 == 4719772409484279809   # src/range/tests.rs:59
    assert (P - 1) % 2**55 == 0 and ((P - 1) >> 55) == 131  # two-adicity 55
    assert pow(3, (P - 1) // 2, P) != 1 and pow(3, (P - 1) // 131, P) != 1  # 3 generates F_p^*


def test_field_ops_vs_bigint(oracle):
    rng = random.Random(1)
    xs = [0, 1, 2, P - 1, P - 2, 2**62, 2**63 - 1 - P if 2**63 - 1 > P else 5] + [rng.randrange(P) for _ in range(500)]
    ys = [rng.randrange(P) for _ in xs]
    a, b = oracle.to_mont(xs), oracle.to_mont(ys)
    assert [int(v) for v in a] == [mont(x) for x in xs]
    assert [int(v) for v in oracle.from_mont(a)] == xs
    assert [int(v) for v in oracle.from_mont(oracle.fp_mul(a, b))] == [x * y % P for x, y in zip(xs, ys)]
    assert [int(v) for v in oracle.from_mont(oracle.fp_add(a, b))] == [(x + y) % P for x, y in zip(xs, ys)]
    assert [int(v) for v in oracle.from_mont(oracle.fp_sub(a, b))] == [(x - y) % P for x, y in zip(xs, ys)]
    inv = oracle.from_mont(oracle.fp_inv(a))
    assert [int(v) for v in inv] == [pow(x, P - 2, P) for x in xs]
    e = 3146514939656186539  # INV_ALPHA, src/utils/rescue.rs:383
    assert [int(v) for v in oracle.from_mont(oracle.fp_pow(a, e))] == [pow(x, e, P) for x in xs]
    # from_u64 reduces mod p (BaseElement::from(u64))
    big = np.array([2**64 - 1, P, P + 5], np.uint64)
    assert [int(v) for v in oracle.from_mont(oracle.to_mont(big))] == [(2**64 - 1) % P, 0, 5]


def test_roots_of_unity(oracle):
    for k in (1, 3, 10, 20, 23, 55):
        w = unmont(oracle.root_of_unity(k))
        assert pow(w, 2**k, P) == 1 and pow(w, 2**(k - 1), P) == P - 1
    assert unmont(oracle.root_of_unity(55)) == pow(3, 131, P)
    assert pow(unmont(oracle.root_of_unity(23)), 8, P) == unmont(oracle.root_of_unity(20))


def test_sbox_exponents_inverse():
    assert 3 * 3146514939656186539 % (P - 1) == 1


def _fp6_mul_py(a, b):
    """Schoolbook product in F_p[u,v]/(u^2-2u-2, v^3+v+1) -- independent of the Karatsuba form in ecc.rs."""
    def mul2(x, y):  # (x0 + x1 u)(y0 + y1 u), u^2 = 2u + 2
        c0 = x[0] * y[0] + 2 * x[1] * y[1]
        c1 = x[0] * y[1] + x[1] * y[0] + 2 * x[1] * y[1]
        return (c0 % P, c1 % P)
    def add2(x, y): return ((x[0] + y[0]) % P, (x[1] + y[1]) % P)
    def neg2(x): return ((-x[0]) % P, (-x[1]) % P)
    A = [(a[0], a[1]), (a[2], a[3]), (a[4], a[5])]
    B = [(b[0], b[1]), (b[2], b[3]), (b[4], b[5])]
    c = [(0, 0)] * 5
    for i in range(3):
        for j in range(3):
            c[i + j] = add2(c[i + j], mul2(A[i], B[j]))
    # v^3 = -v - 1, v^4 = -v^2 - v
    c[1] = add2(c[1], neg2(c[3])); c[0] = add2(c[0], neg2(c[3]))
    c[2] = add2(c[2], neg2(c[4])); c[1] = add2(c[1], neg2(c[4]))
    return [c[0][0], c[0][1], c[1][0], c[1][1], c[2][0], c[2][1]]


def test_fp6_vs_schoolbook(oracle):
    import ctypes as C
    rng = random.Random(2)
    L = oracle.lib()
    for _ in range(50):
        a = [rng.randrange(P) for _ in range(6)]
        b = [rng.randrange(P) for _ in range(6)]
        am, bm = oracle.to_mont(a), oracle.to_mont(b)
        out = np.zeros(6, np.uint64)
        L.cso_fp6_mul(oracle._p(am), oracle._p(bm), oracle._p(out))
        assert [int(v) for v in oracle.from_mont(out)] == _fp6_mul_py(a, b)
        L.cso_fp6_sqr(oracle._p(am), oracle._p(out))
        assert [int(v) for v in oracle.from_mont(out)] == _fp6_mul_py(a, a)
        inv = np.zeros(6, np.uint64)
        L.cso_fp6_inv(oracle._p(am), oracle._p(inv))
        L.cso_fp6_mul(oracle._p(am), oracle._p(inv), oracle._p(out))
        assert [int(v) for v in oracle.from_mont(out)] == [1, 0, 0, 0, 0, 0]  # invert_fp6(x) * x = 1


GX = [2398517019392108645, 4508025770867562887, 3052857668015466949, 1056103921720638754, 2633256936270674947, 288076929228681448]
GY = [3894155704139868264, 1225290585625954719, 1961556908722893436, 3024200307602630234, 4227116334258416103, 3289504647774244396]
GEN_RAW = [0xf6798582c92ece1, 0x2b7c30a4c7d886c0, 0x1269cdae98dc2fd0, 0x11b78ef6c71c6132, 0x3ac2244dfc47537, 0x36dfeea4b9051daf,
           0x334807e450d55e2f, 0x200a54d42b84bd17, 0x271af7bb20ab32e1, 0x3df7b90927efc7ec, 0xab8bbf4a53af6a0, 0xe13dca26b2ac6ab]


def test_generator_on_curve(oracle):
    """the de-Montgomerised GENERATOR (src/utils/ecc.rs:23-36) lies on y^2 = x^3 + x + B3/3."""
    g = np.array(GEN_RAW, np.uint64)
    assert [int(v) for v in oracle.from_mont(g)] == GX + GY
    assert oracle.lib().cso_ecc_on_curve_affine(oracle._p(g)) == 1
    bad = g.copy(); bad[0] ^= np.uint64(1)
    assert oracle.lib().cso_ecc_on_curve_affine(oracle._p(bad)) == 0


def test_group_law_consistency(oracle):
    """double(P) == add(P,P) == add_mixed(P,P) projectively; 5G two ways; results stay on the curve."""
    import ctypes as C
    L = oracle.lib()
    g = np.array(GEN_RAW, np.uint64)
    one = oracle.to_mont([1])[0]
    proj = np.zeros(18, np.uint64); proj[:12] = g; proj[12] = one
    d = proj.copy(); L.cso_ecc_double(oracle._p(d))
    a = proj.copy(); L.cso_ecc_add(oracle._p(a), oracle._p(proj))
    m = proj.copy(); L.cso_ecc_add_mixed(oracle._p(m), oracle._p(g))

    def affine(p):
        zi = np.zeros(6, np.uint64); L.cso_fp6_inv(oracle._p(p[12:18].copy()), oracle._p(zi))
        x = np.zeros(6, np.uint64); y = np.zeros(6, np.uint64)
        L.cso_fp6_mul(oracle._p(p[0:6].copy()), oracle._p(zi), oracle._p(x))
        L.cso_fp6_mul(oracle._p(p[6:12].copy()), oracle._p(zi), oracle._p(y))
        return np.concatenate([x, y])
    ad, aa, am = affine(d), affine(a), affine(m)
    assert (ad == aa).all() and (ad == am).all()
    assert L.cso_ecc_on_curve_affine(oracle._p(ad)) == 1
    # 5G = 2(2G)+G  vs scalar_mul(5)
    q = d.copy(); L.cso_ecc_double(oracle._p(q)); L.cso_ecc_add_mixed(oracle._p(q), oracle._p(g))
    k = np.array([5], np.uint64); out = np.zeros(12, np.uint64)
    L.cso_ecc_scalar_mul_affine(oracle._p(k), C.c_uint(1), oracle._p(g), oracle._p(out))
    assert (affine(q) == out).all()
    assert L.cso_ecc_on_curve_affine(oracle._p(out)) == 1


def test_rescue_round_then_enforce_is_zero(oracle):
    """apply_round followed by enforce_round on (before, after) gives 14 zeros, rounds 0-6."""
    import ctypes as C
    rng = random.Random(3)
    L = oracle.lib()
    pc = oracle.tx_periodic_columns(15)
    one = int(oracle.to_mont([1])[0])
    for step in range(7):
        before = oracle.to_mont([rng.randrange(P) for _ in range(14)])
        after = before.copy()
        L.cso_rescue_round(oracle._p(after), C.c_uint32(step))
        ark = np.ascontiguousarray(pc[20:48, step])
        res = np.zeros(14, np.uint64)
        L.cso_rescue_enforce_round(oracle._p(res), oracle._p(before), oracle._p(after), oracle._p(ark), C.c_uint64(one))
        assert not res.any()
        after[3] ^= np.uint64(1)
        res[:] = 0
        L.cso_rescue_enforce_round(oracle._p(res), oracle._p(before), oracle._p(after), oracle._p(ark), C.c_uint64(one))
        assert res.any()


def test_rescue_permutation_is_seven_rounds_and_merge(oracle):
    import ctypes as C
    rng = random.Random(4)
    L = oracle.lib()
    s = oracle.to_mont([rng.randrange(P) for _ in range(14)])
    a = s.copy(); L.cso_rescue_permutation(oracle._p(a))
    b = s.copy()
    for i in range(7):
        L.cso_rescue_round(oracle._p(b), C.c_uint32(i))
    assert (a == b).all()
    out = np.zeros(7, np.uint64)
    L.cso_rescue_merge(oracle._p(s[:7].copy()), oracle._p(s[7:].copy()), oracle._p(out))
    assert (out == a[:7]).all()
    # digest absorbs 7 at a time without padding (src/utils/rescue.rs:108-130)
    d = np.zeros(7, np.uint64)
    L.cso_rescue_digest(oracle._p(s[:7].copy()), C.c_size_t(7), oracle._p(d))
    t = np.zeros(14, np.uint64); t[:7] = s[:7]; L.cso_rescue_permutation(oracle._p(t))
    assert (d == t[:7]).all()
