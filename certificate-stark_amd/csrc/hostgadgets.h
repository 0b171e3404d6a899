// Host-side Rescue-Prime, F_p6 tower and Cheetah-curve arithmetic for the witness synthesis that the reference does on the
// CPU before proving (TransactionMetadata::build_random, /root/reference/src/lib.rs:235-465; schnorr::sign,
// src/schnorr/mod.rs:197-217).  Same algorithms as the device code (rescue.cuh, tower.cuh), plain C++ over hostfield.h.
#pragma once
#include <string.h>
#include "constants_gen.h"
#include "hostfield.h"

namespace cs { namespace hostg {
using namespace cs::host;
typedef uint64_t fp;

inline fp neg(fp a) { return a ? P - a : 0; }
inline fp dbl(fp a) { return add(a, a); }
inline fp sqr(fp a) { return mul(a, a); }

// ---- Rescue-Prime (src/utils/rescue.rs:236-263, :327-375) ------------------------------------------------------------
inline void matvec(const uint64_t *m, fp *s) {
    fp out[14];
    for (int i = 0; i < 14; i++) {
        fp acc = 0;
        for (int j = 0; j < 14; j++) acc = add(acc, mul(m[i * 14 + j], s[j]));
        out[i] = acc;
    }
    memcpy(s, out, sizeof out);
}
inline void permutation(fp *s) {
    for (int r = 0; r < 7; r++) {
        const uint64_t *ark = CS_ARK_MONT + r * 28;
        for (int i = 0; i < 14; i++) s[i] = mul(s[i], sqr(s[i]));
        matvec(CS_MDS_MONT, s);
        for (int i = 0; i < 14; i++) s[i] = pow(add(s[i], ark[i]), CS_INV_ALPHA);
        matvec(CS_MDS_MONT, s);
        for (int i = 0; i < 14; i++) s[i] = add(s[i], ark[14 + i]);
    }
}
inline void merge(const fp *a, const fp *b, fp *out) { // Rescue63::merge, rescue.rs:143-152
    fp s[14];
    memcpy(s, a, 56); memcpy(s + 7, b, 56);
    permutation(s);
    memcpy(out, s, 56);
}
inline void digest(const fp *data, size_t n, fp *out) { // Rescue63::digest without padding, rescue.rs:108-130
    fp s[14] = {0};
    size_t i = 0;
    for (size_t k = 0; k < n; k++) {
        s[i] = add(s[i], data[k]);
        if (++i % 7 == 0) { permutation(s); i = 0; }
    }
    if (i > 0) permutation(s);
    memcpy(out, s, 56);
}
inline void hash_message(const fp *rx6, const fp *msg28, fp *out7) { // src/schnorr/mod.rs:247-288
    fp h[7];
    digest(rx6, 6, h);
    for (int k = 0; k < 4; k++) merge(h, msg28 + 7 * k, h);
    memcpy(out7, h, 56);
}

// ---- F_p2 = F_p[u]/(u^2 - 2u - 2), F_p6 = F_p2[v]/(v^3 + v + 1)  (src/utils/ecc.rs:407-591) ------------------------------
struct F2 { fp a, b; };
struct F6 { F2 c[3]; };
inline F2 add2(F2 x, F2 y) { return {add(x.a, y.a), add(x.b, y.b)}; }
inline F2 sub2(F2 x, F2 y) { return {sub(x.a, y.a), sub(x.b, y.b)}; }
inline F2 neg2(F2 x) { return {neg(x.a), neg(x.b)}; }
inline F2 mul2(F2 x, F2 y) {
    const fp aa = mul(x.a, y.a), bb = mul(x.b, y.b), t = mul(sub(x.a, x.b), sub(y.b, y.a));
    const fp c0 = add(dbl(bb), aa);
    return {c0, add(add(bb, c0), t)};
}
inline F2 inv2(F2 x) {
    const fp t = inv(sub(add(sqr(x.a), mul(dbl(x.a), x.b)), dbl(sqr(x.b))));
    return {mul(add(x.a, dbl(x.b)), t), mul(neg(x.b), t)};
}
inline F6 add6(const F6 &x, const F6 &y) { return {{add2(x.c[0], y.c[0]), add2(x.c[1], y.c[1]), add2(x.c[2], y.c[2])}}; }
inline F6 sub6(const F6 &x, const F6 &y) { return {{sub2(x.c[0], y.c[0]), sub2(x.c[1], y.c[1]), sub2(x.c[2], y.c[2])}}; }
inline F6 dbl6(const F6 &x) { return add6(x, x); }
inline F6 mul6(const F6 &x, const F6 &y) {
    const F2 aa = mul2(x.c[0], y.c[0]), bb = mul2(x.c[1], y.c[1]), cc = mul2(x.c[2], y.c[2]);
    const F2 ab = mul2(add2(x.c[0], x.c[1]), add2(y.c[0], y.c[1])), ac = mul2(add2(x.c[0], x.c[2]), add2(y.c[0], y.c[2]));
    const F2 bc = mul2(add2(x.c[1], x.c[2]), add2(y.c[1], y.c[2]));
    const F2 tmp = add2(add2(aa, bb), cc);
    return {{sub2(tmp, bc), sub2(sub2(ab, bc), aa), add2(sub2(sub2(ac, tmp), cc), add2(bb, bb))}};
}
inline F6 inv6(const F6 &x) {
    const F2 c0 = x.c[0], c1 = x.c[1], c2 = x.c[2];
    const F2 s0 = mul2(c0, c0), s1 = mul2(c1, c1), s2 = mul2(c2, c2);
    F2 t = mul2(c0, add2(s0, s1));
    t = sub2(t, mul2(c1, s1));
    t = add2(t, mul2(add2(c0, sub2(c2, c1)), s2));
    F2 w = mul2(add2(add2(c0, c0), c0), c1);
    w = mul2(sub2(add2(s0, s0), w), c2);
    t = inv2(sub2(t, w));
    const F2 r0 = mul2(sub2(add2(add2(s0, s1), s2), mul2(sub2(add2(c0, c0), c1), c2)), t);
    const F2 r1 = mul2(neg2(add2(mul2(c0, c1), s2)), t);
    const F2 r2 = mul2(add2(sub2(s1, mul2(c0, c2)), s2), t);
    return {{r0, r1, r2}};
}
inline F6 load6(const fp *p) { return {{{p[0], p[1]}, {p[2], p[3]}, {p[4], p[5]}}}; }
inline void store6(fp *p, const F6 &x) { for (int i = 0; i < 3; i++) { p[2 * i] = x.c[i].a; p[2 * i + 1] = x.c[i].b; } }

// ---- complete projective formulas on y^2 = x^3 + x + B (src/utils/ecc.rs:186-404) ------------------------------------------
struct Pt { F6 x, y, z; };
inline Pt pt_double(const Pt &p) {
    const F6 b3 = load6(CS_B3_MONT);
    F6 t0 = mul6(p.x, p.x), t1 = mul6(p.y, p.y), t2 = mul6(p.z, p.z);
    F6 t3 = dbl6(mul6(p.x, p.y)), z3 = dbl6(mul6(p.x, p.z));
    F6 y3 = add6(z3, mul6(b3, t2)), x3 = sub6(t1, y3);
    y3 = add6(t1, y3);
    y3 = mul6(x3, y3);
    x3 = mul6(t3, x3);
    z3 = mul6(b3, z3);
    t3 = add6(sub6(t0, t2), z3);
    t0 = add6(add6(dbl6(t0), t0), t2);
    y3 = add6(y3, mul6(t0, t3));
    t2 = dbl6(mul6(p.y, p.z));
    x3 = sub6(x3, mul6(t2, t3));
    z3 = dbl6(dbl6(mul6(t2, t1)));
    return {x3, y3, z3};
}
inline Pt pt_add_mixed(const Pt &p, const F6 &qx, const F6 &qy) {
    const F6 b3 = load6(CS_B3_MONT);
    F6 t0 = mul6(p.x, qx), t1 = mul6(p.y, qy);
    F6 t3 = sub6(mul6(add6(qx, qy), add6(p.x, p.y)), add6(t0, t1));
    F6 t4 = add6(mul6(qx, p.z), p.x), t5 = add6(mul6(qy, p.z), p.y);
    F6 z3 = add6(mul6(p.z, b3), t4), x3 = sub6(t1, z3);
    z3 = add6(t1, z3);
    F6 y3 = mul6(x3, z3);
    t1 = add6(add6(dbl6(t0), t0), p.z);
    t4 = add6(mul6(t4, b3), sub6(t0, p.z));
    y3 = add6(y3, mul6(t1, t4));
    x3 = sub6(mul6(t3, x3), mul6(t5, t4));
    z3 = add6(mul6(t5, z3), mul6(t3, t1));
    return {x3, y3, z3};
}
// k * base (affine [12]) -> affine [12]; k little-endian 64-bit limbs, MSB-first double-and-add
inline void scalar_mul_affine(const uint64_t *k, unsigned n_limbs, const fp *base12, fp *out12) {
    Pt acc{};
    acc.y.c[0].a = ONE; // identity (0 : 1 : 0)
    const F6 bx = load6(base12), by = load6(base12 + 6);
    for (int i = (int)n_limbs * 64 - 1; i >= 0; i--) {
        acc = pt_double(acc);
        if ((k[i / 64] >> (i % 64)) & 1) acc = pt_add_mixed(acc, bx, by);
    }
    const F6 zi = inv6(acc.z);
    store6(out12, mul6(acc.x, zi));
    store6(out12 + 6, mul6(acc.y, zi));
}

}} // namespace cs::hostg
