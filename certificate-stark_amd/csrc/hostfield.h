// Host-side f63 arithmetic for the small amount of scalar work the C ABI does on the CPU (roots of unity,
// coset offsets, table parameters).  Same representation as the device code: Montgomery form, R = 2^64.
#pragma once
#include <stdint.h>
#include "../../include/cstark_conventions.h"

namespace cs {
namespace host {

typedef unsigned __int128 u128;
constexpr uint64_t P = 0x4180000000000001ULL;
constexpr uint64_t NPINV = 0x417fffffffffffffULL;
constexpr uint64_t ONE = 0x3b7ffffffffffffdULL;
constexpr uint64_t R2 = 0x32734c36b7b1d512ULL;

inline uint64_t mul(uint64_t a, uint64_t b) {
    u128 t = (u128)a * b;
    uint64_t m = (uint64_t)t * NPINV;
    uint64_t r = (uint64_t)((t + (u128)m * P) >> 64);
    return r >= P ? r - P : r;
}
inline uint64_t add(uint64_t a, uint64_t b) { uint64_t s = a + b; return s >= P ? s - P : s; }
inline uint64_t sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (P - b); }
inline uint64_t from_u64(uint64_t x) { return mul(x % P, R2); }
inline uint64_t to_u64(uint64_t a) { return mul(a, 1); }
inline uint64_t pow(uint64_t b, uint64_t e) {
    uint64_t r = ONE;
    while (e) { if (e & 1) r = mul(r, b); b = mul(b, b); e >>= 1; }
    return r;
}
inline uint64_t inv(uint64_t a) { return pow(a, P - 2); }
// multiplicative generator, the 2^55-th root of unity and the offset of the evaluation domains: engine conventions
// [UPSTREAM-RECALL], include/cstark_conventions.h
inline uint64_t generator() { return from_u64(CSTARK_CONV_FIELD_GENERATOR); }
inline uint64_t lde_offset() { return from_u64(CSTARK_CONV_LDE_OFFSET); }
inline uint64_t root_of_unity(unsigned log_n) {
    uint64_t g = pow(generator(), CSTARK_CONV_TWO_ADIC_ROOT_EXP);
    for (unsigned i = log_n; i < 55; i++) g = mul(g, g);
    return g;
}
// x times a small signed integer (the reduction coefficients of the extension polynomials)
inline uint64_t mul_small(uint64_t x, int c) {
    uint64_t r = 0;
    for (int i = 0; i < (c < 0 ? -c : c); i++) r = add(r, x);
    return c < 0 ? sub(0, r) : r;
}

// Extension fields on the host (FieldExtension::Quadratic / Cubic); the assumed polynomials are CSTARK_CONV_E2_* / E3_* of
// include/cstark_conventions.h: u^2 = C1 u + C0, v^3 = C2 v^2 + C1 v + C0; unused high coefficients stay zero
struct EX { uint64_t c[3]; };
inline EX ex_zero() { return {{0, 0, 0}}; }
inline EX ex_one() { return {{ONE, 0, 0}}; }
inline EX ex_load(const uint64_t *p, unsigned m) { EX r = ex_zero(); for (unsigned i = 0; i < m; i++) r.c[i] = p[i]; return r; }
inline EX ex_add(EX x, EX y) { return {{add(x.c[0], y.c[0]), add(x.c[1], y.c[1]), add(x.c[2], y.c[2])}}; }
inline EX ex_scale(EX x, uint64_t s) { return {{mul(x.c[0], s), mul(x.c[1], s), mul(x.c[2], s)}}; }
inline EX ex_mul(EX x, EX y, unsigned m) {
    if (m == 2) {
        const uint64_t bd = mul(x.c[1], y.c[1]);
        return {{add(mul(x.c[0], y.c[0]), mul_small(bd, CSTARK_CONV_E2_C0)),
                 add(add(mul(x.c[0], y.c[1]), mul(x.c[1], y.c[0])), mul_small(bd, CSTARK_CONV_E2_C1)), 0}};
    }
    const uint64_t d0 = mul(x.c[0], y.c[0]), d1 = add(mul(x.c[0], y.c[1]), mul(x.c[1], y.c[0]));
    const uint64_t d2 = add(add(mul(x.c[0], y.c[2]), mul(x.c[1], y.c[1])), mul(x.c[2], y.c[0]));
    const uint64_t d3 = add(mul(x.c[1], y.c[2]), mul(x.c[2], y.c[1])), d4 = mul(x.c[2], y.c[2]);
    // v^3 = C2 v^2 + C1 v + C0;  v^4 = v v^3 = (C2^2 + C1) v^2 + (C2 C1 + C0) v + C2 C0
    constexpr int C0 = CSTARK_CONV_E3_C0, C1 = CSTARK_CONV_E3_C1, C2 = CSTARK_CONV_E3_C2;
    return {{add(add(d0, mul_small(d3, C0)), mul_small(d4, C2 * C0)),
             add(add(d1, mul_small(d3, C1)), mul_small(d4, C2 * C1 + C0)),
             add(add(d2, mul_small(d3, C2)), mul_small(d4, C2 * C2 + C1))}};
}
inline EX ex_pow(EX x, uint64_t e, unsigned m) {
    EX r = ex_one();
    while (e) { if (e & 1) r = ex_mul(r, x, m); x = ex_mul(x, x, m); e >>= 1; }
    return r;
}

// in-place inverse transform of a short sequence (m = 2^log_m values over the m-th roots of unity) -> coefficients
inline void intt_small(uint64_t *a, unsigned log_m) {
    const size_t m = (size_t)1 << log_m;
    for (size_t i = 1, j = 0; i < m; i++) {
        size_t bit = m >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    const uint64_t winv = inv(root_of_unity(log_m));
    for (unsigned s = 1; s <= log_m; s++) {
        const size_t len = (size_t)1 << s, half = len >> 1;
        uint64_t wl = winv;
        for (unsigned i = s; i < log_m; i++) wl = mul(wl, wl);
        for (size_t k = 0; k < m; k += len) {
            uint64_t tw = ONE;
            for (size_t j = 0; j < half; j++) {
                const uint64_t u = a[k + j], v = mul(a[k + j + half], tw);
                a[k + j] = add(u, v);
                a[k + j + half] = sub(u, v);
                tw = mul(tw, wl);
            }
        }
    }
    const uint64_t minv = inv(from_u64(m));
    for (size_t i = 0; i < m; i++) a[i] = mul(a[i], minv);
}

} // namespace host
} // namespace cs
