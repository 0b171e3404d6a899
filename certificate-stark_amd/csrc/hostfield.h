// Host-side f63 arithmetic for the small amount of scalar work the C ABI does on the CPU (roots of unity,
// coset offsets, table parameters).  Same representation as the device code: Montgomery form, R = 2^64.
#pragma once
#include <stdint.h>

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
// multiplicative generator 3 and the 2^55-th root of unity 3^131 (engine conventions [UPSTREAM-RECALL])
inline uint64_t generator() { return from_u64(3); }
inline uint64_t root_of_unity(unsigned log_n) {
    uint64_t g = pow(from_u64(3), 131);
    for (unsigned i = log_n; i < 55; i++) g = mul(g, g);
    return g;
}

// Extension fields on the host (FieldExtension::Quadratic / Cubic; ext.hip has the assumed polynomials):
// m = 2: F_p[u]/(u^2 - 2u - 2), m = 3: F_p[v]/(v^3 + v + 1); unused high coefficients stay zero
struct EX { uint64_t c[3]; };
inline EX ex_zero() { return {{0, 0, 0}}; }
inline EX ex_one() { return {{ONE, 0, 0}}; }
inline EX ex_load(const uint64_t *p, unsigned m) { EX r = ex_zero(); for (unsigned i = 0; i < m; i++) r.c[i] = p[i]; return r; }
inline EX ex_add(EX x, EX y) { return {{add(x.c[0], y.c[0]), add(x.c[1], y.c[1]), add(x.c[2], y.c[2])}}; }
inline EX ex_scale(EX x, uint64_t s) { return {{mul(x.c[0], s), mul(x.c[1], s), mul(x.c[2], s)}}; }
inline EX ex_mul(EX x, EX y, unsigned m) {
    if (m == 2) {
        const uint64_t bd = mul(x.c[1], y.c[1]), bd2 = add(bd, bd);
        return {{add(mul(x.c[0], y.c[0]), bd2), add(add(mul(x.c[0], y.c[1]), mul(x.c[1], y.c[0])), bd2), 0}};
    }
    const uint64_t d0 = mul(x.c[0], y.c[0]), d1 = add(mul(x.c[0], y.c[1]), mul(x.c[1], y.c[0]));
    const uint64_t d2 = add(add(mul(x.c[0], y.c[2]), mul(x.c[1], y.c[1])), mul(x.c[2], y.c[0]));
    const uint64_t d3 = add(mul(x.c[1], y.c[2]), mul(x.c[2], y.c[1])), d4 = mul(x.c[2], y.c[2]);
    return {{sub(d0, d3), sub(sub(d1, d3), d4), sub(d2, d4)}};
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
