/* ORACLE (test infrastructure, not product code).
 *
 * CPU restatement of the base field used by the reference: winterfell-fork `fields::f63::BaseElement`
 * (un-vendored dependency, Cargo.toml:20 @8e37310).  p = 2^62 + 2^56 + 2^55 + 1 (the modulus is
 * spelled out in src/range/tests.rs:59 and benches/range.rs:23).  Elements are held in Montgomery
 * form with R = 2^64, fully reduced to [0,p): this is the in-memory representation the reference
 * relies on (GENERATOR is given as raw words via from_raw_unchecked, src/utils/ecc.rs:23-36, while
 * BaseElement::new() takes canonical integers, src/utils/ecc.rs:38-45).
 *
 * Pinning: the arithmetic here is checked against Python big-integer arithmetic in
 * tests/test_oracle_field.py; the engine-level conventions (generator, roots of unity) are
 * [UPSTREAM-RECALL] and flagged "parity unpinned" in DESIGN.md.
 */
#ifndef CS_ORACLE_FP_H
#define CS_ORACLE_FP_H
#include <stdint.h>
#include <stddef.h>

typedef uint64_t fp;
typedef unsigned __int128 u128;

#define FP_P     0x4180000000000001ULL
#define FP_NPINV 0x417fffffffffffffULL /* -p^-1 mod 2^64 */
#define FP_ONE   0x3b7ffffffffffffdULL /* 2^64 mod p   = mont(1) */
#define FP_R2    0x32734c36b7b1d512ULL /* 2^128 mod p */
#define FP_ZERO  0ULL
#define FP_TWO_ADICITY 55
#include "../include/cstark_conventions.h"   /* the engine conventions [UPSTREAM-RECALL], shared with the product */
#define FP_GENERATOR_CAN ((uint64_t)CSTARK_CONV_FIELD_GENERATOR)   /* multiplicative generator = offset of the evaluation domains */
#define FP_LDE_OFFSET_CAN ((uint64_t)CSTARK_CONV_LDE_OFFSET)

static inline fp fp_add(fp a, fp b) { uint64_t s = a + b; return s >= FP_P ? s - FP_P : s; }
static inline fp fp_sub(fp a, fp b) { return a >= b ? a - b : a + (FP_P - b); }
static inline fp fp_neg(fp a) { return a ? FP_P - a : 0; }
static inline fp fp_dbl(fp a) { return fp_add(a, a); }

/* Montgomery product a*b*2^-64 mod p */
static inline fp fp_mul(fp a, fp b) {
    u128 t = (u128)a * b;
    uint64_t m = (uint64_t)t * FP_NPINV;
    u128 u = t + (u128)m * FP_P;
    uint64_t r = (uint64_t)(u >> 64);
    return r >= FP_P ? r - FP_P : r;
}
static inline fp fp_sqr(fp a) { return fp_mul(a, a); }

/* canonical integer -> element (BaseElement::new / From<u64>: reduces mod p) */
static inline fp fp_from_u64(uint64_t x) { return fp_mul(x % FP_P, FP_R2); }
/* element -> canonical integer (to_repr / to_bytes little-endian, src/lib.rs:355, src/prover.rs:54) */
static inline uint64_t fp_to_u64(fp a) { return fp_mul(a, 1); }

static inline fp fp_pow(fp base, uint64_t e) {
    fp r = FP_ONE;
    while (e) { if (e & 1) r = fp_mul(r, base); base = fp_sqr(base); e >>= 1; }
    return r;
}
static inline fp fp_inv(fp a) { return fp_pow(a, FP_P - 2); } /* inv(0) = 0 */
/* x times a small signed integer (reduction coefficients of the extension polynomials) */
static inline fp fp_mul_small(fp x, int c) {
    fp r = 0;
    for (int i = 0; i < (c < 0 ? -c : c); i++) r = fp_add(r, x);
    return c < 0 ? fp_neg(r) : r;
}

/* primitive 2^k-th root of unity: G^(2^(55-k)), G = generator^131   [UPSTREAM-RECALL get_root_of_unity] */
static inline fp fp_root_of_unity(unsigned log_n) {
    fp g = fp_pow(fp_from_u64(FP_GENERATOR_CAN), CSTARK_CONV_TWO_ADIC_ROOT_EXP);
    for (unsigned i = log_n; i < FP_TWO_ADICITY; i++) g = fp_sqr(g);
    return g;
}
#endif
