// F_p2 = F_p[u]/(u^2 - 2u - 2) and F_p6 = F_p2[v]/(v^3 + v + 1): the tower the reference uses for
// Cheetah-curve coordinates (/root/reference/src/utils/ecc.rs:407-648).  Exact field arithmetic, so
// any evaluation order gives the reference's values; the forms below are chosen for register
// pressure on CDNA4 (everything stays in VGPRs, no arrays indexed at run time).
#pragma once
#include "fp.cuh"

namespace cs {

struct Fp2 { fp a, b; };           // a + b*u
struct Fp6 { fp c[6]; };           // (c0 + c1 u) + (c2 + c3 u) v + (c4 + c5 u) v^2

__device__ __forceinline__ Fp2 fp2_add(Fp2 x, Fp2 y) { return {fp_add(x.a, y.a), fp_add(x.b, y.b)}; }
__device__ __forceinline__ Fp2 fp2_sub(Fp2 x, Fp2 y) { return {fp_sub(x.a, y.a), fp_sub(x.b, y.b)}; }
__device__ __forceinline__ Fp2 fp2_dbl(Fp2 x) { return {fp_dbl(x.a), fp_dbl(x.b)}; }
__device__ __forceinline__ Fp2 fp2_neg(Fp2 x) { return {fp_neg(x.a), fp_neg(x.b)}; }

// (a0 + a1 u)(b0 + b1 u) with u^2 = 2u + 2:  c0 = a0 b0 + 2 a1 b1,  c1 = a0 b1 + a1 b0 + 2 a1 b1.
// Karatsuba on UNREDUCED 128-bit products: V0 = a0 b0, V1 = a1 b1, V2 = (a0 + a1)(b0 + b1) (the operand sums stay below 2p
// and need no reduction), C0 = V0 + 2 V1 < 3 p^2, C1 = V2 - V0 + V1 in [0, 5 p^2): two Montgomery reductions instead of three,
// and the linear steps are plain 128-bit additions.  Same field values as the reference's form (ecc.rs:424-439).
#ifndef CS_FP2_EAGER
typedef unsigned __int128 u128_t;
// full 128-bit product, four v_mad_u64_u32.  REDUCED: both operands < p, so the two middle carries fit a 32-bit sum; otherwise
// (operand sums < 2p) the second one enters through a multiply by one (fp.cuh).
template <bool REDUCED>
__device__ __forceinline__ u128_t mul_wide(uint64_t a, uint64_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t t0 = (uint64_t)a0 * b0;
    const uint64_t t1 = mad_u64_u32(a1, b0, t0 >> 32);
    const uint64_t t2 = mad_u64_u32(a0, b1, (uint32_t)t1);
    uint64_t t3;
    if (REDUCED) {
        t3 = mad_u64_u32(a1, b1, (uint32_t)((uint32_t)(t1 >> 32) + (uint32_t)(t2 >> 32)));
    } else {
        t3 = mad_u64_u32(a1, b1, t1 >> 32);
        CS_KEEP(t3);
        t3 = mad_u64_u32((uint32_t)(t2 >> 32), fp_opaque_one(), t3);
        CS_KEEP(t3);
    }
    return ((u128_t)t3 << 64) | ((t2 << 32) | (uint32_t)t0);
}
__device__ __forceinline__ fp reduce_wide(u128_t v) { return acc_reduce(Acc128{(uint64_t)v, (uint64_t)(v >> 64)}); } // v < 2p 2^64
__device__ __forceinline__ fp reduce_wide_below_p(u128_t v) { return acc_reduce_below_p(Acc128{(uint64_t)v, (uint64_t)(v >> 64)}); } // v < p 2^64
__device__ __forceinline__ Fp2 fp2_mul(Fp2 x, Fp2 y) {
    const u128_t v0 = mul_wide<true>(x.a, y.a), v1 = mul_wide<true>(x.b, y.b), v2 = mul_wide<false>(x.a + x.b, y.a + y.b);
    return {reduce_wide_below_p(v0 + (v1 << 1)), reduce_wide(v2 - v0 + v1)}; // c0 < 3 p^2 < 0.77 p 2^64; c1 < 4 p^2 = 1.03 p 2^64
}
__device__ __forceinline__ Fp2 fp2_sqr(Fp2 x) {
    const u128_t v0 = mul_wide<true>(x.a, x.a), v1 = mul_wide<true>(x.b, x.b), v2 = mul_wide<false>(x.a + x.b, x.a + x.b);
    return {reduce_wide_below_p(v0 + (v1 << 1)), reduce_wide(v2 - v0 + v1)}; // c0 < 3 p^2 < 0.77 p 2^64; c1 < 4 p^2 = 1.03 p 2^64
}
#else
__device__ __forceinline__ Fp2 fp2_mul(Fp2 x, Fp2 y) {
    fp p0 = fp_mul(x.a, y.a), p1 = fp_mul(x.b, y.b);
    fp cross = fp_mul(fp_sub(x.a, x.b), fp_sub(y.b, y.a)); // -(a0-a1)(b0-b1)
    fp c0 = fp_add(p0, fp_dbl(p1));
    return {c0, fp_add(fp_add(c0, p1), cross)};
}
__device__ __forceinline__ Fp2 fp2_sqr(Fp2 x) {
    fp p0 = fp_sqr(x.a), p1 = fp_sqr(x.b);
    fp d = fp_sqr(fp_sub(x.a, x.b));
    fp c0 = fp_add(p0, fp_dbl(p1));
    return {c0, fp_sub(fp_add(c0, p1), d)};
}
#endif
// 1/(a + b u) = (a + 2b - b u) / (a^2 + 2ab - 2b^2)
__device__ inline Fp2 fp2_inv(Fp2 x) {
    fp n = fp_sub(fp_add(fp_sqr(x.a), fp_mul(fp_dbl(x.a), x.b)), fp_dbl(fp_sqr(x.b)));
    fp t = fp_inv(n);
    return {fp_mul(fp_add(x.a, fp_dbl(x.b)), t), fp_mul(fp_neg(x.b), t)};
}

__device__ __forceinline__ Fp2 f6c(const Fp6 &x, int k) { return {x.c[2 * k], x.c[2 * k + 1]}; }
__device__ __forceinline__ Fp6 f6pack(Fp2 c0, Fp2 c1, Fp2 c2) { return {{c0.a, c0.b, c1.a, c1.b, c2.a, c2.b}}; }

__device__ __forceinline__ Fp6 fp6_add(const Fp6 &x, const Fp6 &y) {
    Fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = fp_add(x.c[i], y.c[i]);
    return r;
}
__device__ __forceinline__ Fp6 fp6_sub(const Fp6 &x, const Fp6 &y) {
    Fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = fp_sub(x.c[i], y.c[i]);
    return r;
}
__device__ __forceinline__ Fp6 fp6_dbl(const Fp6 &x) { return fp6_add(x, x); }
__device__ __forceinline__ Fp6 fp6_mul_small(const Fp6 &x, int c) {
    Fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = fp_mul_small(x.c[i], c);
    return r;
}

// Toom/Karatsuba over the cubic extension with v^3 = -v - 1: six F_p2 products
//   d0 = a0 b0, d1 = a1 b1, d2 = a2 b2, e01 = (a0+a1)(b0+b1), e02 = (a0+a2)(b0+b2), e12 = (a1+a2)(b1+b2)
//   c0 = d0 - (e12 - d1 - d2);  c1 = (e01 - d0 - d1) - (e12 - d1 - d2) - d2;  c2 = (e02 - d0 - d2) + d1 - d2
__device__ __forceinline__ Fp6 fp6_combine(Fp2 d0, Fp2 d1, Fp2 d2, Fp2 e01, Fp2 e02, Fp2 e12) {
    Fp2 s = fp2_add(fp2_add(d0, d1), d2);
    Fp2 c0 = fp2_sub(s, e12);
    Fp2 c1 = fp2_sub(fp2_sub(e01, e12), d0);
    Fp2 c2 = fp2_add(fp2_sub(fp2_sub(e02, s), d2), fp2_dbl(d1));
    return f6pack(c0, c1, c2);
}
__device__ __forceinline__ Fp6 fp6_mul(const Fp6 &x, const Fp6 &y) {
    Fp2 a0 = f6c(x, 0), a1 = f6c(x, 1), a2 = f6c(x, 2), b0 = f6c(y, 0), b1 = f6c(y, 1), b2 = f6c(y, 2);
    return fp6_combine(fp2_mul(a0, b0), fp2_mul(a1, b1), fp2_mul(a2, b2), fp2_mul(fp2_add(a0, a1), fp2_add(b0, b1)),
                       fp2_mul(fp2_add(a0, a2), fp2_add(b0, b2)), fp2_mul(fp2_add(a1, a2), fp2_add(b1, b2)));
}
__device__ __forceinline__ Fp6 fp6_sqr(const Fp6 &x) {
    Fp2 a0 = f6c(x, 0), a1 = f6c(x, 1), a2 = f6c(x, 2);
    return fp6_combine(fp2_sqr(a0), fp2_sqr(a1), fp2_sqr(a2), fp2_sqr(fp2_add(a0, a1)), fp2_sqr(fp2_add(a0, a2)),
                       fp2_sqr(fp2_add(a1, a2)));
}
// Inverse through the norm to F_p2 (cofactor form of the 3x3 multiplication matrix of x).
__device__ inline Fp6 fp6_inv(const Fp6 &x) {
    Fp2 a = f6c(x, 0), b = f6c(x, 1), c = f6c(x, 2);
    Fp2 aa = fp2_sqr(a), bb = fp2_sqr(b), cc = fp2_sqr(c);
    // cofactors (up to the common norm)
    Fp2 r0 = fp2_sub(fp2_add(fp2_add(aa, bb), cc), fp2_mul(fp2_sub(fp2_dbl(a), b), c));
    Fp2 r1 = fp2_neg(fp2_add(fp2_mul(a, b), cc));
    Fp2 r2 = fp2_add(fp2_sub(bb, fp2_mul(a, c)), cc);
    // norm = a r0 - c r1 - b r2 + ... ; computed as x * (r0 + r1 v + r2 v^2) constant term
    // constant term of (a + b v + c v^2)(r0 + r1 v + r2 v^2) with v^3 = -v - 1, v^4 = -v^2 - v:
    //   a r0 - (b r2 + c r1)
    Fp2 nrm = fp2_sub(fp2_mul(a, r0), fp2_add(fp2_mul(b, r2), fp2_mul(c, r1)));
    Fp2 t = fp2_inv(nrm);
    return f6pack(fp2_mul(r0, t), fp2_mul(r1, t), fp2_mul(r2, t));
}

__device__ __forceinline__ Fp6 fp6_load(const fp *p) {
    Fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = p[i];
    return r;
}
__device__ __forceinline__ void fp6_store(fp *p, const Fp6 &x) {
#pragma unroll
    for (int i = 0; i < 6; i++) p[i] = x.c[i];
}
// strided access for column-major tables: element i at p[i * stride]
__device__ __forceinline__ Fp6 fp6_load_strided(const fp *p, size_t stride) {
    Fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = p[i * stride];
    return r;
}

} // namespace cs
