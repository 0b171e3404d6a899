// Device-side arithmetic in the 63-bit prime field p = 2^62 + 2^56 + 2^55 + 1 used by the reference
// (winterfell-fork f63::BaseElement; modulus at /root/reference/src/range/tests.rs:59).
//
// Elements are uint64_t in Montgomery form (R = 2^64), reduced to [0,p) -- the in-memory form of
// BaseElement -- so HBM buffers can be shared byte-for-byte with a Rust caller.
//
// CDNA4 notes: there is no 64x64 multiplier; a product is four v_mad_u64_u32.  Because
// p = P1 * 2^32 + 1, -p^-1 mod 2^32 = 0xFFFFFFFF: each 32-bit REDC step is one negate plus one
// v_mad_u64_u32 (the low word cancels by construction and only contributes a carry bit).  Two more
// v_mad_u64_u32 (multiplies by an opaque 1) carry a word into a 64-bit sum where the compiler would
// otherwise emit two moves and a 64-bit add: eight v_mad_u64_u32 per modular multiplication, 52 issue
// cycles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cs {

typedef uint64_t fp;

constexpr uint64_t FP_P = 0x4180000000000001ULL;
constexpr uint32_t FP_P1 = 0x41800000u;            // high word of p (low word is 1)
constexpr uint64_t FP_ONE = 0x3b7ffffffffffffdULL; // 2^64 mod p
constexpr uint64_t FP_R2 = 0x32734c36b7b1d512ULL;  // 2^128 mod p
constexpr uint64_t FP_INV_ALPHA = 3146514939656186539ULL; // Rescue inverse S-box exponent (rescue.rs:383)

__device__ __forceinline__ uint64_t mad_u64_u32(uint32_t a, uint32_t b, uint64_t c) {
    return (uint64_t)a * b + c; // one v_mad_u64_u32
}

__device__ __forceinline__ fp fp_reduce_once(uint64_t r) { return r >= FP_P ? r - FP_P : r; }

__device__ __forceinline__ fp fp_add(fp a, fp b) { return fp_reduce_once(a + b); } // a + b < 2^64
__device__ __forceinline__ fp fp_sub(fp a, fp b) { return a >= b ? a - b : a + (FP_P - b); }
__device__ __forceinline__ fp fp_neg(fp a) { return a ? FP_P - a : 0; }
// a - b + p in (0, 2p): admissible as the FIRST factor of fp_mul / fp_mul_lazy (12 instead of 20 issue cycles)
__device__ __forceinline__ uint64_t fp_sub_lazy(fp a, fp b) { return a + FP_P - b; }
__device__ __forceinline__ fp fp_dbl(fp a) { return fp_reduce_once(a << 1); }

// A register holding 1 that the compiler cannot see through: x * one + c stays ONE v_mad_u64_u32 (4 issue cycles) where the
// compiler's own form of "64-bit value + zero-extended 32-bit word" is two moves and a v_lshl_add_u64 (8 cycles).  The barrier
// macro stops the reassociation pass from pulling such a chain apart again; neither emits an instruction in the loop.
__device__ __forceinline__ uint32_t fp_opaque_one() {
    uint32_t v;
    asm("v_mov_b32 %0, 1" : "=v"(v));
    return v;
}
#define CS_KEEP(x) asm("" : "+v"(x))

// Montgomery product without the final conditional subtraction: result in (0, 2p) for a < 2p, b < p; in general the result is
// below a b / 2^64 + p for a, b < 2p (every intermediate sum below keeps its headroom up to b1 < 3 * 2^30): fp_inv_sbox's lazy chain.
// Word-serial REDC with q = 2^32 - t0 (never 0): (T + q p) / 2^32 = (T >> 32) + q P1 + 1 exactly, and
// q P1 + 1 = ~t0 * P1 + (P1 + 1), so a reduction step is one NOT folded into one v_mad_u64_u32 whose addend
// carries the constant K = P1 + 1 -- no carry bit to materialise.  (q = 2^32 when t0 = 0 merely adds p.)
// Eight v_mad_u64_u32, two NOTs, one move and one 32-bit add: K joins the first step through (t >> 32) * 1 + K, the second
// as the 32-bit sum (w >> 32) + K (w < 2^32 b1 + 2^32 with b1 < 2^30.1, so the sum stays below 2^32), and v >> 32 enters
// through a multiply by one.  52 issue cycles with the conditional subtraction instead of 60 (tools/isa_mix.py).
__device__ __forceinline__ uint64_t fp_mul_lazy(uint64_t a, uint64_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32);
    const uint32_t b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    constexpr uint64_t K = (uint64_t)FP_P1 + 1;
#ifdef CS_FP_MUL_PLAIN // measurement builds: the form left to the compiler (6 multiply-adds, 4 64-bit adds, 4 moves)
    const uint64_t t_ = (uint64_t)a0 * b0;
    const uint64_t u_ = mad_u64_u32(a1, b0, (t_ >> 32) + K);
    const uint64_t v_ = mad_u64_u32(~(uint32_t)t_, FP_P1, u_);
    const uint64_t w_ = mad_u64_u32(a0, b1, (uint32_t)v_);
    const uint64_t x_ = mad_u64_u32(a1, b1, (w_ >> 32) + (v_ >> 32) + K);
    return mad_u64_u32(~(uint32_t)w_, FP_P1, x_);
#endif
    const uint32_t one = fp_opaque_one();
    const uint64_t t = (uint64_t)a0 * b0;
    uint64_t s = mad_u64_u32((uint32_t)(t >> 32), one, K);
    CS_KEEP(s);
    uint64_t u = mad_u64_u32(a1, b0, s);
    CS_KEEP(u);
    const uint64_t v = mad_u64_u32(~(uint32_t)t, FP_P1, u);        // (a * b0 + q p) / 2^32  < 2^63.7
    const uint64_t w = mad_u64_u32(a0, b1, (uint32_t)v);
    const uint32_t c = (uint32_t)(w >> 32) + (uint32_t)K;
    uint64_t x = mad_u64_u32(a1, b1, c);
    CS_KEEP(x);
    x = mad_u64_u32((uint32_t)(v >> 32), one, x);
    CS_KEEP(x);
    return mad_u64_u32(~(uint32_t)w, FP_P1, x);                     // < a1 b1 + 2^33 + p < 2p
}
__device__ __forceinline__ fp fp_mul(fp a, fp b) { return fp_reduce_once(fp_mul_lazy(a, b)); }
__device__ __forceinline__ fp fp_sqr(fp a) { return fp_mul(a, a); }

__device__ __forceinline__ fp fp_from_u64(uint64_t x) { return fp_mul(x % FP_P, FP_R2); }
__device__ __forceinline__ uint64_t fp_to_u64(fp a) { return fp_mul(a, 1); }

__device__ inline fp fp_pow(fp base, uint64_t e) {
    fp r = FP_ONE;
    while (e) {
        if (e & 1) r = fp_mul(r, base);
        base = fp_sqr(base);
        e >>= 1;
    }
    return r;
}
// a^(p-2), p - 2 = 0b 1000001 0 1^55: x^65 by six squarings and a product, one squaring for the zero, then eleven windows of five
// ones (five squarings, * x^31): 66 squarings + 16 products instead of the 117 of the generic square-and-multiply.  0 -> 0.
__device__ inline fp fp_inv(fp a) {
    const fp a3 = fp_mul(fp_sqr(a), a), a7 = fp_mul(fp_sqr(a3), a), a15 = fp_mul(fp_sqr(a7), a), a31 = fp_mul(fp_sqr(a15), a);
    fp r = a;
#pragma unroll 1
    for (int i = 0; i < 6; i++) r = fp_sqr(r);
    r = fp_sqr(fp_mul(r, a));
#pragma unroll 1
    for (int k = 0; k < 11; k++) {
#pragma unroll
        for (int i = 0; i < 5; i++) r = fp_sqr(r);
        r = fp_mul(r, a31);
    }
    return r;
}

// x^INV_ALPHA for the fixed 62-bit exponent 0b 101011 (10)^27 11 (rescue.rs:383), as an addition chain on its pattern: x^42 = 0b101010
// from x^2, x^3, x^5, x^10, x^20, x^21; the prefix 0b101011 = x^42 * x; nine times (six squarings, * x^42) append the 27 pairs "10";
// two squarings and * x^3 append the final "11".  60 squarings + 14 products = 74 field products (plain square-and-multiply: 93; a
// generic width-3 sliding window, tools/gen_invsbox_chain.py: 78).  Checked against the oracle's pow() in tests/test_gpu_field.py.
__device__ inline fp fp_inv_sbox(fp x) {
#ifdef CS_INVSBOX_W3 // measurement builds: the generic width-3 sliding window (78 products)
    {
        const fp y2 = fp_sqr(x), y3 = fp_mul(y2, x), y5 = fp_mul(y3, y2), y7 = fp_mul(y5, y2);
        fp r = y5;
#pragma unroll 1
        for (int i = 0; i < 4; i++) r = fp_sqr(r);
        r = fp_mul(r, y7);
#pragma unroll 1
        for (int k = 0; k < 13; k++) {
#pragma unroll
            for (int i = 0; i < 4; i++) r = fp_sqr(r);
            r = fp_mul(r, y5);
        }
        return fp_mul(fp_sqr(fp_sqr(fp_sqr(r))), y3);
    }
#endif
    // The chain runs on UNREDUCED values: fp_mul_lazy(a, b) < a b / 2^64 + p holds for both factors below 2p (its word sums keep
    // their headroom: b1 < 3 * 2^30), and with p / 2^64 = 0.2559 a chain of squarings of a value below 1.78 p interrupted by products
    // with reduced elements stays below 1.78 p (six squarings from 1.45 p: 1.54, 1.61, 1.66, 1.71, 1.74, 1.78; times x^42: 1.46).
    // Only the elements that are second factors again and again (x^2, x^3, x^42) and the result are reduced: 5 conditional
    // subtractions instead of 74 on a chain in which every instruction waits for the one before it.
    const fp x2 = fp_sqr(x), x3 = fp_mul(x2, x);
    const uint64_t x5 = fp_mul_lazy(x3, x2), x10 = fp_mul_lazy(x5, x5), x21 = fp_mul_lazy(fp_mul_lazy(x10, x10), x);
    const fp x42 = fp_reduce_once(fp_mul_lazy(x21, x21));
    uint64_t r = fp_mul_lazy(x42, x);
#pragma unroll 1
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i < 6; i++) r = fp_mul_lazy(r, r);
        r = fp_mul_lazy(r, x42);
    }
    r = fp_mul_lazy(r, r);
    r = fp_mul_lazy(r, r);
    return fp_reduce_once(fp_mul_lazy(r, x3));
}

// small-integer multiples by repeated addition (|c| <= 4), for the linear steps of the curve formulas
__device__ __forceinline__ fp fp_mul_small(fp a, int c) {
    fp r;
    int k = c < 0 ? -c : c;
    switch (k) {
        case 0: r = 0; break;
        case 1: r = a; break;
        case 2: r = fp_dbl(a); break;
        case 3: r = fp_add(fp_dbl(a), a); break;
        default: r = fp_dbl(fp_dbl(a)); break;
    }
    return c < 0 ? fp_neg(r) : r;
}

} // namespace cs

// value of the next lane (lane l reads lane l + 1; lane 63 gets `last`): one DPP wave shift per dword.  With the coset-major
// layout the next row of a column IS the next lane's current row, so a kernel that needs both loads each column once.
namespace cs {
__device__ __forceinline__ uint64_t wave_next(uint64_t v, uint64_t last) {
    const unsigned lo = __builtin_amdgcn_update_dpp((unsigned)last, (unsigned)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
    const unsigned hi = __builtin_amdgcn_update_dpp((unsigned)(last >> 32), (unsigned)(v >> 32), 0x130, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}
} // namespace cs

// ---- lazy (unreduced) 128-bit accumulation ------------------------------------------------------------
// Sums of products of reduced elements are accumulated in 128 bits and Montgomery-reduced once.  A product
// of two reduced elements is < p^2 < 2^124.07, so seven of them fit below 2^126.9; `fold()` brings the
// accumulator back under 2p * 2^64 (< 2^127.04) so that another seven terms can be added without overflow.
namespace cs {

struct Acc128 {
    uint64_t lo, hi;
};

__device__ __forceinline__ Acc128 acc_zero() { return {0, 0}; }

// acc += a * b for REDUCED a, b (full 128-bit product, no reduction).  The two middle carries t1 >> 32 and t2 >> 32 are below
// 2^30.1 each for reduced operands, so they are added in 32 bits; the accumulation is one four-word carry chain.
__device__ __forceinline__ void acc_mad(Acc128 &acc, uint64_t a, uint64_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t t0 = (uint64_t)a0 * b0;
    const uint64_t t1 = mad_u64_u32(a1, b0, t0 >> 32);
    const uint64_t t2 = mad_u64_u32(a0, b1, (uint32_t)t1);
    const uint64_t t3 = mad_u64_u32(a1, b1, (uint32_t)((uint32_t)(t1 >> 32) + (uint32_t)(t2 >> 32)));
    unsigned cy;
    const uint32_t l0 = __builtin_addc((uint32_t)acc.lo, (uint32_t)t0, 0u, &cy);
    const uint32_t l1 = __builtin_addc((uint32_t)(acc.lo >> 32), (uint32_t)t2, cy, &cy);
    const uint32_t h0 = __builtin_addc((uint32_t)acc.hi, (uint32_t)t3, cy, &cy);
    const uint32_t h1 = __builtin_addc((uint32_t)(acc.hi >> 32), (uint32_t)(t3 >> 32), cy, &cy);
    acc.lo = ((uint64_t)l1 << 32) | l0;
    acc.hi = ((uint64_t)h1 << 32) | h0;
}
// keep the value (mod p * 2^64 multiples are free: they vanish after reduction) below 2p * 2^64
__device__ __forceinline__ void acc_fold(Acc128 &acc) {
    if (acc.hi >= 2 * FP_P) acc.hi -= 2 * FP_P;
}
// The same with ONE conditional subtraction (12 issue cycles saved) for an accumulator whose high word is at most p - 2^32.
// PRECONDITION acc.hi <= p - 2^32 (NOT merely hi < p): the second REDC step gives r = (2^32 - 1 - v0) P1 + hi + (v >> 32) + P1 + 1 with
// (v >> 32) + P1 + 1 <= 2 (P1 + 1), i.e. r <= 2^32 P1 + hi + 2 P1 + 2 = p + hi + 2 P1 + 1 < p + hi + 2^31.04, and r < 2p needs
// hi < p - 2 P1 - 1, which hi <= p - 2^32 guarantees.  Both callers: the F_p2 component a0 b0 + 2 a1 b1 < 3 p^2 = 0.77 p * 2^64
// (tower.cuh).  tests/test_gpu_field.py drives the boundary hi = p - 2^32 and the first failing hi = p - 2 P1 - 1.
__device__ __forceinline__ fp acc_reduce_below_p(const Acc128 &acc) {
    constexpr uint64_t K = (uint64_t)FP_P1 + 1;
    const uint64_t v = mad_u64_u32(~(uint32_t)acc.lo, FP_P1, (acc.lo >> 32) + K);
    return fp_reduce_once(mad_u64_u32(~(uint32_t)v, FP_P1, acc.hi + (v >> 32) + K));
}
// Montgomery reduction of an accumulator < 2p * 2^64 to a fully reduced element (same two REDC steps as fp_mul_lazy)
__device__ __forceinline__ fp acc_reduce(const Acc128 &acc) {
    constexpr uint64_t K = (uint64_t)FP_P1 + 1;
    const uint64_t v = mad_u64_u32(~(uint32_t)acc.lo, FP_P1, (acc.lo >> 32) + K); // (lo + q p) / 2^32 < 2^62.1
    uint64_t r = mad_u64_u32(~(uint32_t)v, FP_P1, acc.hi + (v >> 32) + K);        // < 2p + p + 2^33 < 2^64
    if (r >= 2 * FP_P) r -= 2 * FP_P;
    return fp_reduce_once(r);
}

} // namespace cs
