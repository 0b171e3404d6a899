// BLAKE3 pieces shared by the commitment kernels (blake3.hip) and the device-side Fiat-Shamir channel (channel.hip): constants, the G
// function, and the FOUR-LANES-PER-COMPRESSION form (lane c of a quad holds column c of the 4 x 4 state; DPP quad rotations between
// the column and the diagonal step).  Everything lives in an anonymous namespace: one copy per translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cs {
namespace {

constexpr uint32_t IV0 = 0x6A09E667u, IV1 = 0xBB67AE85u, IV2 = 0x3C6EF372u, IV3 = 0xA54FF53Au, IV4 = 0x510E527Fu, IV5 = 0x9B05688Cu,
                   IV6 = 0x1F83D9ABu, IV7 = 0x5BE0CD19u;
enum : uint32_t { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };

__device__ __forceinline__ uint32_t rotr(uint32_t x, int r) { return __funnelshift_r(x, x, r); } // v_alignbit_b32

#define B3_G(a, b, c, d, mx, my)            \
    a = a + b + (mx); d = rotr(d ^ a, 16);  \
    c = c + d;        b = rotr(b ^ c, 12);  \
    a = a + b + (my); d = rotr(d ^ a, 8);   \
    c = c + d;        b = rotr(b ^ c, 7);


__constant__ uint32_t c_quad_sched[4][7] = { // lane c, round r: bytes = message word indices of (column mx, my, diagonal mx, my)
#define QS(r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15, c) \
    ((uint32_t)(c == 0 ? r0 : c == 1 ? r2 : c == 2 ? r4 : r6) | (uint32_t)(c == 0 ? r1 : c == 1 ? r3 : c == 2 ? r5 : r7) << 8 | \
     (uint32_t)(c == 0 ? r8 : c == 1 ? r10 : c == 2 ? r12 : r14) << 16 | (uint32_t)(c == 0 ? r9 : c == 1 ? r11 : c == 2 ? r13 : r15) << 24)
#define QROW(c)                                                                                                                       \
    {QS(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, c), QS(2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8, c),         \
     QS(3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1, c), QS(10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6, c),         \
     QS(12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4, c), QS(9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7, c),         \
     QS(11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13, c)}
    QROW(0), QROW(1), QROW(2), QROW(3)
#undef QROW
#undef QS
};
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true); }
// Blake3 of the single-block message m[0..15] (LDS; block_len bytes, zero-padded) by the four lanes of a quad: lane c returns words c
// and 4 + c of the digest
__device__ __forceinline__ uint32_t quad_iv_lo(unsigned c) { return c == 0 ? IV0 : c == 1 ? IV1 : c == 2 ? IV2 : IV3; }
__device__ __forceinline__ uint32_t quad_iv_hi(unsigned c) { return c == 0 ? IV4 : c == 1 ? IV5 : c == 2 ? IV6 : IV7; }
// The general compression by a quad: chaining value words c and 4 + c in (cv_lo, cv_hi) of lane c, block m[0..15] (LDS), chunk
// counter t, block length and flags; lane c returns words c and 4 + c of the new chaining value.
__device__ __forceinline__ void quad_compress(const uint32_t *m, const uint32_t (&sched)[7], unsigned c, uint32_t cv_lo, uint32_t cv_hi, uint32_t t_lo,
                                              uint32_t block_len, uint32_t flags, uint32_t &lo, uint32_t &hi) {
    uint32_t w[28];
#pragma unroll
    for (int r = 0; r < 7; r++)
#pragma unroll
        for (int q = 0; q < 4; q++) w[4 * r + q] = m[(sched[r] >> (8 * q)) & 15];
    uint32_t a = cv_lo, b = cv_hi, cc = quad_iv_lo(c), d = c == 0 ? t_lo : c == 2 ? block_len : c == 3 ? flags : 0u;
#pragma unroll
    for (int r = 0; r < 7; r++) {
        B3_G(a, b, cc, d, w[4 * r], w[4 * r + 1])
        b = quad_perm<0x39>(b); cc = quad_perm<0x4E>(cc); d = quad_perm<0x93>(d);  // lane c <- lanes c + 1, c + 2, c + 3: the diagonals
        B3_G(a, b, cc, d, w[4 * r + 2], w[4 * r + 3])
        b = quad_perm<0x93>(b); cc = quad_perm<0x4E>(cc); d = quad_perm<0x39>(d);  // and back to columns
    }
    lo = a ^ cc;
    hi = b ^ d;
}
__device__ __forceinline__ void quad_hash_block(const uint32_t *m, const uint32_t (&sched)[7], unsigned c, uint32_t block_len, uint32_t &lo, uint32_t &hi) {
    uint32_t w[28];
#pragma unroll
    for (int r = 0; r < 7; r++)
#pragma unroll
        for (int q = 0; q < 4; q++) w[4 * r + q] = m[(sched[r] >> (8 * q)) & 15];
    const uint32_t iv_lo = quad_iv_lo(c), iv_hi = quad_iv_hi(c);
    uint32_t a = iv_lo, b = iv_hi, cc = iv_lo, d = c == 2 ? block_len : c == 3 ? (uint32_t)(CHUNK_START | CHUNK_END | ROOT) : 0u;
#pragma unroll
    for (int r = 0; r < 7; r++) {
        B3_G(a, b, cc, d, w[4 * r], w[4 * r + 1])
        b = quad_perm<0x39>(b); cc = quad_perm<0x4E>(cc); d = quad_perm<0x93>(d);  // lane c <- lanes c + 1, c + 2, c + 3: the diagonals
        B3_G(a, b, cc, d, w[4 * r + 2], w[4 * r + 3])
        b = quad_perm<0x93>(b); cc = quad_perm<0x4E>(cc); d = quad_perm<0x39>(d);  // and back to columns
    }
    lo = a ^ cc;
    hi = b ^ d;
}
__device__ __forceinline__ void quad_hash64(const uint32_t *m, const uint32_t (&sched)[7], unsigned c, uint32_t &lo, uint32_t &hi) {
    quad_hash_block(m, sched, c, 64u, lo, hi);
}

} // namespace
} // namespace cs
