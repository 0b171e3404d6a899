// Host-side BLAKE3 (hash mode, 32-byte output) for the Fiat-Shamir channel of the prover: seeds, commitments of
// out-of-domain frames and of the FRI remainder are a few KB at most, so this is a plain scalar implementation of the
// public BLAKE3 specification (section 2: compression function, chunk chaining, binary tree of chunk values).
// The bulk hashing (LDE rows, Merkle levels) runs on the GPU in blake3.hip.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace cs { namespace hostb3 {

enum : uint32_t { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };
static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};

static inline uint32_t rotr(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }
static inline void g(uint32_t *v, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
    v[a] = v[a] + v[b] + mx; v[d] = rotr(v[d] ^ v[a], 16);
    v[c] = v[c] + v[d];      v[b] = rotr(v[b] ^ v[c], 12);
    v[a] = v[a] + v[b] + my; v[d] = rotr(v[d] ^ v[a], 8);
    v[c] = v[c] + v[d];      v[b] = rotr(v[b] ^ v[c], 7);
}
// one compression; writes the 8-word chaining value
static inline void compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len, uint32_t flags, uint32_t out[8]) {
    uint32_t v[16], m[16], t[16];
    for (int i = 0; i < 8; i++) v[i] = cv[i];
    for (int i = 0; i < 4; i++) v[8 + i] = IV[i];
    v[12] = (uint32_t)counter; v[13] = (uint32_t)(counter >> 32); v[14] = block_len; v[15] = flags;
    for (int i = 0; i < 16; i++) m[i] = block[i];
    for (int r = 0; r < 7; r++) {
        g(v, 0, 4, 8, 12, m[0], m[1]);   g(v, 1, 5, 9, 13, m[2], m[3]);
        g(v, 2, 6, 10, 14, m[4], m[5]);  g(v, 3, 7, 11, 15, m[6], m[7]);
        g(v, 0, 5, 10, 15, m[8], m[9]);  g(v, 1, 6, 11, 12, m[10], m[11]);
        g(v, 2, 7, 8, 13, m[12], m[13]); g(v, 3, 4, 9, 14, m[14], m[15]);
        for (int i = 0; i < 16; i++) t[i] = m[PERM[i]];
        for (int i = 0; i < 16; i++) m[i] = t[i];
    }
    for (int i = 0; i < 8; i++) out[i] = v[i] ^ v[i + 8];
}
static inline void load_block(const uint8_t *p, size_t len, uint32_t w[16]) {
    uint8_t buf[64] = {0};
    memcpy(buf, p, len);
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)buf[4 * i] | (uint32_t)buf[4 * i + 1] << 8 | (uint32_t)buf[4 * i + 2] << 16 | (uint32_t)buf[4 * i + 3] << 24;
}
// chaining value of one chunk (<= 1024 bytes); `root` adds the ROOT flag to its last block
static inline void chunk_cv(const uint8_t *p, size_t len, uint64_t index, bool root, uint32_t cv[8]) {
    for (int i = 0; i < 8; i++) cv[i] = IV[i];
    const size_t nblk = len == 0 ? 1 : (len + 63) / 64;
    for (size_t b = 0; b < nblk; b++) {
        const size_t bl = (b + 1 == nblk) ? len - 64 * b : 64;
        uint32_t w[16];
        load_block(p + 64 * b, bl, w);
        uint32_t fl = (b == 0 ? CHUNK_START : 0) | (b + 1 == nblk ? CHUNK_END | (root ? ROOT : 0) : 0);
        compress(cv, w, index, (uint32_t)bl, fl, cv);
    }
}
static inline void subtree_cv(const uint8_t *p, size_t len, uint64_t chunk0, bool root, uint32_t cv[8]) {
    if (len <= 1024) { chunk_cv(p, len, chunk0, root, cv); return; }
    size_t left = 1024; // largest power-of-two number of chunks that leaves at least one byte on the right
    while (2 * left < len) left *= 2;
    uint32_t blk[16];
    subtree_cv(p, left, chunk0, false, blk);
    subtree_cv(p + left, len - left, chunk0 + left / 1024, false, blk + 8);
    compress(IV, blk, 0, 64, PARENT | (root ? ROOT : 0), cv);
}
static inline void hash(const uint8_t *p, size_t len, uint8_t out[32]) {
    uint32_t cv[8];
    subtree_cv(p, len, 0, true, cv);
    for (int i = 0; i < 8; i++) for (int b = 0; b < 4; b++) out[4 * i + b] = (uint8_t)(cv[i] >> (8 * b));
}


// Eight public-coin candidates at once: the first 8 bytes (little-endian) of BLAKE3(seed || counter_le64) for counter = c0 .. c0 + 7.
// A draw of the coin rejects values >= p, i.e. three out of four candidates (p / 2^64 = 0.256), so the 238 composition coefficients
// of a TransactionAir proof are ~930 hashes on the critical path between two launches; the candidates of consecutive counters are
// independent, so eight of them go through one pass of the compression function on 8-lane vectors (AVX2 when the CPU has it, checked
// at run time; plain loop otherwise).
typedef uint32_t v8u __attribute__((vector_size(32)));
#if defined(__x86_64__)
#define CS_B3X8_TARGET __attribute__((target("avx2")))
#else
#define CS_B3X8_TARGET
#endif
CS_B3X8_TARGET static inline v8u rotr8(v8u x, int r) { return (x >> r) | (x << (32 - r)); }
#define CS_B3X8_G(a, b, c, d, mx, my)            \
    a = a + b + (mx); d = rotr8(d ^ a, 16);       \
    c = c + d;        b = rotr8(b ^ c, 12);       \
    a = a + b + (my); d = rotr8(d ^ a, 8);        \
    c = c + d;        b = rotr8(b ^ c, 7);
#define CS_B3X8_ROUND(i0, i1, i2, i3, i4, i5, i6, i7, i8, i9, i10, i11, i12, i13, i14, i15)            \
    CS_B3X8_G(s0, s4, s8, s12, m[i0], m[i1]) CS_B3X8_G(s1, s5, s9, s13, m[i2], m[i3])                  \
    CS_B3X8_G(s2, s6, s10, s14, m[i4], m[i5]) CS_B3X8_G(s3, s7, s11, s15, m[i6], m[i7])                \
    CS_B3X8_G(s0, s5, s10, s15, m[i8], m[i9]) CS_B3X8_G(s1, s6, s11, s12, m[i10], m[i11])              \
    CS_B3X8_G(s2, s7, s8, s13, m[i12], m[i13]) CS_B3X8_G(s3, s4, s9, s14, m[i14], m[i15])
CS_B3X8_TARGET static inline void coin_candidates_x8_simd(const uint8_t seed[32], uint64_t c0, uint64_t out[8]) {
    v8u m[16];
    for (int i = 0; i < 8; i++) {
        const uint32_t w = (uint32_t)seed[4 * i] | (uint32_t)seed[4 * i + 1] << 8 | (uint32_t)seed[4 * i + 2] << 16 | (uint32_t)seed[4 * i + 3] << 24;
        m[i] = (v8u){w, w, w, w, w, w, w, w};
    }
    for (int l = 0; l < 8; l++) { m[8][l] = (uint32_t)(c0 + l); m[9][l] = (uint32_t)((c0 + l) >> 32); }
    for (int i = 10; i < 16; i++) m[i] = (v8u){0, 0, 0, 0, 0, 0, 0, 0};
#define CS_SPLAT(x) ((v8u){x, x, x, x, x, x, x, x})
    v8u s0 = CS_SPLAT(IV[0]), s1 = CS_SPLAT(IV[1]), s2 = CS_SPLAT(IV[2]), s3 = CS_SPLAT(IV[3]), s4 = CS_SPLAT(IV[4]), s5 = CS_SPLAT(IV[5]),
        s6 = CS_SPLAT(IV[6]), s7 = CS_SPLAT(IV[7]), s8 = CS_SPLAT(IV[0]), s9 = CS_SPLAT(IV[1]), s10 = CS_SPLAT(IV[2]), s11 = CS_SPLAT(IV[3]),
        s12 = CS_SPLAT(0u), s13 = CS_SPLAT(0u), s14 = CS_SPLAT(40u), s15 = CS_SPLAT((uint32_t)(CHUNK_START | CHUNK_END | ROOT));
#undef CS_SPLAT
    CS_B3X8_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    CS_B3X8_ROUND(2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8)
    CS_B3X8_ROUND(3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1)
    CS_B3X8_ROUND(10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6)
    CS_B3X8_ROUND(12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4)
    CS_B3X8_ROUND(9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7)
    CS_B3X8_ROUND(11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13)
    const v8u o0 = s0 ^ s8, o1 = s1 ^ s9;
    for (int l = 0; l < 8; l++) out[l] = (uint64_t)o0[l] | (uint64_t)o1[l] << 32;
}
#undef CS_B3X8_ROUND
#undef CS_B3X8_G
static inline void coin_candidates_x8(const uint8_t seed[32], uint64_t c0, uint64_t out[8]) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) { coin_candidates_x8_simd(seed, c0, out); return; }
#endif
    for (int l = 0; l < 8; l++) {
        uint8_t buf[40], dg[32];
        memcpy(buf, seed, 32);
        for (int i = 0; i < 8; i++) buf[32 + i] = (uint8_t)((c0 + l) >> (8 * i));
        hash(buf, 40, dg);
        uint64_t v = 0;
        for (int i = 0; i < 8; i++) v |= (uint64_t)dg[i] << (8 * i);
        out[l] = v;
    }
}

}} // namespace cs::hostb3
