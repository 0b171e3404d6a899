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

}} // namespace cs::hostb3
