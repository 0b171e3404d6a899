// Keccak-f[1600] / SHA3-256 (FIPS 202), shared by the device kernels (sha3.hip) and the host-side channel (prove.hip).
// HashFunction::Sha3_256 is the second hash the reference's ProofOptions accept (examples/state-transition.rs:67-71).
#pragma once
#include <stdint.h>
#include <string.h>

namespace cs { namespace keccak {

#if defined(__HIPCC__)
#define CS_HD __host__ __device__ __forceinline__
#else
#define CS_HD inline
#endif

CS_HD uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

// 24 rounds on 25 named lanes (fully unrolled: the state stays in registers on the device)
CS_HD void permute(uint64_t (&s)[25]) {
    const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
                             0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
                             0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
                             0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                             0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#pragma unroll 1
    for (int rnd = 0; rnd < 24; rnd++) {
        // theta
        const uint64_t c0 = s[0] ^ s[5] ^ s[10] ^ s[15] ^ s[20], c1 = s[1] ^ s[6] ^ s[11] ^ s[16] ^ s[21], c2 = s[2] ^ s[7] ^ s[12] ^ s[17] ^ s[22];
        const uint64_t c3 = s[3] ^ s[8] ^ s[13] ^ s[18] ^ s[23], c4 = s[4] ^ s[9] ^ s[14] ^ s[19] ^ s[24];
        const uint64_t d0 = c4 ^ rotl(c1, 1), d1 = c0 ^ rotl(c2, 1), d2 = c1 ^ rotl(c3, 1), d3 = c2 ^ rotl(c4, 1), d4 = c3 ^ rotl(c0, 1);
        // theta + rho + pi: b[y + 5 ((2x + 3y) mod 5)] = rotl(s[x + 5y] ^ d[x], r[x][y])
        const uint64_t b0 = s[0] ^ d0, b10 = rotl(s[1] ^ d1, 1), b20 = rotl(s[2] ^ d2, 62), b5 = rotl(s[3] ^ d3, 28), b15 = rotl(s[4] ^ d4, 27);
        const uint64_t b16 = rotl(s[5] ^ d0, 36), b1 = rotl(s[6] ^ d1, 44), b11 = rotl(s[7] ^ d2, 6), b21 = rotl(s[8] ^ d3, 55), b6 = rotl(s[9] ^ d4, 20);
        const uint64_t b7 = rotl(s[10] ^ d0, 3), b17 = rotl(s[11] ^ d1, 10), b2 = rotl(s[12] ^ d2, 43), b12 = rotl(s[13] ^ d3, 25), b22 = rotl(s[14] ^ d4, 39);
        const uint64_t b23 = rotl(s[15] ^ d0, 41), b8 = rotl(s[16] ^ d1, 45), b18 = rotl(s[17] ^ d2, 15), b3 = rotl(s[18] ^ d3, 21), b13 = rotl(s[19] ^ d4, 8);
        const uint64_t b14 = rotl(s[20] ^ d0, 18), b24 = rotl(s[21] ^ d1, 2), b9 = rotl(s[22] ^ d2, 61), b19 = rotl(s[23] ^ d3, 56), b4 = rotl(s[24] ^ d4, 14);
        // chi + iota
        s[0] = b0 ^ (~b1 & b2) ^ RC[rnd]; s[1] = b1 ^ (~b2 & b3); s[2] = b2 ^ (~b3 & b4); s[3] = b3 ^ (~b4 & b0); s[4] = b4 ^ (~b0 & b1);
        s[5] = b5 ^ (~b6 & b7); s[6] = b6 ^ (~b7 & b8); s[7] = b7 ^ (~b8 & b9); s[8] = b8 ^ (~b9 & b5); s[9] = b9 ^ (~b5 & b6);
        s[10] = b10 ^ (~b11 & b12); s[11] = b11 ^ (~b12 & b13); s[12] = b12 ^ (~b13 & b14); s[13] = b13 ^ (~b14 & b10); s[14] = b14 ^ (~b10 & b11);
        s[15] = b15 ^ (~b16 & b17); s[16] = b16 ^ (~b17 & b18); s[17] = b17 ^ (~b18 & b19); s[18] = b18 ^ (~b19 & b15); s[19] = b19 ^ (~b15 & b16);
        s[20] = b20 ^ (~b21 & b22); s[21] = b21 ^ (~b22 & b23); s[22] = b22 ^ (~b23 & b24); s[23] = b23 ^ (~b24 & b20); s[24] = b24 ^ (~b20 & b21);
    }
}

// SHA3-256 of a byte string (host side: channel seeds, small commitments)
inline void sha3_256(const uint8_t *in, size_t len, uint8_t out[32]) {
    uint64_t s[25] = {0};
    uint8_t blk[136];
    while (len >= 136) {
        for (int i = 0; i < 17; i++) { uint64_t w; memcpy(&w, in + 8 * i, 8); s[i] ^= w; } // little-endian host
        permute(s);
        in += 136; len -= 136;
    }
    memset(blk, 0, sizeof blk);
    memcpy(blk, in, len);
    blk[len] ^= 0x06;
    blk[135] ^= 0x80;
    for (int i = 0; i < 17; i++) { uint64_t w; memcpy(&w, blk + 8 * i, 8); s[i] ^= w; }
    permute(s);
    memcpy(out, s, 32);
}

}} // namespace cs::keccak
