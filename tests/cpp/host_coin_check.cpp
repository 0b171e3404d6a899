// The vectorised public-coin candidates (hostblake3.h: eight BLAKE3(seed || counter) per pass, AVX2) against the scalar hash.
// Built and run by tests/test_abi_cpu.py (host code only, no GPU).
#include <cstdio>
#include <cstdlib>
#include "../../certificate-stark_amd/csrc/hostblake3.h"
int main() {
    using namespace cs::hostb3;
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (int rep = 0; rep < 200; rep++) {
        uint8_t seed[32];
        for (int i = 0; i < 32; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; seed[i] = (uint8_t)x; }
        const uint64_t c0 = rep < 4 ? (uint64_t[]){1, 0xFFFFFFFFull - 3, 0xFFFFFFFFFFFFFFF0ull, 0}[rep] : x;
        uint64_t got[8];
        coin_candidates_x8(seed, c0, got);
        for (int l = 0; l < 8; l++) {
            uint8_t buf[40], dg[32];
            memcpy(buf, seed, 32);
            for (int i = 0; i < 8; i++) buf[32 + i] = (uint8_t)((c0 + l) >> (8 * i));
            hash(buf, 40, dg);
            uint64_t v = 0;
            for (int i = 0; i < 8; i++) v |= (uint64_t)dg[i] << (8 * i);
            if (v != got[l]) { std::printf("mismatch rep %d lane %d\n", rep, l); return 1; }
        }
    }
    std::printf("ok avx2=%d\n", (int)__builtin_cpu_supports("avx2"));
    return 0;
}
