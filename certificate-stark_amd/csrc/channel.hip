// The Fiat-Shamir channel on the device: the engine's public coin (ProverChannel / RandomCoin inside Prover::prove,
// /root/reference/src/lib.rs:140 [UPSTREAM-RECALL winterfell v0.3, parity unpinned]) as ONE workgroup per channel step, so that the host
// enqueues a whole proof without reading a root, a frame or a remainder back in between.  The bytes are those of prove.hip's host Coin
// (which stays the reference implementation: CSTARK_HOST_CHANNEL=1, every Sha3 / extension / sub-AIR proof).
//
// Four lanes share a BLAKE3 compression (blake3_quad.cuh): a reseed is one compression of quad 0; the candidates of a draw are hashed
// 256 counters per pass by the 256 quads of the workgroup (a candidate is a field element with probability 0.256) and compacted in
// counter order; the digest of a list of field elements is hashed chunk-parallel (one quad per 1024-byte chunk) and merged by quad 0.
#include "channel.h"
#include "blake3_quad.cuh"
#include "fp.cuh"
#include "../../include/cstark_conventions.h"

namespace cs {
namespace {

constexpr int CT = 1024, NQUAD = CT / 4, MAX_CHUNKS = 64, MAX_CAND = 1024, MAX_POS = 128;
constexpr int PER_PASS = 2 * NQUAD; // candidates per pass of a field-element draw: two counters per quad (half the barriers per candidate)

struct Shared {
    uint32_t msg[NQUAD][16];
    uint32_t seed[8], dig[8];
    uint32_t cvs[MAX_CHUNKS][8];
    uint32_t wave_cnt[CT / 64];
    uint32_t total, ncand;
    uint64_t drawn;           // CHAN_DRAW_POINT
    uint32_t cand[MAX_CAND];
    uint32_t cur[MAX_POS], nxt[MAX_POS];
    uint8_t prefix[32];
};

// words c and 4 + c of a digest held by the lanes of a quad -> the quad's LDS block (words [at, at + 8))
__device__ __forceinline__ void put8(uint32_t *m, unsigned at, unsigned c, uint32_t lo, uint32_t hi) { m[at + c] = lo; m[at + 4 + c] = hi; }

// seed <- Blake3(seed || d[0..8))  (quad 0)
__device__ __forceinline__ void reseed_digest(Shared &S, const uint32_t (&sched)[7], unsigned c, const uint32_t *d) {
    uint32_t *m = S.msg[0];
    m[c] = S.seed[c]; m[4 + c] = S.seed[4 + c]; m[8 + c] = d[c]; m[12 + c] = d[4 + c];
    __builtin_amdgcn_wave_barrier();
    uint32_t lo, hi;
    quad_hash_block(m, sched, c, 64u, lo, hi);
    __builtin_amdgcn_wave_barrier();
    S.seed[c] = lo; S.seed[4 + c] = hi;
    __builtin_amdgcn_wave_barrier();
}

// Blake3(seed || le64(v))[words c, 4 + c] by one quad (its own LDS block)
__device__ __forceinline__ void hash_seed_int(const Shared &S, uint32_t *m, const uint32_t (&sched)[7], unsigned c, uint64_t v, uint32_t &lo, uint32_t &hi) {
    m[c] = S.seed[c]; m[4 + c] = S.seed[4 + c];
    m[8 + c] = c == 0 ? (uint32_t)v : c == 1 ? (uint32_t)(v >> 32) : 0u;
    m[12 + c] = 0;
    __builtin_amdgcn_wave_barrier();
    quad_hash_block(m, sched, c, 40u, lo, hi);
    __builtin_amdgcn_wave_barrier();
}

// order-preserving position of every flagged thread among the flagged threads of the workgroup, counted from `S.total`; afterwards
// S.total holds the new count.  Called by all threads.
__device__ __forceinline__ uint32_t compact_position(Shared &S, bool flag) {
    const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint64_t votes = __ballot(flag);
    if (lane == 0) S.wave_cnt[wave] = (uint32_t)__popcll(votes);
    __syncthreads();
    uint32_t pos = S.total;
    for (unsigned w = 0; w < wave; w++) pos += S.wave_cnt[w];
    pos += (uint32_t)__popcll(votes & ((1ull << lane) - 1));
    __syncthreads();
    if (tid == 0) {
        uint32_t t = S.total;
        for (unsigned w = 0; w < CT / 64; w++) t += S.wave_cnt[w];
        S.total = t;
    }
    __syncthreads();
    return pos;
}

__device__ __forceinline__ void store_draw(const ChanStep &s, Shared &S, uint32_t i, fp val) {
    switch (s.draw) {
    case CHAN_DRAW_LINEAR: s.out[i] = val; break;
    case CHAN_DRAW_COEFFS:
        if (i < 2 * s.a) s.out[(i & 1) * s.stride + (i >> 1)] = val;
        else { const uint32_t r = i - 2 * s.a; s.out[2 * s.stride + (r & 1) * s.b + (r >> 1)] = val; }
        break;
    case CHAN_DRAW_DEEP:
        if (i < s.per * s.a) {
            const uint32_t reg = i / s.per, k = i % s.per;
            if (k == 0) s.out[reg] = val;
            else if (k == 1) s.out[s.a + reg] = val;
        } else {
            const uint32_t r = i - s.per * s.a;
            if (r < s.b) s.out[2 * s.a + r] = val;
            else s.out2[3 + (r - s.b)] = val;
        }
        break;
    case CHAN_DRAW_POINT: S.drawn = val; break;
    default: break;
    }
}

__global__ __launch_bounds__(CT) void k_chan_step(ChanStep s) {
    __shared__ Shared S;
    const unsigned tid = threadIdx.x, quad = tid >> 2, c = tid & 3;
    uint32_t sched[7];
#pragma unroll
    for (int r = 0; r < 7; r++) sched[r] = c_quad_sched[c][r];
    uint32_t *m = S.msg[quad];

    // ---- the coin's seed ---------------------------------------------------------------------------------------------------------
    if (s.init) {
        if (tid < 32) S.prefix[tid] = s.prefix[tid];
        __syncthreads();
        if (quad == 0) { // one chunk: prefix || canonical little-endian public inputs
            const uint32_t total = s.prefix_len + 8 * s.npub, nblocks = total == 0 ? 1 : (total + 63) / 64;
            uint32_t cv_lo = quad_iv_lo(c), cv_hi = quad_iv_hi(c);
            for (uint32_t b = 0; b < nblocks; b++) {
                for (int wi = 0; wi < 4; wi++) {
                    uint32_t word = 0;
                    for (int by = 0; by < 4; by++) {
                        const uint32_t i = 64 * b + 16 * c + 4 * wi + by;
                        uint32_t v = 0;
                        if (i < s.prefix_len) v = S.prefix[i];
                        else if (i < total) {
                            const uint32_t k = i - s.prefix_len;
                            v = (uint32_t)(fp_to_u64(s.pub[k >> 3]) >> (8 * (k & 7))) & 0xFF;
                        }
                        word |= v << (8 * by);
                    }
                    m[4 * c + wi] = word;
                }
                __builtin_amdgcn_wave_barrier();
                const uint32_t len = total - 64 * b < 64 ? total - 64 * b : 64;
                const uint32_t flags = (b == 0 ? CHUNK_START : 0u) | (b + 1 == nblocks ? (uint32_t)(CHUNK_END | ROOT) : 0u);
                uint32_t lo, hi;
                quad_compress(m, sched, c, cv_lo, cv_hi, 0, len, flags, lo, hi);
                __builtin_amdgcn_wave_barrier();
                cv_lo = lo; cv_hi = hi;
            }
            S.seed[c] = cv_lo; S.seed[4 + c] = cv_hi;
        }
    } else if (tid < 8) {
        S.seed[tid] = s.seed[tid];
    }
    __syncthreads();

    // ---- absorb ---------------------------------------------------------------------------------------------------------------------
    for (int k = 0; k < 3; k++) {
        const uint32_t kind = s.absorb[k].kind;
        if (kind == CHAN_NONE) continue;
        if (kind == CHAN_ELEMS) {
            const uint32_t count = s.absorb[k].count, bytes = 8 * count, nchunks = bytes == 0 ? 1 : (bytes + 1023) / 1024;
            const uint64_t *e = (const uint64_t *)s.absorb[k].ptr;
            if (quad < nchunks && quad < MAX_CHUNKS) {
                const uint32_t cbytes = bytes - 1024 * quad < 1024 ? bytes - 1024 * quad : 1024, nblocks = cbytes == 0 ? 1 : (cbytes + 63) / 64;
                uint32_t cv_lo = quad_iv_lo(c), cv_hi = quad_iv_hi(c);
                for (uint32_t b = 0; b < nblocks; b++) {
                    const uint32_t i0 = 128 * quad + 8 * b + 2 * c; // two elements per lane
                    uint64_t v0 = i0 < count ? e[i0] : 0, v1 = i0 + 1 < count ? e[i0 + 1] : 0;
#if !CSTARK_CONV_HASHED_ELEMENT_BYTES_MONTGOMERY
                    if (i0 < count) v0 = fp_to_u64(v0);
                    if (i0 + 1 < count) v1 = fp_to_u64(v1);
#endif
                    m[4 * c] = (uint32_t)v0; m[4 * c + 1] = (uint32_t)(v0 >> 32); m[4 * c + 2] = (uint32_t)v1; m[4 * c + 3] = (uint32_t)(v1 >> 32);
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t len = cbytes - 64 * b < 64 ? cbytes - 64 * b : 64;
                    const uint32_t flags = (b == 0 ? CHUNK_START : 0u) | (b + 1 == nblocks ? (uint32_t)CHUNK_END : 0u) | (b + 1 == nblocks && nchunks == 1 ? (uint32_t)ROOT : 0u);
                    uint32_t lo, hi;
                    quad_compress(m, sched, c, cv_lo, cv_hi, quad, len, flags, lo, hi);
                    __builtin_amdgcn_wave_barrier();
                    cv_lo = lo; cv_hi = hi;
                }
                S.cvs[quad][c] = cv_lo; S.cvs[quad][4 + c] = cv_hi;
            }
            __syncthreads();
            if (quad == 0) {
                // BLAKE3's tree: chaining values are merged left to right through a stack (add_chunk_chaining_value), the last chunk's
                // output joins the stack top down, the final parent carries ROOT.  The stack lives in S.cvs itself (entries below the
                // read position are free): at most log2(chunks) entries.
                uint32_t sp = 0; // stack = S.cvs[0 .. sp)
                uint32_t cur_lo = 0, cur_hi = 0;
                for (uint32_t i = 0; i < nchunks; i++) {
                    cur_lo = S.cvs[i][c]; cur_hi = S.cvs[i][4 + c];
                    __builtin_amdgcn_wave_barrier();
                    if (i + 1 == nchunks) break;
                    uint32_t total = i + 1;
                    while ((total & 1) == 0) { // merge with the stack top: parent(left = top, right = cur)
                        sp--;
                        m[c] = S.cvs[sp][c]; m[4 + c] = S.cvs[sp][4 + c]; m[8 + c] = cur_lo; m[12 + c] = cur_hi;
                        __builtin_amdgcn_wave_barrier();
                        quad_compress(m, sched, c, quad_iv_lo(c), quad_iv_hi(c), 0, 64, PARENT, cur_lo, cur_hi);
                        __builtin_amdgcn_wave_barrier();
                        total >>= 1;
                    }
                    S.cvs[sp][c] = cur_lo; S.cvs[sp][4 + c] = cur_hi; // (sp <= i: never ahead of the read position)
                    sp++;
                    __builtin_amdgcn_wave_barrier();
                }
                while (sp > 0) {
                    sp--;
                    m[c] = S.cvs[sp][c]; m[4 + c] = S.cvs[sp][4 + c]; m[8 + c] = cur_lo; m[12 + c] = cur_hi;
                    __builtin_amdgcn_wave_barrier();
                    quad_compress(m, sched, c, quad_iv_lo(c), quad_iv_hi(c), 0, 64, sp == 0 ? (uint32_t)(PARENT | ROOT) : (uint32_t)PARENT, cur_lo, cur_hi);
                    __builtin_amdgcn_wave_barrier();
                }
                S.dig[c] = cur_lo; S.dig[4 + c] = cur_hi;
                __builtin_amdgcn_wave_barrier();
            }
        } else if (kind == CHAN_DIGEST) {
            if (tid < 8) S.dig[tid] = ((const uint32_t *)s.absorb[k].ptr)[tid];
            __syncthreads();
        }
        if (quad == 0) {
            if (kind == CHAN_INT) {
                uint32_t lo, hi;
                hash_seed_int(S, m, sched, c, s.absorb[k].value, lo, hi);
                S.seed[c] = lo; S.seed[4 + c] = hi;
            } else {
                if (s.absorb[k].copy_out) { ((uint32_t *)s.absorb[k].copy_out)[c] = S.dig[c]; ((uint32_t *)s.absorb[k].copy_out)[4 + c] = S.dig[4 + c]; }
                reseed_digest(S, sched, c, S.dig);
            }
        }
        __syncthreads();
    }

    // ---- draw -----------------------------------------------------------------------------------------------------------------------
    if (tid == 0) { S.total = 0; S.ncand = 0; }
    __syncthreads();
    const unsigned lane = tid & 63;
    if (s.draw >= CHAN_DRAW_LINEAR && s.draw <= CHAN_DRAW_POINT) {
        uint64_t base = CSTARK_CONV_COIN_FIRST_COUNTER;
        for (int pass = 0; pass < 4096 && S.total < s.count; pass++) { // (uniform: S.total is read after a barrier)
            // quad q takes counters base + 2 q and base + 2 q + 1: lane 0 flags the first, lane 2 the second -- lane order = counter order
            uint32_t lo, hi, lo2, hi2;
            hash_seed_int(S, m, sched, c, base + 2 * quad, lo, hi);
            hash_seed_int(S, m, sched, c, base + 2 * quad + 1, lo2, hi2);
            const unsigned q0 = lane & ~3u;
            const uint64_t v1 = (uint64_t)__shfl(lo, (int)q0) | (uint64_t)__shfl(lo, (int)(q0 + 1)) << 32;
            const uint64_t v2 = (uint64_t)__shfl(lo2, (int)q0) | (uint64_t)__shfl(lo2, (int)(q0 + 1)) << 32;
            const uint64_t v = c == 0 ? v1 : v2;
            const bool ok = (c == 0 || c == 2) && (!CSTARK_CONV_COIN_REJECT_ABOVE_P || v < FP_P);
            const uint32_t pos = compact_position(S, ok);
            if (ok && pos < s.count) store_draw(s, S, pos, fp_from_u64(v));
            base += PER_PASS;
        }
        __syncthreads();
        if (s.draw == CHAN_DRAW_POINT && tid == 0) {
            const fp z = S.drawn, zw = fp_mul(z, s.w), ze = fp_pow(z, s.b);
            s.out[0] = z; s.out[1] = zw; s.out[2] = ze;
            if (s.out2) { s.out2[0] = z; s.out2[1] = zw; s.out2[2] = ze; }
        }
    } else if (s.draw == CHAN_DRAW_QUERIES) {
        const uint32_t mask = (1u << s.log_domain) - 1;
        uint64_t base = CSTARK_CONV_COIN_FIRST_COUNTER;
        while (S.total < s.count && S.ncand < MAX_CAND) { // uniform
            const uint32_t nc0 = S.ncand;
            uint32_t lo, hi;
            hash_seed_int(S, m, sched, c, base + quad, lo, hi);
            const uint64_t v = (uint64_t)__shfl(lo, (int)(lane & ~3u)) | (uint64_t)__shfl(lo, (int)((lane & ~3u) + 1)) << 32;
            if (c == 0) S.cand[nc0 + quad] = (uint32_t)v & mask;
            __syncthreads();
            bool keep = false;
            if (tid < NQUAD) { // candidate nc0 + tid: kept unless an earlier candidate has its value (first occurrences, in counter order)
                keep = true;
#if CSTARK_CONV_QUERY_DEDUP
                const uint32_t mine = S.cand[nc0 + tid];
                for (uint32_t j = 0; j < nc0 + tid; j++) keep = keep && S.cand[j] != mine;
#endif
            }
            const uint32_t pos = compact_position(S, keep);
            if (keep && pos < s.count) { S.cur[pos] = S.cand[nc0 + tid]; s.pos[pos] = S.cand[nc0 + tid]; }
            if (tid == 0) S.ncand = nc0 + NQUAD;
            base += NQUAD;
            __syncthreads();
        }
        // cnt[0] = count, or ~0 if the candidates ran out (a domain with fewer than 2 count points is refused by the host: never seen)
        if (tid == 0) s.cnt[0] = S.total >= s.count ? s.count : 0xFFFFFFFFu;
        __syncthreads();
        if (tid < 64) { // the folded positions, layer by layer: ONE wave, entries t and t + 64 per lane, no workgroup barriers
            uint32_t cur_cnt = s.count;
            for (uint32_t l = 0; l < s.n_layers; l++) {
                const uint32_t rmask = (1u << (s.log_domain - (l + 1) * s.log_f)) - 1;
                bool k0 = false, k1 = false;
                uint32_t m0 = 0, m1 = 0;
                if (tid < cur_cnt) {
                    m0 = S.cur[tid] & rmask; k0 = true;
                    for (uint32_t j = 0; j < tid; j++) k0 = k0 && (S.cur[j] & rmask) != m0;
                }
                if (tid + 64 < cur_cnt) {
                    m1 = S.cur[tid + 64] & rmask; k1 = true;
                    for (uint32_t j = 0; j < tid + 64; j++) k1 = k1 && (S.cur[j] & rmask) != m1;
                }
                const uint64_t v0 = __ballot(k0), v1 = __ballot(k1), below = (1ull << tid) - 1;
                const uint32_t n0 = (uint32_t)__popcll(v0), p0 = (uint32_t)__popcll(v0 & below), p1 = n0 + (uint32_t)__popcll(v1 & below);
                __builtin_amdgcn_wave_barrier();
                if (k0) { S.nxt[p0] = m0; s.pos[(size_t)s.slot * (l + 1) + p0] = m0; }
                if (k1) { S.nxt[p1] = m1; s.pos[(size_t)s.slot * (l + 1) + p1] = m1; }
                __builtin_amdgcn_wave_barrier();
                cur_cnt = n0 + (uint32_t)__popcll(v1);
                if (tid < cur_cnt) S.cur[tid] = S.nxt[tid];
                if (tid + 64 < cur_cnt) S.cur[tid + 64] = S.nxt[tid + 64];
                __builtin_amdgcn_wave_barrier();
                if (tid == 0) s.cnt[l + 1] = cur_cnt;
            }
        }
    }
    __syncthreads();
    if (tid < 8) s.seed[tid] = S.seed[tid];
}

} // namespace

hipError_t channel_step(const ChanStep &s, hipStream_t stream) {
    if (!s.seed) return hipErrorInvalidValue;
    if (s.init && s.prefix_len + 8 * s.npub > 1024) return hipErrorInvalidValue;
    for (int k = 0; k < 3; k++)
        if (s.absorb[k].kind == CHAN_ELEMS && (size_t)s.absorb[k].count * 8 > (size_t)MAX_CHUNKS * 1024) return hipErrorInvalidValue;
    if (s.draw == CHAN_DRAW_QUERIES && (s.count == 0 || s.count > MAX_POS || s.log_domain > 31 || s.slot < s.count)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_chan_step, dim3(1), dim3(CT), 0, stream, s);
    return hipGetLastError();
}

} // namespace cs
