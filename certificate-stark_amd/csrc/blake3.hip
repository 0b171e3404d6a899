// K4/K5 -- Blake3-256 row hashing and Merkle tree for the trace commitment.
//
// Engine stage behind `prover.prove(trace)` (/root/reference/src/lib.rs:140); the hash function is the one
// the reference selects (HashFunction::Blake3_256, src/lib.rs:82).  BLAKE3 is implemented from its public
// specification.  A row of W <= 128 field elements is a single chunk (W*8 <= 1024 bytes): ceil(W/8)
// chained compressions.  One lane hashes one row; with the coset-major column layout the 64 lanes of a wave
// read 64 consecutive rows of the same column, i.e. one contiguous 512-byte line per load.
#include "blake3.h"
#include "fp.cuh"
#include "blake3_quad.cuh"
#include "../../include/cstark_conventions.h"
#include <hip/hip_runtime.h>
#include <string.h>

namespace cs {
namespace {

// message word indices for each of the 7 rounds (the BLAKE3 permutation applied repeatedly)
#define B3_ROUND(m, i0, i1, i2, i3, i4, i5, i6, i7, i8, i9, i10, i11, i12, i13, i14, i15) \
    B3_G(s0, s4, s8, s12, m[i0], m[i1]) B3_G(s1, s5, s9, s13, m[i2], m[i3])               \
    B3_G(s2, s6, s10, s14, m[i4], m[i5]) B3_G(s3, s7, s11, s15, m[i6], m[i7])             \
    B3_G(s0, s5, s10, s15, m[i8], m[i9]) B3_G(s1, s6, s11, s12, m[i10], m[i11])           \
    B3_G(s2, s7, s8, s13, m[i12], m[i13]) B3_G(s3, s4, s9, s14, m[i14], m[i15])

// cv <- first 8 words of compress(cv, m, counter = 0, block_len, flags)
__device__ __forceinline__ void compress(uint32_t (&cv)[8], const uint32_t (&m)[16], uint32_t block_len, uint32_t flags) {
    uint32_t s0 = cv[0], s1 = cv[1], s2 = cv[2], s3 = cv[3], s4 = cv[4], s5 = cv[5], s6 = cv[6], s7 = cv[7];
    uint32_t s8 = IV0, s9 = IV1, s10 = IV2, s11 = IV3, s12 = 0, s13 = 0, s14 = block_len, s15 = flags;
    B3_ROUND(m, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    B3_ROUND(m, 2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8)
    B3_ROUND(m, 3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1)
    B3_ROUND(m, 10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6)
    B3_ROUND(m, 12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4)
    B3_ROUND(m, 9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7)
    B3_ROUND(m, 11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13)
    cv[0] = s0 ^ s8; cv[1] = s1 ^ s9; cv[2] = s2 ^ s10; cv[3] = s3 ^ s11;
    cv[4] = s4 ^ s12; cv[5] = s5 ^ s13; cv[6] = s6 ^ s14; cv[7] = s7 ^ s15;
}

// grid = (ceil(n / 256), nk); one lane per row j of coset k0 + blockIdx.y.
// log_s > 0: the table holds the cosets in BLOCK ORDER (blake3.h, lde_slot_coset): slot kk is LDE coset (kk mod ce) 2^log_s + kk / ce,
// ce = 2^(log_b - log_s) -- the prover's trace table when the blowup factor exceeds the AIR's constraint-evaluation blowup.
__global__ __launch_bounds__(256) void k_hash_rows(const uint64_t *__restrict__ lde, uint8_t *__restrict__ leaves, unsigned width, unsigned log_n,
                                                   unsigned log_b, unsigned k0, unsigned log_s) {
    const size_t n = (size_t)1 << log_n;
    const size_t j = blockIdx.x * (size_t)256 + threadIdx.x;
    if (j >= n) return;
    const unsigned kk = blockIdx.y;
    const uint64_t *col = lde + (size_t)kk * width * n + j;
    uint32_t cv[8] = {IV0, IV1, IV2, IV3, IV4, IV5, IV6, IV7};
    const unsigned nblocks = (width + 7) / 8;
    // The eight columns of block b + 1 are requested before block b is compressed.  All waves of a CU run this loop nearly in step, so
    // without the overlap inside a wave the CU alternates between every wave waiting on its loads and every wave compressing: the kernel
    // then takes its memory time PLUS its issue time (2.5 ms for 6.3 GB at 2^20 x 8 x 94: 1.2 + 1.3).
    uint64_t nx[8];
    auto load_block = [&](unsigned b) __attribute__((always_inline)) {
        const unsigned c0 = b * 8;
        const unsigned cnt = width - c0 < 8 ? width - c0 : 8;
#pragma unroll
        for (int i = 0; i < 8; i++) {
#ifdef CS_HASH_NOLOAD // measurement build: the compression work without the table reads
            nx[i] = (unsigned)i < cnt ? (uint64_t)(j * 0x9E3779B97F4A7C15ull + c0 + i) : 0;
#else
            nx[i] = (unsigned)i < cnt ? col[(size_t)(c0 + i) * n] : 0;
#endif
        }
    };
    load_block(0);
    for (unsigned b = 0; b < nblocks; b++) {
        uint32_t m[16];
        const unsigned c0 = b * 8;
        const unsigned cnt = width - c0 < 8 ? width - c0 : 8;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t v = nx[i];
#if !CSTARK_CONV_HASHED_ELEMENT_BYTES_MONTGOMERY
            if ((unsigned)i < cnt) v = fp_to_u64(v); // canonical little-endian bytes
#endif
            m[2 * i] = (uint32_t)v;
            m[2 * i + 1] = (uint32_t)(v >> 32);
        }
#ifndef CS_HASH_NO_PREFETCH
        if (b + 1 < nblocks) load_block(b + 1);
        __builtin_amdgcn_sched_barrier(0);
#endif
        const uint32_t flags = (b == 0 ? CHUNK_START : 0u) | (b + 1 == nblocks ? (CHUNK_END | ROOT) : 0u);
        compress(cv, m, cnt * 8, flags);
#ifdef CS_HASH_NO_PREFETCH
        if (b + 1 < nblocks) load_block(b + 1);
#endif
    }
    const size_t leaf = (j << log_b) + lde_slot_coset(k0 + kk, log_b, log_s);
    uint4 *dst = reinterpret_cast<uint4 *>(leaves + 32 * leaf);
    dst[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    dst[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}

// Narrow tables (one or two compressions per row: the composition columns): one lane per LEAF, the cosets of a row in neighbouring
// lanes, so that the 32-byte digests of a wave are 2 KB of consecutive leaves (leaf = b j + k).  The lane-per-row kernel above writes
// them 32 bytes at a stride of 32 b -- hidden behind twelve compressions for the trace, not behind one: the composition stage 1.55 -> 1.49 ms for the
// 8 x 2^23 composition table.  Reads become 64-byte segments (8 consecutive rows of one coset and column).  All b cosets present.
__global__ __launch_bounds__(256) void k_hash_rows_narrow(const uint64_t *__restrict__ lde, uint8_t *__restrict__ leaves, unsigned width, unsigned log_n,
                                                          unsigned log_b, unsigned log_s) {
    const size_t n = (size_t)1 << log_n;
    const size_t leaf = blockIdx.x * (size_t)256 + threadIdx.x;
    if (leaf >= (n << log_b)) return;
    const size_t j = leaf >> log_b;
    const unsigned kk = lde_coset_slot((unsigned)(leaf & ((1u << log_b) - 1)), log_b, log_s);
    const uint64_t *col = lde + (size_t)kk * width * n + j;
    uint32_t cv[8] = {IV0, IV1, IV2, IV3, IV4, IV5, IV6, IV7};
    const unsigned nblocks = (width + 7) / 8;
    for (unsigned b = 0; b < nblocks; b++) {
        uint32_t m[16];
        const unsigned c0 = b * 8;
        const unsigned cnt = width - c0 < 8 ? width - c0 : 8;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t v = (unsigned)i < cnt ? col[(size_t)(c0 + i) * n] : 0;
#if !CSTARK_CONV_HASHED_ELEMENT_BYTES_MONTGOMERY
            if ((unsigned)i < cnt) v = fp_to_u64(v);
#endif
            m[2 * i] = (uint32_t)v;
            m[2 * i + 1] = (uint32_t)(v >> 32);
        }
        const uint32_t flags = (b == 0 ? CHUNK_START : 0u) | (b + 1 == nblocks ? (CHUNK_END | ROOT) : 0u);
        compress(cv, m, cnt * 8, flags);
    }
    uint4 *dst = reinterpret_cast<uint4 *>(leaves + 32 * leaf);
    dst[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    dst[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}

__device__ __forceinline__ void merge_node(const uint8_t *__restrict__ children, uint8_t *__restrict__ parent) {
    const uint4 *src = reinterpret_cast<const uint4 *>(children);
    const uint4 a = src[0], b = src[1], c = src[2], d = src[3];
    const uint32_t m[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    uint32_t cv[8] = {IV0, IV1, IV2, IV3, IV4, IV5, IV6, IV7};
    compress(cv, m, 64, CHUNK_START | CHUNK_END | ROOT); // Blake3 hash of a 64-byte message
    uint4 *dst = reinterpret_cast<uint4 *>(parent);
    dst[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    dst[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}

// one tree level: parents [cnt, 2 cnt) from children [2 cnt, 4 cnt)
__global__ __launch_bounds__(256) void k_merkle_level(uint8_t *__restrict__ nodes, size_t cnt) {
    const size_t t = blockIdx.x * (size_t)256 + threadIdx.x;
    if (t >= cnt) return;
    const size_t i = cnt + t;
    merge_node(nodes + 64 * i, nodes + 32 * i);
}
// Two tree levels per launch: thread t merges the parents cnt + 2t and cnt + 2t + 1 from their four children and then their own
// parent cnt / 2 + t from the two digests it still holds.  Same number of compressions, half the launches -- the levels below a few
// thousand nodes are bound by launch latency (4.6 us each), and a proof builds eleven trees.
__global__ __launch_bounds__(256) void k_merkle_level2(uint8_t *__restrict__ nodes, size_t cnt) {
    const size_t t = blockIdx.x * (size_t)256 + threadIdx.x;
    if (t >= cnt / 2) return;
    const size_t i0 = cnt + 2 * t;
    const uint4 *src = reinterpret_cast<const uint4 *>(nodes + 64 * i0); // children 2 i0 .. 2 i0 + 3: 128 contiguous bytes
    uint32_t m[16];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint4 a = src[4 * h], b = src[4 * h + 1], c = src[4 * h + 2], d = src[4 * h + 3];
        const uint32_t mm[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
        uint32_t cv[8] = {IV0, IV1, IV2, IV3, IV4, IV5, IV6, IV7};
        compress(cv, mm, 64, CHUNK_START | CHUNK_END | ROOT);
        uint4 *dst = reinterpret_cast<uint4 *>(nodes + 32 * (i0 + h));
        dst[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
        dst[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
#pragma unroll
        for (int q = 0; q < 8; q++) m[8 * h + q] = cv[q];
    }
    uint32_t cv[8] = {IV0, IV1, IV2, IV3, IV4, IV5, IV6, IV7};
    compress(cv, m, 64, CHUNK_START | CHUNK_END | ROOT);
    uint4 *dst = reinterpret_cast<uint4 *>(nodes + 32 * (cnt / 2 + t));
    dst[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    dst[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}
// the last levels (<= 1024 parents) in one workgroup
__global__ __launch_bounds__(1024) void k_merkle_top(uint8_t *__restrict__ nodes, size_t cnt) {
    for (; cnt >= 1; cnt >>= 1) {
        if (threadIdx.x < cnt) {
            const size_t i = cnt + threadIdx.x;
            merge_node(nodes + 64 * i, nodes + 32 * i);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x < 2) reinterpret_cast<uint4 *>(nodes)[threadIdx.x] = make_uint4(0, 0, 0, 0); // nodes[0] unused
}

// ---- the upper levels, bound by latency ------------------------------------------------------------------------------------------------
// Below ~2^17 parents a level no longer fills the GPU: what a launch costs is its dependent chain -- one compression is ~670 dependent
// instructions (3 us) for a single lane, the lane-per-node kernels above chain two or three of them, and k_merkle_top walks eleven levels
// (24 us).  A proof builds about ten trees one after the other, each ending in such a tail.  Here FOUR lanes share a compression, the
// way SIMD implementations of BLAKE3 do: lane c of a quad holds column c of the 4 x 4 state, the column step is one G per lane, the
// diagonal step another after rotating rows 1..3 by 1..3 lanes (DPP quad_perm, no LDS); 7 x (2 G + 6 moves) = ~210 dependent
// instructions.  A workgroup of 256 quads takes 512 nodes and reduces them through nine levels in LDS (every level also goes to the
// node array); a second launch of one workgroup finishes the tree.
// The public coin of a FRI layer on the device (prove.hip's Coin::reseed + Coin::draw for the Blake3 coin, same bytes):
//   seed <- Blake3(seed || root);  alpha = the first of Blake3(seed || le64(counter))[0..8), counter = FIRST_COUNTER, ..., that is a field
//   element (CSTARK_CONV_COIN_REJECT_ABOVE_P; otherwise the first one, reduced), in memory form.
// One wave: every quad recomputes the new seed, then the sixteen quads try sixteen consecutive counters at a time.  The root is also
// copied to root_out, so that the host collects all layer roots with one transfer after the last layer.
__global__ __launch_bounds__(64) void k_fri_coin(uint32_t *__restrict__ seed, const uint8_t *__restrict__ root, uint64_t *__restrict__ alpha_out,
                                                 uint32_t *__restrict__ root_out) {
    __shared__ uint32_t msg[16][16];
    const unsigned tid = threadIdx.x, quad = tid >> 2, c = tid & 3;
    uint32_t sched[7];
#pragma unroll
    for (int r = 0; r < 7; r++) sched[r] = c_quad_sched[c][r];
    const uint32_t *rw = reinterpret_cast<const uint32_t *>(root);
    const uint32_t r_lo = rw[c], r_hi = rw[4 + c];
    if (quad == 0) { root_out[c] = r_lo; root_out[4 + c] = r_hi; }
    uint32_t *m = msg[quad];
    m[c] = seed[c]; m[4 + c] = seed[4 + c]; m[8 + c] = r_lo; m[12 + c] = r_hi;
    __builtin_amdgcn_wave_barrier();
    uint32_t s_lo, s_hi;
    quad_hash64(m, sched, c, s_lo, s_hi); // the reseeded coin: words c and 4 + c
    __builtin_amdgcn_wave_barrier();
    uint64_t counter = CSTARK_CONV_COIN_FIRST_COUNTER;
    // a candidate is a field element with probability 0.256: sixteen at a time, the loop ends after 1.01 passes on average; the bound
    // (2^-6900 to be reached) only guarantees that the wave terminates whatever the hash does
    for (int pass = 0; pass < 1024; pass++) {
        const uint64_t ctr = counter + quad;
        m[c] = s_lo; m[4 + c] = s_hi;
        m[8 + c] = c == 0 ? (uint32_t)ctr : c == 1 ? (uint32_t)(ctr >> 32) : 0u;
        m[12 + c] = 0;
        __builtin_amdgcn_wave_barrier();
        uint32_t lo, hi;
        quad_hash_block(m, sched, c, 40u, lo, hi);
        __builtin_amdgcn_wave_barrier();
        const uint64_t v = (uint64_t)__shfl(lo, (int)(quad * 4)) | (uint64_t)__shfl(lo, (int)(quad * 4 + 1)) << 32; // digest words 0, 1
        const bool ok = !CSTARK_CONV_COIN_REJECT_ABOVE_P || v < FP_P;
        const uint64_t votes = __ballot(ok && c == 0);
        if (votes) {
            const unsigned first = (unsigned)__builtin_ctzll(votes) >> 2; // lowest counter that was accepted
            if (quad == first && c == 0) *alpha_out = fp_from_u64(v);
            break;
        }
        counter += 16;
    }
    if (quad == 0) { seed[c] = s_lo; seed[4 + c] = s_hi; }
}
// Workgroup w reduces the parents [cnt + P0 w, cnt + P0 (w + 1)), P0 = min(cnt, 256), through `levels` levels (P0 >> (levels - 1) >= 1).
// cnt: a power of two; grid = cnt / P0.
__global__ __launch_bounds__(1024) void k_merkle_quad(uint8_t *__restrict__ nodes, size_t cnt, int levels) {
    __shared__ __attribute__((aligned(16))) uint32_t buf[2][512 * 8];
    const unsigned tid = threadIdx.x, quad = tid >> 2, c = tid & 3;
    const unsigned p0 = cnt < 256 ? (unsigned)cnt : 256u;
    const size_t base = (size_t)blockIdx.x * p0;
    uint32_t sched[7];
#pragma unroll
    for (int r = 0; r < 7; r++) sched[r] = c_quad_sched[c][r];
    if (tid < 4 * p0) reinterpret_cast<uint4 *>(buf[0])[tid] = reinterpret_cast<const uint4 *>(nodes + 64 * (cnt + base))[tid]; // 2 p0 children
    __syncthreads();
    for (int l = 0; l < levels; l++) {
        const unsigned p = p0 >> l;
        const uint32_t *cur = buf[l & 1];
        uint32_t *nxt = buf[(l & 1) ^ 1];
        if (quad < p) {
            uint32_t lo, hi;
            quad_hash64(cur + 16 * quad, sched, c, lo, hi);
            uint32_t *g = reinterpret_cast<uint32_t *>(nodes + 32 * ((cnt >> l) + (base >> l) + quad));
            g[c] = lo; g[4 + c] = hi;
            nxt[8 * quad + c] = lo; nxt[8 * quad + 4 + c] = hi;
        }
        __syncthreads();
    }
    if (cnt >> (levels - 1) == 1 && tid < 8) reinterpret_cast<uint32_t *>(nodes)[tid] = 0; // the root was written: nodes[0] unused
}

// ---- batches of small tables (the batched range prover, prove.hip): `batch` independent tables side by side ---------------------------
// Table t owns columns [t gw, (t + 1) gw) of a coset-major table of width_total columns; one lane per leaf (leaf = b j + k of table t),
// digests to leaves + t leaf_stride.  A row holds at most 8 elements (one compression).
__global__ __launch_bounds__(256) void k_hash_rows_batch(const uint64_t *__restrict__ lde, uint8_t *__restrict__ leaves, unsigned gw, unsigned width_total,
                                                         unsigned log_n, unsigned log_b, size_t leaf_stride) {
    const size_t n = (size_t)1 << log_n;
    const size_t leaf = blockIdx.x * (size_t)256 + threadIdx.x;
    if (leaf >= (n << log_b)) return;
    const unsigned t = blockIdx.y;
    const size_t j = leaf >> log_b;
    const unsigned kk = (unsigned)(leaf & ((1u << log_b) - 1));
    const uint64_t *col = lde + ((size_t)kk * width_total + (size_t)t * gw) * n + j;
    uint32_t m[16];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t v = (unsigned)i < gw ? col[(size_t)i * n] : 0;
#if !CSTARK_CONV_HASHED_ELEMENT_BYTES_MONTGOMERY
        if ((unsigned)i < gw) v = fp_to_u64(v);
#endif
        m[2 * i] = (uint32_t)v;
        m[2 * i + 1] = (uint32_t)(v >> 32);
    }
    uint32_t cv[8] = {IV0, IV1, IV2, IV3, IV4, IV5, IV6, IV7};
    compress(cv, m, gw * 8, CHUNK_START | CHUNK_END | ROOT);
    uint4 *dst = reinterpret_cast<uint4 *>(leaves + (size_t)t * leaf_stride + 32 * leaf);
    dst[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    dst[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}
// one workgroup per tree of at most 2048 leaves: every level in turn
__global__ __launch_bounds__(1024) void k_merkle_batch(uint8_t *__restrict__ nodes_all, size_t cnt, size_t node_stride) {
    uint8_t *nodes = nodes_all + (size_t)blockIdx.x * node_stride;
    for (; cnt >= 1; cnt >>= 1) {
        if (threadIdx.x < cnt) {
            const size_t i = cnt + threadIdx.x;
            merge_node(nodes + 64 * i, nodes + 32 * i);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x < 2) reinterpret_cast<uint4 *>(nodes)[threadIdx.x] = make_uint4(0, 0, 0, 0);
}

} // namespace

hipError_t hash_rows_batch(const uint64_t *d_lde, uint8_t *d_leaves, unsigned gw, unsigned width_total, unsigned log_n, unsigned log_b, unsigned batch,
                           size_t leaf_stride, hipStream_t stream) {
    if (gw == 0 || gw > 8 || batch == 0) return hipErrorInvalidValue;
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(k_hash_rows_batch, dim3((unsigned)(((n << log_b) + 255) / 256), batch), dim3(256), 0, stream, d_lde, d_leaves, gw, width_total, log_n,
                       log_b, leaf_stride);
    return hipGetLastError();
}
hipError_t merkle_build_batch(uint8_t *d_nodes, unsigned log_leaves, unsigned batch, size_t node_stride, hipStream_t stream) {
    if (log_leaves < 1 || log_leaves > 11 || batch == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_merkle_batch, dim3(batch), dim3(1024), 0, stream, d_nodes, ((size_t)1 << log_leaves) >> 1, node_stride);
    return hipGetLastError();
}

#ifndef CS_HASH_NARROW_MAX
#define CS_HASH_NARROW_MAX 16 // widest table hashed leaf-per-lane (the 94-column trace table that way: 2.47 vs 2.29 ms)
#endif
hipError_t hash_rows(const uint64_t *d_lde, uint8_t *d_leaves, unsigned width, unsigned log_n, unsigned log_b, unsigned k0, unsigned nk,
                     hipStream_t stream, unsigned log_s) {
    if (width == 0 || width > 128 || log_s > log_b) return hipErrorInvalidValue; // single-chunk rows only
    const size_t n = (size_t)1 << log_n;
    if (width <= CS_HASH_NARROW_MAX && log_b >= 1 && k0 == 0 && nk == (1u << log_b)) {
        hipLaunchKernelGGL(k_hash_rows_narrow, dim3((unsigned)(((n << log_b) + 255) / 256)), dim3(256), 0, stream, d_lde, d_leaves, width, log_n, log_b, log_s);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_hash_rows, dim3((unsigned)((n + 255) / 256), nk), dim3(256), 0, stream, d_lde, d_leaves, width, log_n, log_b, k0, log_s);
    return hipGetLastError();
}

// Proof of work (ProofOptions::grinding_factor): the smallest nonce >= base of this chunk whose Blake3(seed || le64(nonce))[0..8), read
// as a little-endian integer, has its low `bits` bits zero.  One lane per nonce, atomicMin into *found (preset to ~0).
struct GrindSeed { uint32_t w[8]; };
__global__ __launch_bounds__(256) void k_grind(GrindSeed seed, uint64_t base, uint64_t count, uint64_t mask, unsigned long long *__restrict__ found) {
    const uint64_t i = blockIdx.x * (uint64_t)256 + threadIdx.x;
    if (i >= count) return;
    const uint64_t nonce = base + i;
    uint32_t m[16] = {seed.w[0], seed.w[1], seed.w[2], seed.w[3], seed.w[4], seed.w[5], seed.w[6], seed.w[7], (uint32_t)nonce, (uint32_t)(nonce >> 32), 0, 0, 0, 0, 0, 0};
    uint32_t cv[8] = {IV0, IV1, IV2, IV3, IV4, IV5, IV6, IV7};
    compress(cv, m, 40, CHUNK_START | CHUNK_END | ROOT);
    const uint64_t v = (uint64_t)cv[0] | (uint64_t)cv[1] << 32;
    if ((v & mask) == 0) atomicMin(found, (unsigned long long)nonce);
}
hipError_t grind_chunk(const uint8_t seed[32], uint64_t base, uint64_t count, unsigned bits, unsigned long long *d_found, hipStream_t stream) {
    GrindSeed s;
    memcpy(s.w, seed, 32); // little-endian host: the digest bytes are the words
    hipError_t e = hipMemsetAsync(d_found, 0xFF, 8, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_grind, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, s, base, count, bits >= 64 ? ~0ull : ((1ull << bits) - 1), d_found);
    return hipGetLastError();
}
// the same for `batch` seeds at once (the batched range prover): grid.y = proof; found[t] preset to ~0 by the caller before the first chunk
__global__ __launch_bounds__(256) void k_grind_batch(const uint32_t *__restrict__ seeds, uint64_t base, uint64_t count, uint64_t mask,
                                                     unsigned long long *__restrict__ found) {
    const uint64_t i = blockIdx.x * (uint64_t)256 + threadIdx.x;
    const unsigned t = blockIdx.y;
    if (i >= count || found[t] < base) return; // (found in an earlier chunk: nothing smaller is left to find)
    const uint64_t nonce = base + i;
    const uint32_t *s = seeds + 8 * (size_t)t;
    uint32_t m[16] = {s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], (uint32_t)nonce, (uint32_t)(nonce >> 32), 0, 0, 0, 0, 0, 0};
    uint32_t cv[8] = {IV0, IV1, IV2, IV3, IV4, IV5, IV6, IV7};
    compress(cv, m, 40, CHUNK_START | CHUNK_END | ROOT);
    const uint64_t v = (uint64_t)cv[0] | (uint64_t)cv[1] << 32;
    if ((v & mask) == 0) atomicMin(found + t, (unsigned long long)nonce);
}
hipError_t grind_batch_chunk(const uint32_t *d_seeds, unsigned batch, uint64_t base, uint64_t count, unsigned bits, unsigned long long *d_found, hipStream_t stream) {
    hipLaunchKernelGGL(k_grind_batch, dim3((unsigned)((count + 255) / 256), batch), dim3(256), 0, stream, d_seeds, base, count,
                       bits >= 64 ? ~0ull : ((1ull << bits) - 1), d_found);
    return hipGetLastError();
}
hipError_t fri_coin(uint32_t *d_seed, const uint8_t *d_root, uint64_t *d_alpha, uint32_t *d_root_out, hipStream_t stream) {
    hipLaunchKernelGGL(k_fri_coin, dim3(1), dim3(64), 0, stream, d_seed, d_root, d_alpha, d_root_out);
    return hipGetLastError();
}
hipError_t merkle_build(uint8_t *d_nodes, unsigned log_leaves, hipStream_t stream) {
    size_t cnt = ((size_t)1 << log_leaves) >> 1;
    static const bool quad = [] { const char *e = getenv("CSTARK_MERKLE_QUAD"); return !e || atoi(e) != 0; }(); // 0 (tuning / debugging): lane-per-node kernels throughout
    if (quad) {
        while (cnt > ((size_t)1 << 17)) { // wide levels: throughput, one lane per node, two levels per launch
            hipLaunchKernelGGL(k_merkle_level2, dim3((unsigned)((cnt / 2 + 255) / 256)), dim3(256), 0, stream, d_nodes, cnt);
            cnt >>= 2;
        }
        if (cnt > 256) { // 256 parents per workgroup, nine levels: 512 children -> 1
            hipLaunchKernelGGL(k_merkle_quad, dim3((unsigned)(cnt / 256)), dim3(1024), 0, stream, d_nodes, cnt, 9);
            cnt >>= 9;
        }
        if (cnt >= 1) {
            int levels = 1;
            while (((size_t)1 << (levels - 1)) < cnt) levels++;
            hipLaunchKernelGGL(k_merkle_quad, dim3(1), dim3(1024), 0, stream, d_nodes, cnt, levels);
        }
        return hipGetLastError();
    }
    while (cnt > 1024) {
        if (cnt >= 4096) { // two levels: parents [cnt, 2 cnt) and [cnt / 2, cnt)
            hipLaunchKernelGGL(k_merkle_level2, dim3((unsigned)((cnt / 2 + 255) / 256)), dim3(256), 0, stream, d_nodes, cnt);
            cnt >>= 2;
        } else {
            hipLaunchKernelGGL(k_merkle_level, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, d_nodes, cnt);
            cnt >>= 1;
        }
    }
    if (cnt >= 1) hipLaunchKernelGGL(k_merkle_top, dim3(1), dim3(1024), 0, stream, d_nodes, cnt);
    return hipGetLastError();
}

} // namespace cs
