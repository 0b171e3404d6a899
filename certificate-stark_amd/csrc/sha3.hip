// Sha3_256 variants of K4 / K5: row hashing and the Merkle levels with SHA3-256 instead of Blake3 (selected by
// ProofOptions::hash_fn; engine: hash_elements / merge of the Sha3_256 hasher).  Lane per row / per parent node, the 25-lane
// Keccak state in registers; a row of `width` elements is width 8-byte little-endian words absorbed 17 per block.
#include "blake3.h"
#include "keccak.cuh"
#include "fp.cuh"
#include "../../include/cstark_conventions.h"

namespace cs {
namespace {

__global__ __launch_bounds__(256) void k_hash_rows_sha3(const uint64_t *__restrict__ lde, uint8_t *__restrict__ leaves, unsigned width, unsigned log_n,
                                                        unsigned log_b, unsigned k0, unsigned log_s) {
    const size_t n = (size_t)1 << log_n;
    const size_t j = blockIdx.x * (size_t)256 + threadIdx.x;
    if (j >= n) return;
    const unsigned kk = blockIdx.y;
    const uint64_t *col = lde + (size_t)kk * width * n + j;
    uint64_t s[25];
#pragma unroll
    for (int i = 0; i < 25; i++) s[i] = 0;
    const unsigned nblocks = width / 17 + 1; // the padding always fits the block that holds the (possibly empty) tail
#pragma unroll 1
    for (unsigned b = 0; b < nblocks; b++) {
        const unsigned c0 = b * 17, cnt = width - c0 < 17 ? width - c0 : 17;
#pragma unroll
        for (int i = 0; i < 17; i++) {
            uint64_t w = (unsigned)i < cnt ? col[(size_t)(c0 + i) * n] : 0;
#if !CSTARK_CONV_HASHED_ELEMENT_BYTES_MONTGOMERY
            if ((unsigned)i < cnt) w = fp_to_u64(w); // canonical little-endian bytes
#endif
            if (b + 1 == nblocks) { // pad10*1 with the SHA-3 domain bits: 0x06 after the message, 0x80 at the end of the rate
                if ((unsigned)i == cnt) w ^= 0x06;
                if (i == 16) w ^= 0x8000000000000000ULL;
            }
            s[i] ^= w;
        }
        keccak::permute(s);
    }
    const size_t leaf = (j << log_b) + lde_slot_coset(k0 + kk, log_b, log_s); // block order of the cosets: blake3.h
    uint4 *dst = reinterpret_cast<uint4 *>(leaves + 32 * leaf);
    dst[0] = make_uint4((uint32_t)s[0], (uint32_t)(s[0] >> 32), (uint32_t)s[1], (uint32_t)(s[1] >> 32));
    dst[1] = make_uint4((uint32_t)s[2], (uint32_t)(s[2] >> 32), (uint32_t)s[3], (uint32_t)(s[3] >> 32));
}

__device__ __forceinline__ void merge_node_sha3(const uint8_t *__restrict__ children, uint8_t *__restrict__ parent) {
    const uint4 *src = reinterpret_cast<const uint4 *>(children);
    uint64_t s[25];
#pragma unroll
    for (int i = 0; i < 25; i++) s[i] = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint4 v = src[q];
        s[2 * q] = ((uint64_t)v.y << 32) | v.x;
        s[2 * q + 1] = ((uint64_t)v.w << 32) | v.z;
    }
    s[8] ^= 0x06;
    s[16] ^= 0x8000000000000000ULL;
    keccak::permute(s);
    uint4 *dst = reinterpret_cast<uint4 *>(parent);
    dst[0] = make_uint4((uint32_t)s[0], (uint32_t)(s[0] >> 32), (uint32_t)s[1], (uint32_t)(s[1] >> 32));
    dst[1] = make_uint4((uint32_t)s[2], (uint32_t)(s[2] >> 32), (uint32_t)s[3], (uint32_t)(s[3] >> 32));
}
__global__ __launch_bounds__(256) void k_merkle_level_sha3(uint8_t *__restrict__ nodes, size_t cnt) {
    const size_t t = blockIdx.x * (size_t)256 + threadIdx.x;
    if (t >= cnt) return;
    const size_t i = cnt + t;
    merge_node_sha3(nodes + 64 * i, nodes + 32 * i);
}
__global__ __launch_bounds__(1024) void k_merkle_top_sha3(uint8_t *__restrict__ nodes, size_t cnt) {
    for (; cnt >= 1; cnt >>= 1) {
        if (threadIdx.x < cnt) {
            const size_t i = cnt + threadIdx.x;
            merge_node_sha3(nodes + 64 * i, nodes + 32 * i);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x < 2) reinterpret_cast<uint4 *>(nodes)[threadIdx.x] = make_uint4(0, 0, 0, 0);
}

// batches of small tables (see blake3.hip: k_hash_rows_batch / k_merkle_batch); a row holds at most 8 elements
__global__ __launch_bounds__(256) void k_hash_rows_batch_sha3(const uint64_t *__restrict__ lde, uint8_t *__restrict__ leaves, unsigned gw, unsigned width_total,
                                                              unsigned log_n, unsigned log_b, size_t leaf_stride) {
    const size_t n = (size_t)1 << log_n;
    const size_t leaf = blockIdx.x * (size_t)256 + threadIdx.x;
    if (leaf >= (n << log_b)) return;
    const unsigned t = blockIdx.y;
    const size_t j = leaf >> log_b;
    const unsigned kk = (unsigned)(leaf & ((1u << log_b) - 1));
    const uint64_t *col = lde + ((size_t)kk * width_total + (size_t)t * gw) * n + j;
    uint64_t s[25];
#pragma unroll
    for (int i = 0; i < 25; i++) s[i] = 0;
#pragma unroll
    for (int i = 0; i < 17; i++) {
        uint64_t w = (unsigned)i < gw ? col[(size_t)i * n] : 0;
#if !CSTARK_CONV_HASHED_ELEMENT_BYTES_MONTGOMERY
        if ((unsigned)i < gw) w = fp_to_u64(w);
#endif
        if ((unsigned)i == gw) w ^= 0x06;
        if (i == 16) w ^= 0x8000000000000000ULL;
        s[i] ^= w;
    }
    keccak::permute(s);
    uint4 *dst = reinterpret_cast<uint4 *>(leaves + (size_t)t * leaf_stride + 32 * leaf);
    dst[0] = make_uint4((uint32_t)s[0], (uint32_t)(s[0] >> 32), (uint32_t)s[1], (uint32_t)(s[1] >> 32));
    dst[1] = make_uint4((uint32_t)s[2], (uint32_t)(s[2] >> 32), (uint32_t)s[3], (uint32_t)(s[3] >> 32));
}
__global__ __launch_bounds__(1024) void k_merkle_batch_sha3(uint8_t *__restrict__ nodes_all, size_t cnt, size_t node_stride) {
    uint8_t *nodes = nodes_all + (size_t)blockIdx.x * node_stride;
    for (; cnt >= 1; cnt >>= 1) {
        if (threadIdx.x < cnt) {
            const size_t i = cnt + threadIdx.x;
            merge_node_sha3(nodes + 64 * i, nodes + 32 * i);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x < 2) reinterpret_cast<uint4 *>(nodes)[threadIdx.x] = make_uint4(0, 0, 0, 0);
}

} // namespace

hipError_t hash_rows_batch_sha3(const uint64_t *d_lde, uint8_t *d_leaves, unsigned gw, unsigned width_total, unsigned log_n, unsigned log_b, unsigned batch,
                                size_t leaf_stride, hipStream_t stream) {
    if (gw == 0 || gw > 8 || batch == 0) return hipErrorInvalidValue;
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(k_hash_rows_batch_sha3, dim3((unsigned)(((n << log_b) + 255) / 256), batch), dim3(256), 0, stream, d_lde, d_leaves, gw, width_total,
                       log_n, log_b, leaf_stride);
    return hipGetLastError();
}
hipError_t merkle_build_batch_sha3(uint8_t *d_nodes, unsigned log_leaves, unsigned batch, size_t node_stride, hipStream_t stream) {
    if (log_leaves < 1 || log_leaves > 11 || batch == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_merkle_batch_sha3, dim3(batch), dim3(1024), 0, stream, d_nodes, ((size_t)1 << log_leaves) >> 1, node_stride);
    return hipGetLastError();
}

hipError_t hash_rows_sha3(const uint64_t *d_lde, uint8_t *d_leaves, unsigned width, unsigned log_n, unsigned log_b, unsigned k0, unsigned nk,
                          hipStream_t stream, unsigned log_s) {
    if (width == 0 || width > 128 || log_s > log_b) return hipErrorInvalidValue;
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(k_hash_rows_sha3, dim3((unsigned)((n + 255) / 256), nk), dim3(256), 0, stream, d_lde, d_leaves, width, log_n, log_b, k0, log_s);
    return hipGetLastError();
}
// Proof of work with the Sha3 coin (blake3.hip k_grind / k_grind_batch are the Blake3 forms): SHA3-256(seed || le64(nonce)), one
// Keccak block; seeds = [batch][4] 64-bit words, found[batch] preset to ~0, grid.y = proof
__global__ __launch_bounds__(256) void k_grind_sha3(const uint64_t *__restrict__ seeds, uint64_t base, uint64_t count, uint64_t mask,
                                                    unsigned long long *__restrict__ found) {
    const uint64_t i = blockIdx.x * (uint64_t)256 + threadIdx.x;
    const unsigned t = blockIdx.y;
    if (i >= count || found[t] < base) return;
    const uint64_t nonce = base + i;
    uint64_t s[25];
#pragma unroll
    for (int q = 0; q < 25; q++) s[q] = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) s[q] = seeds[4 * (size_t)t + q];
    s[4] = nonce;
    s[5] = 0x06;                       // pad10*1 with the SHA-3 domain bits after the 40-byte message
    s[16] = 0x8000000000000000ULL;     // ... and at the end of the 136-byte rate
    keccak::permute(s);
    if ((s[0] & mask) == 0) atomicMin(found + t, (unsigned long long)nonce);
}
hipError_t grind_batch_chunk_sha3(const uint64_t *d_seeds, unsigned batch, uint64_t base, uint64_t count, unsigned bits, unsigned long long *d_found,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(k_grind_sha3, dim3((unsigned)((count + 255) / 256), batch), dim3(256), 0, stream, d_seeds, base, count,
                       bits >= 64 ? ~0ull : ((1ull << bits) - 1), d_found);
    return hipGetLastError();
}
hipError_t merkle_build_sha3(uint8_t *d_nodes, unsigned log_leaves, hipStream_t stream) {
    size_t cnt = ((size_t)1 << log_leaves) >> 1;
    for (; cnt > 1024; cnt >>= 1)
        hipLaunchKernelGGL(k_merkle_level_sha3, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, d_nodes, cnt);
    if (cnt >= 1) hipLaunchKernelGGL(k_merkle_top_sha3, dim3(1), dim3(1024), 0, stream, d_nodes, cnt);
    return hipGetLastError();
}

} // namespace cs
