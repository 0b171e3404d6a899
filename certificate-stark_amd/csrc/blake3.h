// Host-visible declarations for the Blake3 commitment kernels (blake3.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace cs {
// BLOCK ORDER of the cosets of a trace table whose blowup factor b = 2^log_b exceeds the AIR's constraint-evaluation blowup ce = b >> log_s:
// the table is s = 2^log_s blocks of ce cosets, block r = the LDE cosets k = r (mod s), i.e. the blowup-ce extension with offset
// g w_(b n)^r -- block 0 IS the constraint-evaluation domain, so every evaluator reads a plain [ce][width][n] table whatever the blowup
// factor.  slot = (k mod s) ce + k / s.  log_s = 0: natural order.  (Used inside the prover only; the stage entry points of cstark.h take
// natural order.)
__host__ __device__ inline unsigned lde_coset_slot(unsigned k, unsigned log_b, unsigned log_s) {
    return ((k & ((1u << log_s) - 1)) << (log_b - log_s)) | (k >> log_s);
}
__host__ __device__ inline unsigned lde_slot_coset(unsigned slot, unsigned log_b, unsigned log_s) {
    const unsigned log_ce = log_b - log_s;
    return ((slot & ((1u << log_ce) - 1)) << log_s) | (slot >> log_ce);
}
// leaf i = b*j + k of row j of coset k: d_leaves[32 i ..]; d_lde holds cosets [k0, k0+nk) coset-major (log_s > 0: slots, block order)
hipError_t hash_rows(const uint64_t *d_lde, uint8_t *d_leaves, unsigned width, unsigned log_n, unsigned log_b, unsigned k0, unsigned nk,
                     hipStream_t stream, unsigned log_s = 0);
// d_nodes: 2 * 2^log_leaves digests, leaves in the upper half; fills nodes[1 .. 2^log_leaves)
// Proof of work over the nonces [base, base + count): *d_found <- the smallest one whose Blake3(seed || le64(nonce)) starts with `bits`
// zero bits (low bits of the first 8 bytes read little-endian), or ~0 if there is none in the chunk
hipError_t grind_chunk(const uint8_t seed[32], uint64_t base, uint64_t count, unsigned bits, unsigned long long *d_found, hipStream_t stream);
// one chunk of the same search for `batch` seeds ([batch][8] words) at once; d_found[batch] must hold ~0 before the first chunk
hipError_t grind_batch_chunk(const uint32_t *d_seeds, unsigned batch, uint64_t base, uint64_t count, unsigned bits, unsigned long long *d_found, hipStream_t stream);
// ... with the Sha3 coin (sha3.hip); d_seeds = [batch][4] 64-bit words
hipError_t grind_batch_chunk_sha3(const uint64_t *d_seeds, unsigned batch, uint64_t base, uint64_t count, unsigned bits, unsigned long long *d_found,
                                  hipStream_t stream);
// The Blake3 coin of one FRI layer on the device: d_seed (8 words) <- Blake3(seed || root), *d_alpha = the drawn field element (memory
// form), d_root_out <- the root (8 words).  Same bytes as the host coin of prove.hip.
hipError_t fri_coin(uint32_t *d_seed, const uint8_t *d_root, uint64_t *d_alpha, uint32_t *d_root_out, hipStream_t stream);
hipError_t merkle_build(uint8_t *d_nodes, unsigned log_leaves, hipStream_t stream);
// the same two stages with SHA3-256 (sha3.hip)
hipError_t hash_rows_sha3(const uint64_t *d_lde, uint8_t *d_leaves, unsigned width, unsigned log_n, unsigned log_b, unsigned k0, unsigned nk,
                          hipStream_t stream, unsigned log_s = 0);
hipError_t merkle_build_sha3(uint8_t *d_nodes, unsigned log_leaves, hipStream_t stream);
// `batch` small tables side by side (the batched range prover): table t = columns [t gw, (t + 1) gw) of a coset-major table of
// width_total columns, gw <= 8; its leaves at d_leaves + t leaf_stride (bytes), its tree (<= 2048 leaves) at d_nodes + t node_stride
hipError_t hash_rows_batch(const uint64_t *d_lde, uint8_t *d_leaves, unsigned gw, unsigned width_total, unsigned log_n, unsigned log_b, unsigned batch,
                           size_t leaf_stride, hipStream_t stream);
hipError_t merkle_build_batch(uint8_t *d_nodes, unsigned log_leaves, unsigned batch, size_t node_stride, hipStream_t stream);
hipError_t hash_rows_batch_sha3(const uint64_t *d_lde, uint8_t *d_leaves, unsigned gw, unsigned width_total, unsigned log_n, unsigned log_b, unsigned batch,
                                size_t leaf_stride, hipStream_t stream);
hipError_t merkle_build_batch_sha3(uint8_t *d_nodes, unsigned log_leaves, unsigned batch, size_t node_stride, hipStream_t stream);
} // namespace cs
