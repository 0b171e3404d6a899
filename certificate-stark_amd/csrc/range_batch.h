// Device stages of the batched range prover (range_batch.hip; host side: cstark_range_prove_batch in prove.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace cs {

constexpr unsigned RB_LOG_N = 6, RB_N = 64, RB_LOG_B = 3, RB_LDE = 512, RB_CE = 2; // RANGE_LOG rows (src/range/mod.rs:34), blowup 8, 2 composition columns

// per-launch constants shared by the stages (host-computed field elements, memory form)
struct RangeBatchConsts {
    uint64_t shift[8];      // g w_512^k
    uint64_t zinv[8];       // 1 / (shift_k^64 - 1)
    uint64_t w_last;        // w_64^63
    uint64_t adj[2], badj;  // degree adjustments of the two transition constraints and of the single-step assertions
    uint64_t inv128, ginv;  // 1 / 128, 1 / g
    uint64_t offset_inv, inv4; // FRI folding: 1 / g, 1 / 4
    const uint64_t *w64, *winv128, *winv512; // twiddle tables (device): powers of w_64, of w_128^-1, of w_512^-1
};

// B proofs, tables laid out as 2 B columns of 64 rows: column 2 t + c = register c of proof t
hipError_t rb_trace(const uint64_t *d_numbers_canonical, uint64_t *d_trace, unsigned batch, hipStream_t stream);
// merged constraint evaluations on the two cosets of the constraint-evaluation domain (LDE cosets 0 and 4): d_out[t][2][64]
// d_coefs[t][8] = t_alpha[2] t_beta[2] b_alpha[2] b_beta[2]; d_numbers: memory form
hipError_t rb_combine(const RangeBatchConsts &c, const uint64_t *d_lde, const uint64_t *d_coefs, const uint64_t *d_numbers, uint64_t *d_out, unsigned batch,
                      hipStream_t stream);
// composition polynomial of each proof: d_combined[t][2][64] -> coefficient columns d_ccoef[2 t + i][64]
hipError_t rb_composition(const RangeBatchConsts &c, const uint64_t *d_combined, uint64_t *d_ccoef, unsigned batch, hipStream_t stream);
// out-of-domain frame: d_out[t][6] = T0(z) T1(z) T0(z w) T1(z w) H0(z^2) H1(z^2)
hipError_t rb_ood(const RangeBatchConsts &c, const uint64_t *d_coeffs, const uint64_t *d_ccoef, const uint64_t *d_z, uint64_t *d_out, unsigned batch, hipStream_t stream);
// DEEP composition at the 512 points of the LDE domain, natural order: d_layer[t][512]; d_dcoef[t][8] = alpha[2] beta[2] delta[2] deg_a deg_b
hipError_t rb_deep(const RangeBatchConsts &c, const uint64_t *d_lde, const uint64_t *d_clde, const uint64_t *d_z, const uint64_t *d_ood, const uint64_t *d_dcoef,
                   uint64_t *d_layer, unsigned batch, hipStream_t stream);
// FRI folding of each proof's 512 evaluations with its own alpha: d_out[t][128]
hipError_t rb_fold(const RangeBatchConsts &c, const uint64_t *d_layer, const uint64_t *d_alpha, uint64_t *d_out, unsigned batch, hipStream_t stream);
// openings: per proof nq trace rows [2] + paths [9][32], nq composition rows [2] + paths, np layer rows [4] + paths [7][32] into a slot
// of `slot` bytes (layout: rb_open_layout)
struct RangeBatchOpen {
    const uint64_t *lde, *clde, *layer;       // [8][2 B][64], [8][2 B][64], [B][512]
    const uint8_t *tnodes, *cnodes, *lnodes;  // [B][1024][32], [B][1024][32], [B][256][32]
    const uint32_t *pos, *lpos, *lcount;      // [B][nq], [B][nq], [B]
    uint8_t *out;
    uint32_t nq, batch, n_layers;
    size_t slot;
};
hipError_t rb_open(const RangeBatchOpen &o, hipStream_t stream);

} // namespace cs
