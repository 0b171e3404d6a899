// Host-visible declarations for the constraint-evaluation kernels (constraints.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace cs {

// Degree group of transition constraint i of TransactionAir (src/air.rs:76-108): constraints are merged per
// evaluation degree; the AIR has exactly five (base degree; number of 1024-cycles) classes.
//   0: (5;2)  1: (4;2)  2: (3;1)  3: (2;1)  4: (1;1)
__host__ __device__ constexpr int tx_degree_group(int i) {
    return (i < 6 || (i >= 19 && i < 37)) ? 0 : (i >= 6 && i < 18) ? 1 : (i == 18 || (i >= 37 && i < 58)) ? 2 : (i == 92) ? 3 : 4;
}
constexpr unsigned TX_GROUP_BASE[5] = {5, 4, 3, 2, 1};
constexpr unsigned TX_GROUP_CYCLES[5] = {2, 2, 1, 1, 1};

constexpr int CE_RTAB_WORDS = 12288;
constexpr int CE_COEF_WORDS = 115 * 2 + 8; // one coefficient set: alpha[115] | beta[115] | b_alpha[4] | b_beta[4]
constexpr int CE_MAX_SETS = 3;             // coefficient sets merged in one pass (the components of an extension proof)
constexpr int CE_COSET_CONSTS = 10; // per coset: shift, 1/(shift^n - 1), shift^adj[0..5), shift^badj, shift^(n-1), unused

struct CeParams {
    const uint64_t *lde;   // cosets [k0, k0+nk), coset-major [kk][94][n]
    const uint64_t *ptab;  // periodic table [b][48][1024]
    const uint64_t *w;     // [n] powers of w_n
    const uint64_t *coset; // [b][CE_COSET_CONSTS]
    const uint64_t *coef;  // m sets of CE_COEF_WORDS: alpha[115] | beta[115] | b_alpha[4] | b_beta[4]   (device)
    const uint64_t *binv;  // [b][2][n]: 1/(x-1), 1/(x-w^(n-1)) over the evaluation domain
    uint64_t *rtab;        // [m][CE_RTAB_WORDS] scratch: per-proof folded coefficients of the Rescue windows (k_rounds_setup)
    uint64_t *out;         // merged evaluations of coefficient set 0
    uint64_t *out_ext[CE_MAX_SETS - 1]; // ... of sets 1, 2 (m > 1)
    uint32_t m;            // number of coefficient sets (0 reads as 1)
    uint64_t pub[4];       // initial_root[0..2], final_root[0..2]
    const uint64_t *pubd;  // non-null: the 14 public inputs on the device (first / last row of registers 58..64) -- [0], [1], [7], [8] replace pub[]
    uint64_t w_last;       // w_n^(n-1)
    uint32_t adj_mod_n[5]; // degree adjustments reduced mod n (x^adj = shift^adj * w^(j*adj mod n))
    uint32_t badj_mod_n;
    uint32_t log_n, log_b, k0;
    uint32_t nkc;          // split evaluation of a coset window (one rank of a sharded proof): even cosets in p.lde, 1 or 2; 0 = all four
};

constexpr int AIR_MAX_GROUPS = 8;
// generic merge of materialised constraint evaluations (standalone sub-AIRs); all pointers are device memory
struct AirCombineParams {
    const uint64_t *lde, *evals, *w, *shifts; // shifts[k] = g * w_{bn}^k for every LDE coset
    const uint64_t *t_alpha, *t_beta;         // [n_constraints]
    const uint64_t *b_alpha, *b_beta, *a_value;
    const uint32_t *a_reg;                    // [n_assertions]
    const int32_t *a_seq;                     // column of `avals` holding the asserted values, or -1 for the constant a_value
    const uint64_t *avals;                    // [nk][n_avals][n] LDE of the sequence-value polynomials (may be null)
    const uint64_t *tsum;                     // [nk][n] merged transition sum of a fused evaluator (then `evals` is not read), or null
    uint64_t *out;
    uint64_t w_last;
    uint32_t width, n_constraints, n_assertions, n_avals, stride, log_n, k0;
    // grouping (host-built): constraint i uses x^tgrp_adj[t_grp[i]]; assertion a uses divisor / adjustment group a_grp[a]
    const uint32_t *t_grp, *a_grp; // device
    uint32_t n_tgrp, n_agrp;
    uint64_t tgrp_adj[AIR_MAX_GROUPS], agrp_m[AIR_MAX_GROUPS], agrp_zc[AIR_MAX_GROUPS], agrp_badj[AIR_MAX_GROUPS];
    // powers of a point x = shift_k w_n^j without a square-and-multiply per point: x^e = shift_k^e * w_n^((j e) mod n).  Per LDE coset
    // k (host-built): shift_k^adj of every transition group, shift_k^badj and shift_k^m of every assertion group, 1 / (shift_k^n - 1)
    uint64_t tgrp_shift[8][AIR_MAX_GROUPS], agrp_bshift[8][AIR_MAX_GROUPS], agrp_mshift[8][AIR_MAX_GROUPS], zinv_coset[8];
    // optional (all groups or none): cached 1 / (x^m - zc), agrp_inv[g][k (n / m_g) + (j mod n / m_g)] for LDE coset k; null = one
    // inversion per point (Montgomery's trick over the groups)
    const uint64_t *agrp_inv[AIR_MAX_GROUPS];
};
// tab[k][i] = 1 / (shift_m[k] w_n^(i m) - zc), i < n / m, for the b cosets (shift_m[k] = shift_k^m)
hipError_t launch_assert_inverses(uint64_t *d_tab, const uint64_t *d_w, const uint64_t shift_m[8], unsigned b, uint64_t m, uint64_t zc, unsigned log_n,
                                  hipStream_t stream);
hipError_t launch_air_combine(const AirCombineParams &p, unsigned nk, hipStream_t stream);
hipError_t launch_eval_transitions_merkle(const uint64_t *lde, const uint64_t *ptab, uint64_t *out, unsigned log_n, unsigned k0, unsigned nk,
                                          hipStream_t stream);
hipError_t launch_eval_transitions_schnorr(const uint64_t *lde, const uint64_t *aux, const uint64_t *ptab, uint64_t *out, unsigned log_n, unsigned k0,
                                           unsigned nk, hipStream_t stream);
// SchnorrAir, fused: writes sum_i (alpha_i + beta_i x^adj_i) C_i(x) of the nk cosets into p.out (six launches); aux / ptab as for
// launch_eval_transitions_schnorr.  Follow with launch_air_combine on the same parameters with p.tsum = p.out.
// MerkleAir likewise (one launch; ptab as for launch_eval_transitions_merkle)
// d_rtab != null (MERKLE_RTAB_WORDS device words, scratch): the four Rescue round gadgets through their folded form (k_merkle_rounds; blowup
// at most 8), round_group = p.t_grp of the round slots; null: every constraint through the generic frame evaluator
constexpr int MERKLE_RTAB_WORDS = 2048;
hipError_t launch_merkle_fused(const AirCombineParams &p, const uint64_t *ptab, unsigned nk, hipStream_t stream, uint64_t *d_rtab = nullptr,
                               unsigned round_group = 0);
hipError_t launch_schnorr_fused(const AirCombineParams &p, const uint64_t *aux, const uint64_t *ptab, unsigned nk, hipStream_t stream);
// SchnorrAir's doubling / addition gadgets in the degree-split form (all cosets, k0 = 0; constraints.hip): eight polynomials on the even
// cosets, d_even = [8][4][n] (d_coefs_tx_layout: alpha[i] at word i, beta[i] at word 115 + i, device); after their extension to the odd
// cosets (d_odd = [4][8][n]) the recombination into p.out, to which the final addition and the remaining constraints are then added
constexpr int SCHNORR_SPLIT_EC_TABLES = 8, SCHNORR_SPLIT_TABLES = 11; // without / with the final addition's three sums (tables 8..10)
hipError_t launch_schnorr_ec_split(const AirCombineParams &p, const uint64_t *aux, const uint64_t *d_coefs_tx_layout, uint64_t *d_even, hipStream_t stream);
hipError_t launch_schnorr_split_finish(const AirCombineParams &p, const uint64_t *aux, const uint64_t *ptab, const uint64_t *d_even, const uint64_t *d_odd,
                                       unsigned g0, unsigned g1, hipStream_t stream, uint64_t *d_rtab = nullptr, unsigned round_group = 0,
                                       const uint64_t *d_hi = nullptr);
// The final addition on five cosets: its three sums on the even cosets (coset < 0: d_out = tables 8..10 of d_even, [3][4][n]) or directly
// on LDE coset 1 (coset = 1: d_out = [3][n]); launch_schnorr_final_hi: d_hi[q] = (d_odd's table 8 + q on coset 1 - d_direct[q]) / 2.
// With d_hi = [4 odd cosets][3][n] (coset 1 from launch_schnorr_final_hi, cosets 3, 5, 7 its extension) launch_schnorr_split_finish
// recombines the final addition too (tables of SCHNORR_SPLIT_TABLES); without it the tables are SCHNORR_SPLIT_EC_TABLES wide.
hipError_t launch_schnorr_final_split(const AirCombineParams &p, const uint64_t *d_coefs_tx_layout, uint64_t *d_out, int coset, hipStream_t stream);
hipError_t launch_schnorr_final_hi(const AirCombineParams &p, const uint64_t *d_odd, const uint64_t *d_direct, uint64_t *d_hi, uint64_t half_m, hipStream_t stream);
hipError_t launch_eval_transitions_range(const uint64_t *lde, uint64_t *out, unsigned log_n, unsigned nk, hipStream_t stream);
// RescueAir (benches/rescue.rs): ptab [b][29][8]
hipError_t launch_eval_transitions_rescue(const uint64_t *lde, const uint64_t *ptab, uint64_t *out, unsigned log_n, unsigned k0, unsigned nk,
                                          hipStream_t stream);

hipError_t launch_eval_transitions(const CeParams &p, unsigned nk, hipStream_t stream);
constexpr int CE_NUM_PARTS = 9; // launches of the fused evaluation: rounds, dbl0, add0, dbl1, add1, final, lin_a, lin_b, lin_c
hipError_t launch_eval_constraints(const CeParams &p, unsigned nk, hipStream_t stream, hipEvent_t *part_events = nullptr, unsigned done_mask = 0,
                                   bool record_end = true);
// Split evaluation of the Rescue windows (m = 1, all 8 cosets, k0 = 0; constraints.hip): setup of the per-proof tables; the four
// low-degree polynomials on the even cosets, d_even = [4][4][n]; recombination over all cosets from d_even and their extension to
// the odd cosets d_odd = [4 cosets][4][n] (writes p.out, like the first part of launch_eval_constraints).
hipError_t launch_rounds_setup(const CeParams &p, hipStream_t stream);
hipError_t launch_rounds_split(const CeParams &p, uint64_t *d_even, hipStream_t stream);
// split evaluation of a curve gadget (part 1 = doubling of s*G, 2 = addition of G, 3 = doubling of h*P; 1 and 2 write their family's
// four polynomials, 3 adds to the doubling family): d_even_family = [4][4][n]
// part 4 = addition of the public key: its quartic half is a family of its own (d_even_family), its linear half is ADDED to the
// addition family d_even_linear (after part 2 wrote it)
hipError_t launch_ec_split(const CeParams &p, int part, uint64_t *d_even_family, uint64_t *d_even_linear, hipStream_t stream);
constexpr int CE_SPLIT_TABLES = 13, CE_SPLIT_FAM0 = 4; // first family (Rescue windows + linear groups): four polynomials; doubling 3 | addition 2 | addition x bit 2 | final addition 2
// split evaluation of a linear group (part 6, 7, 8): adds to the first family, d_even_family0 = [4][4][n]
hipError_t launch_lin_split(const CeParams &p, int part, uint64_t *d_even_family0, hipStream_t stream);
// the three linear groups in one pass over the frame (k_lin_all); same four polynomials as the three launch_lin_split parts
hipError_t launch_lin_all(const CeParams &p, uint64_t *d_even_family0, hipStream_t stream);
// one rank of a proof sharded by LDE coset (p.k0 even, p.nkc = 1 or 2 even cosets, m = 1): rows [p.nkc + 4][n] = its even cosets'
// complete values, then its share of the four odd cosets (constraints.hip, k_split_finish_shard); d_bit37_all = register 37 on all cosets
hipError_t launch_split_finish_shard(const CeParams &p, const uint64_t *d_even, const uint64_t *d_odd, const uint64_t *d_hi, const uint64_t *d_bit37_all,
                                     uint64_t *d_out, hipStream_t stream);
// the ranks' rows [4 / nkc][nkc + 4][n] -> merged evaluations of all cosets [8][n]
hipError_t launch_shard_combine(const uint64_t *d_parts, uint64_t *d_out, unsigned log_n, unsigned nkc, hipStream_t stream);
// d_hi = [4 odd cosets][2 m][n]: the high parts of the final-addition polynomials on the odd cosets (launch_final_hi + their extension)
hipError_t launch_split_finish(const CeParams &p, const uint64_t *d_even, const uint64_t *d_odd, const uint64_t *d_hi, hipStream_t stream);
// Final addition (degree 5 (n - 1): one n-coefficient block above the 4n the even cosets determine).  coset < 0: the two sums
// (alpha; beta of groups 0, 1 merged) on the four even cosets into the family's tables d_out = [2][4][n] (per set: stride of the
// table block); coset >= 0: on that one coset, d_out = [m][2][n].
hipError_t launch_final_split(const CeParams &p, int coset, uint64_t *d_out, hipStream_t stream);
// hi[c][q][j] = (T(coset 1)[c][q][j] - direct[c][q][j]) / 2: values of the high-part polynomials on LDE coset 1; d_odd as for
// launch_split_finish, d_direct = launch_final_split(coset 1), d_hi = [m][2][n]
hipError_t launch_final_hi(const CeParams &p, const uint64_t *d_odd, const uint64_t *d_direct, uint64_t *d_hi, uint64_t half /* 1/2, memory form */,
                           hipStream_t stream);
hipError_t build_boundary_inverses(uint64_t *d_table, const uint64_t *d_w, const uint64_t *d_coset, uint64_t w_last, unsigned log_n, unsigned log_b,
                                   hipStream_t stream);

} // namespace cs
