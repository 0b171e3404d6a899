/* ORACLE (test infrastructure, not product code).
 *
 * Public interface of the CPU restatement of the reference's hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (certificate-stark_amd/) never does.  Field elements are uint64_t in f63::BaseElement memory form
 * (Montgomery, R = 2^64), matrices are column-major -- the same conventions as include/cstark.h.
 *
 * PARITY STATUS: the AIR-level code (trace, masks, constraints) is pinned by the algebraic known
 * answers listed in SURVEY.md 8(c).  The engine-level code (LDE, Blake3 commitment, evaluator
 * driver) restates the un-vendored winterfell fork (Cargo.toml:20) from its public design; BLAKE3
 * is pinned by the official test vectors, everything else at that level is "parity unpinned".
 */
#ifndef CS_ORACLE_H
#define CS_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#include "../include/cstark.h"

#ifdef __cplusplus
extern "C" {
#endif

/* field helpers (vectorised, for the Python tests) */
void cso_fp_from_u64(const uint64_t *in, uint64_t *out, size_t n);
void cso_fp_to_u64(const uint64_t *in, uint64_t *out, size_t n);
void cso_fp_mul(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void cso_fp_add(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void cso_fp_sub(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void cso_fp_inv(const uint64_t *a, uint64_t *out, size_t n);
void cso_fp_pow(const uint64_t *a, uint64_t e, uint64_t *out, size_t n);
uint64_t cso_fp_root_of_unity(unsigned log_n);

int cso_num_threads(void);

/* gadgets */
void cso_rescue_permutation(uint64_t *state14);
void cso_rescue_round(uint64_t *state14, uint32_t step);
void cso_rescue_enforce_round(uint64_t *result14, const uint64_t *cur14, const uint64_t *next14, const uint64_t *ark28, uint64_t flag);
void cso_rescue_merge(const uint64_t *a7, const uint64_t *b7, uint64_t *out7);
void cso_rescue_digest(const uint64_t *data, size_t n, uint64_t *out7);
void cso_fp6_mul(const uint64_t *a, const uint64_t *b, uint64_t *out);
void cso_fp6_sqr(const uint64_t *a, uint64_t *out);
void cso_fp6_inv(const uint64_t *a, uint64_t *out);
void cso_ecc_double(uint64_t *p18);
void cso_ecc_add(uint64_t *p18, const uint64_t *q18);
void cso_ecc_add_mixed(uint64_t *p18, const uint64_t *q12);
int cso_ecc_on_curve_affine(const uint64_t *q12);
void cso_ecc_scalar_mul_affine(const uint64_t *k_limbs, unsigned n_limbs, const uint64_t *base12, uint64_t *out12);
void cso_schnorr_hash_message(const uint64_t *rx6, const uint64_t *msg28, uint64_t *out7);

/* state-transition AIR */
int cso_tx_build_trace(const cstark_tx_witness *w, uint64_t *trace);
int cso_tx_periodic_columns(unsigned depth, uint64_t *out /*[48][1024]*/);
void cso_tx_evaluate_transition(const uint64_t *cur94, const uint64_t *next94, const uint64_t *periodic48, uint64_t *result115);
long cso_tx_check_trace(const uint64_t *trace, uint32_t n_tx, unsigned depth);
void cso_tx_constraint_degrees(uint32_t *base115, uint32_t *cycles115);

/* engine stages (engine.c) */
void cso_ntt(uint64_t *a, unsigned log_n);
void cso_intt(uint64_t *a, unsigned log_n);
void cso_dft_naive(const uint64_t *a, uint64_t *out, unsigned log_n);
uint64_t cso_fp_generator(void);
void cso_set_num_threads(int n);
void cso_interpolate_columns(uint64_t *cols, uint32_t width, unsigned log_n);
void cso_lde_columns(const uint64_t *coeffs, uint64_t *lde, uint32_t width, unsigned log_n, unsigned log_b, uint64_t offset,
                     uint32_t k0, uint32_t nk);
void cso_blake3(const uint8_t *in, size_t len, uint8_t out[32]);
void cso_sha3_256(const uint8_t *in, size_t len, uint8_t out[32]);
void cso_digest(int hash_fn, const uint8_t *in, size_t len, uint8_t out[32]);
void cso_hash_rows_fn(int hash_fn, const uint64_t *lde, uint8_t *leaves, uint32_t width, unsigned log_n, unsigned log_b, uint32_t k0, uint32_t nk);
void cso_merkle_build_fn(int hash_fn, uint8_t *nodes, unsigned log_leaves);
void cso_hash_rows(const uint64_t *lde, uint8_t *leaves, uint32_t width, unsigned log_n, unsigned log_b, uint32_t k0, uint32_t nk);
void cso_merkle_build(uint8_t *nodes, unsigned log_leaves);
void cso_tx_periodic_table(unsigned depth, unsigned log_n, unsigned log_b, uint64_t *out /*[b][48][1024]*/);
void cso_tx_degree_adjustments(unsigned log_n, unsigned log_b, uint64_t *adj115);
void cso_tx_evaluate_transitions(const uint64_t *lde, uint64_t *out, unsigned depth, unsigned log_n, unsigned log_b, uint32_t k0, uint32_t nk);
void cso_tx_evaluate_constraints(const uint64_t *lde, const cstark_tx_coeffs *cf, const uint64_t pub_inputs[4], uint64_t *out,
                                 unsigned depth, unsigned log_n, unsigned log_b, uint32_t k0, uint32_t nk);

void cso_composition_columns(const uint64_t *combined, uint64_t *out_cols, unsigned log_n, unsigned log_b);
void cso_deep_composition(const uint64_t *trace_lde, const uint64_t *comp_lde, uint32_t width, uint32_t nb, uint64_t z,
                          const uint64_t *ood_trace, const uint64_t *ood_comp, const uint64_t *alpha, const uint64_t *beta,
                          const uint64_t *delta, uint64_t deg_a, uint64_t deg_b, uint64_t *out, unsigned log_n, unsigned log_b,
                          uint32_t k0, uint32_t nk);
void cso_evaluate_polys_at(const uint64_t *coeffs, uint32_t width, unsigned log_n, const uint64_t *points, uint32_t npts, uint64_t *out);
void cso_evaluate_polys_at_ext(const uint64_t *coeffs, uint32_t width, unsigned log_n, const uint64_t *zp, uint64_t *out, int m);
void cso_deep_composition_ext(const uint64_t *trace_lde, const uint64_t *comp_lde, uint32_t width, uint32_t nb, const uint64_t *z2,
                              const uint64_t *ood_trace, const uint64_t *ood_comp, const uint64_t *alpha, const uint64_t *beta,
                              const uint64_t *delta, const uint64_t *deg_ap, const uint64_t *deg_bp, uint64_t *out, unsigned log_n, unsigned log_b, int m);
void cso_fri_fold4_ext(const uint64_t *evals, uint64_t *out, unsigned log_n, uint64_t offset, const uint64_t *alphap, int m);
void cso_fri_fold4(const uint64_t *evals, uint64_t *out, unsigned log_n, uint64_t offset, uint64_t alpha);
void cso_fri_fold(const uint64_t *evals, uint64_t *out, unsigned log_n, unsigned log_f, uint64_t offset, uint64_t alpha);
void cso_fri_fold_ext(const uint64_t *evals, uint64_t *out, unsigned log_n, unsigned log_f, uint64_t offset, const uint64_t *alphap, int m);
uint64_t cso_poly_eval(const uint64_t *co, size_t n, uint64_t x);
uint64_t cso_tx_combined_from_frame(const uint64_t *cur, const uint64_t *next, const cstark_tx_coeffs *cf, const uint64_t pub_inputs[4],
                                    unsigned depth, unsigned log_n, unsigned log_b, uint64_t z);
uint64_t cso_tx_combined_at(const uint64_t *trace_coeffs, const cstark_tx_coeffs *cf, const uint64_t pub_inputs[4],
                            unsigned depth, unsigned log_n, unsigned log_b, uint64_t z);

/* standalone sub-AIRs and the generic driver (air_small.c) */
typedef struct cso_air_desc {
    uint32_t width, n_constraints, cycle_len, log_ce_blowup;
    const uint32_t *base, *cycles;          /* [n_constraints] */
    uint32_t n_assertions;
    const uint32_t *a_reg, *a_last;         /* [n_assertions]: register, 0 = first step / 1 = last step (single assertions) */
    const uint64_t *a_value;
    /* optional generalisation to periodic / sequence assertions (Assertion::periodic / ::sequence): when a_stride is
     * non-NULL, assertion a holds at steps a_first[a] + k * a_stride[a] (a_stride 0: the single step a_first[a]); its value
     * is a_value[a], or column a_seq[a] of the `avals` table passed to cso_air_combine when a_seq[a] >= 0 */
    const uint32_t *a_first, *a_stride;
    const int32_t *a_seq;
} cso_air_desc;
int cso_merkle_build_trace(const cstark_tx_witness *w, uint64_t *trace);
int cso_merkle_periodic_columns(unsigned depth, uint64_t *out);
void cso_merkle_evaluate_transition(const uint64_t *cur, const uint64_t *next, const uint64_t *pv, uint64_t *res);
void cso_merkle_constraint_degrees(uint32_t *base, uint32_t *cycles);
int cso_range_build_trace(uint64_t number_canonical, uint64_t *trace);
uint64_t cso_range_build_trace_bits(const uint64_t *words, uint32_t log_n, uint64_t *trace);
void cso_range_evaluate_transition(const uint64_t *cur, const uint64_t *next, const uint64_t *pv, uint64_t *res);
int cso_rescue_chain_build_trace(const uint64_t *seed7, uint32_t iterations, uint64_t *trace);
void cso_rescue_compute_hash_chain(const uint64_t *seed7, uint32_t length, uint64_t *out7);
void cso_rescue_chain_evaluate_transition(const uint64_t *cur, const uint64_t *next, const uint64_t *pv, uint64_t *res);
void cso_rescue_chain_periodic_columns(uint64_t *out);
int cso_schnorr_build_trace(uint32_t n_sig, const uint64_t *messages, const uint64_t *sig_rx, const uint8_t *sig_s, uint64_t *trace);
void cso_schnorr_aux_columns(uint32_t n_sig, const uint64_t *messages, uint64_t *out);
void cso_schnorr_mask_columns(uint64_t *out);
void cso_schnorr_evaluate_transitions(const uint64_t *lde, const uint64_t *aux, const uint64_t *ptab, uint64_t *out, unsigned log_n, uint32_t k0, uint32_t nk);
void cso_schnorr_evaluate_transition_at(const uint64_t *cur, const uint64_t *next, const uint64_t *m, const uint64_t *pk, const uint64_t *inp, uint64_t *res);
void cso_schnorr_constraint_degrees(uint32_t n_sig, uint32_t *base, uint32_t *cycles);
int cso_schnorr_witness_generate(uint32_t n_sig, uint64_t seed, uint64_t *messages, uint64_t *sig_rx, uint8_t *sig_s);
void cso_air_combine(const cso_air_desc *d, const uint64_t *lde, const uint64_t *evals, const uint64_t *t_alpha, const uint64_t *t_beta,
                     const uint64_t *b_alpha, const uint64_t *b_beta, uint64_t *out, unsigned log_n, unsigned log_b, uint32_t k0, uint32_t nk,
                     int all_cosets, const uint64_t *avals, uint32_t n_avals);
void cso_sequence_value_polys(const uint64_t *values, uint32_t n_seq, uint32_t m, uint32_t first_step, unsigned log_n, uint64_t *out);
void cso_air_evaluate_transitions(int air, const uint64_t *lde, const uint64_t *ptab, uint64_t *out, uint32_t width, uint32_t nc, uint32_t np,
                                  uint32_t cycle_len, unsigned log_n, uint32_t k0, uint32_t nk);
void cso_periodic_table(const uint64_t *cols, uint32_t np, unsigned log_cycle, unsigned log_n, unsigned log_b, uint64_t *out);

/* deterministic witness synthesis (counterpart of TransactionMetadata::build_random, src/lib.rs:235-465).
 * The caller allocates every array of *w (non-const use). */
int cso_tx_witness_generate(cstark_tx_witness *w, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
