/* ORACLE (test infrastructure, not product code).
 *
 * CPU restatement of the reference's standalone sub-AIRs and of a generic
 * constraint-evaluation driver for them:
 *   MerkleAir / MerkleProver       src/merkle/update/air.rs:36-177, src/merkle/update/prover.rs:19-116
 *   RangeProofAir / RangeProver    src/range/air.rs:23-105, src/range/prover.rs:15-84
 *   Rescue hash-chain AIR          benches/rescue.rs:128-360  (BASELINE config 0: CPU only)
 * The driver restates winterfell's merge of transition and single-step boundary constraints
 * [UPSTREAM-RECALL, see engine.c]; the constraint-evaluation blowup is the smallest power of two >=
 * max(base degree + number of cycles) as AirContext computes it, which can be smaller than the LDE blowup
 * (MerkleAir: 4 under the default blowup 8) -- the evaluation domain is then the sub-coset of the LDE
 * domain with stride lde_blowup / ce_blowup.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "gadgets.h"

/* ---- MerkleAir ----------------------------------------------------------------------------------- */
enum { MK_W = 65, MK_CYCLE = 512, MK_NC = 106, MK_NP = 33 };

/* identical per-transaction recurrence as rows 0..511 of the composite trace (air_tx.c), plus the
 * index-bit poke of src/merkle/update/prover.rs:72-77 (global row 1 only) */
int cso_merkle_build_trace(const cstark_tx_witness *w, uint64_t *trace) {
    const size_t n = (size_t)w->n_tx * MK_CYCLE;
    const unsigned depth = w->merkle_depth;
    const size_t hash_len = 8 * depth + 7;
    if (hash_len > MK_CYCLE - 1 || depth == 0) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t t = 0; t < w->n_tx; t++) {
        const fp *sv = w->s_old_values + 14 * t, *rv = w->r_old_values + 14 * t;
        const fp *sb = w->s_paths + 7 * (depth + 1) * t, *rb = w->r_paths + 7 * (depth + 1) * t;
        fp delta = w->deltas[t], st[MK_W];
        memset(st, 0, sizeof st);
        memcpy(st, sv, 14 * sizeof(fp));
        memcpy(st + 15, sv, 14 * sizeof(fp));
        st[15 + 12] = fp_sub(st[15 + 12], delta);
        st[15 + 13] = fp_add(st[15 + 13], FP_ONE);
        memcpy(st + 29, rv, 14 * sizeof(fp));
        memcpy(st + 44, rv, 14 * sizeof(fp));
        st[44 + 12] = fp_add(st[44 + 12], delta);
        memcpy(st + 58, w->initial_roots + 7 * t, 7 * sizeof(fp));
        const size_t base = t * MK_CYCLE;
        for (int c = 0; c < MK_W; c++) trace[(size_t)c * n + base] = st[c];
        for (size_t step = 0; step < MK_CYCLE - 1; step++) {
            if (step < hash_len) {
                for (int blk = 0; blk < 2; blk++) { /* update_merkle_update_auth_state, trace.rs:96-136 */
                    fp *s = st + 29 * blk;
                    const fp *branch = blk ? rb : sb;
                    uint64_t index = blk ? w->r_indices[t] : w->s_indices[t];
                    size_t cyc = step / 8, pos = step % 8;
                    if (pos < 7) { rescue_apply_round(s, step); rescue_apply_round(s + 15, step); }
                    else {
                        const fp *node = branch + 7 * (cyc + 1);
                        int bit = (index >> cyc) & 1;
                        if (!bit) for (int i = 0; i < 7; i++) { s[7 + i] = node[i]; s[22 + i] = node[i]; }
                        else for (int i = 0; i < 7; i++) { s[7 + i] = s[i]; s[22 + i] = s[15 + i]; s[i] = node[i]; s[15 + i] = node[i]; }
                        s[14] = bit ? FP_ONE : 0;
                    }
                }
                if (step == hash_len - 1) for (int i = 0; i < 7; i++) st[58 + i] = st[44 + i];
            }
            for (int c = 0; c < MK_W; c++) trace[(size_t)c * n + base + step + 1] = st[c];
        }
    }
    trace[(size_t)14 * n + 1] = FP_ONE;
    trace[(size_t)43 * n + 1] = FP_ONE;
    return 0;
}

/* src/merkle/update/air.rs:182-212; columns: setup, hash, hash_input(period 8), finish, hash_mask, 28 ark */
int cso_merkle_periodic_columns(unsigned depth, uint64_t *out /*[33][512]*/) {
    const size_t hash_len = 8 * depth + 7;
    if (hash_len > MK_CYCLE - 1 || depth == 0) return -1;
    memset(out, 0, (size_t)MK_NP * MK_CYCLE * sizeof(fp));
#define COL(c) (out + (size_t)(c) * MK_CYCLE)
    COL(0)[0] = FP_ONE;
    for (size_t i = 0; i < hash_len; i++) { COL(1)[i] = FP_ONE; COL(4)[i] = (i % 8) != 7 ? FP_ONE : 0; }
    for (size_t i = 0; i < MK_CYCLE; i++) COL(2)[i] = (i % 8) == 7 ? FP_ONE : 0;
    COL(3)[hash_len - 1] = FP_ONE;
    for (int j = 0; j < 28; j++) for (size_t i = 0; i < MK_CYCLE; i++) COL(5 + j)[i] = CS_ARK_MONT[(i % 8) * 28 + j];
#undef COL
    return 0;
}

static void mk_auth(fp *res, const fp *cur, const fp *next, const fp *ark, fp tx_hash, fp hash_input, fp hash_flag) {
    fp hash_copy = fp_mul(tx_hash, c_not(fp_add(hash_flag, hash_input))), hash_init = fp_mul(tx_hash, hash_input);
    fp bit = next[14], not_bit = c_not(bit);
    agg(res, 14, tx_hash, c_is_binary(bit));
    for (int k = 0; k < 2; k++) {
        int b = 15 * k;
        rescue_enforce_round(res + b, cur + b, next + b, ark, hash_flag);
        for (int i = 0; i < 7; i++) {
            agg(res, b + i, hash_copy, c_are_equal(cur[b + i], next[b + i]));
            agg(res, b + i, hash_init, fp_mul(not_bit, c_are_equal(cur[b + i], next[b + i])));
            agg(res, b + 7 + i, hash_init, fp_mul(bit, c_are_equal(cur[b + i], next[b + 7 + i])));
        }
    }
    for (int i = 0; i < 7; i++) agg(res, i, hash_init, fp_mul(bit, c_are_equal(next[15 + i], next[i])));
    for (int i = 7; i < 14; i++) agg(res, i, hash_init, fp_mul(not_bit, c_are_equal(next[15 + i], next[i])));
}
/* MerkleAir::evaluate_transition, src/merkle/update/air.rs:64-141 + evaluate_constraints :215-289 */
void cso_merkle_evaluate_transition(const uint64_t *cur, const uint64_t *next, const uint64_t *pv, uint64_t *res) {
    memset(res, 0, MK_NC * sizeof(fp));
    fp setup = pv[0], tx_hash = pv[1], hash_input = pv[2], finish = pv[3], hash_flag = pv[4];
    const fp *ark = pv + 5;
    for (int i = 0; i < 12; i++) {
        agg(res, 65 + i, setup, c_are_equal(cur[i], cur[15 + i]));
        agg(res, 65 + 12 + i, setup, c_are_equal(cur[29 + i], cur[44 + i]));
    }
    agg(res, 65 + 24, setup, c_are_equal(cur[29 + 13], cur[44 + 13]));
    agg(res, 90, setup, c_are_equal(fp_sub(cur[12], cur[15 + 12]), fp_sub(cur[44 + 12], cur[29 + 12])));
    agg(res, 91, setup, c_are_equal(cur[15 + 13], fp_add(cur[13], FP_ONE)));
    mk_auth(res, cur, next, ark, tx_hash, hash_input, hash_flag);
    mk_auth(res + 29, cur + 29, next + 29, ark, tx_hash, hash_input, hash_flag);
    fp not_finish = c_not(finish);
    for (int i = 0; i < 7; i++) {
        agg(res, 58 + i, not_finish, c_are_equal(next[58 + i], cur[58 + i]));
        agg(res, 58 + i, finish, c_are_equal(next[58 + i], next[44 + i]));
        agg(res, 92 + i, finish, c_are_equal(cur[15 + i], cur[29 + i]));
        agg(res, 99 + i, finish, c_are_equal(next[i], cur[58 + i]));
    }
}
/* transition_constraint_degrees(512), src/merkle/update/air.rs:371-401 */
void cso_merkle_constraint_degrees(uint32_t *base, uint32_t *cycles) {
    for (int i = 0; i < MK_NC; i++) { base[i] = 1; cycles[i] = 1; }
    for (int b = 0; b < 58; b += 29) { for (int i = 0; i < 29; i++) base[b + i] = 3; base[b + 14] = 2; }
}

/* ---- RangeProofAir ------------------------------------------------------------------------------- */
/* src/range/prover.rs:24-43, :65-84: 2 registers x 64 rows.  The standalone prover passes range_log - 1 = 63 to the
 * update function (:38), so the 63 steps consume bits 62..0 of the canonical value: inputs must be < 2^63
 * (src/range/tests.rs:44-52 uses 2^63 - 1 as the maximum). */
int cso_range_build_trace(uint64_t number_canonical, uint64_t *trace /*[2][64]*/) {
    fp bit = 0, acc = 0;
    trace[0] = 0; trace[64] = 0;
    for (int step = 0; step < 63; step++) {
        bit = ((number_canonical >> (62 - step)) & 1) ? FP_ONE : 0;
        acc = fp_add(fp_dbl(acc), bit);
        trace[step + 1] = bit;
        trace[64 + step + 1] = acc;
    }
    return 0;
}
/* The same accumulator over a longer trace (BASELINE.json config "range-proof AIR, 2^16 steps": SYNTHETIC, the reference's trace
 * is fixed at 64 rows, src/range/mod.rs:34).  n = 2^log_n rows; the value is the (n-1)-bit integer V given as n/64 little-endian
 * 64-bit words (top bit clear); row q >= 1 holds bit (n-1-q) of V and acc_q = 2 acc_(q-1) + bit = (V >> (n-1-q)) mod p, exactly the
 * update rule of src/range/prover.rs:74-84.  For log_n = 6 this is cso_range_build_trace.  Returns V mod p (memory form), the
 * value asserted at the last row (src/range/air.rs:79-86). */
uint64_t cso_range_build_trace_bits(const uint64_t *words, uint32_t log_n, uint64_t *trace /*[2][n]*/) {
    const size_t n = (size_t)1 << log_n;
    fp acc = 0;
    trace[0] = 0; trace[n] = 0;
    for (size_t q = 1; q < n; q++) {
        const size_t pos = n - 1 - q;
        const fp bit = ((words[pos / 64] >> (pos % 64)) & 1) ? FP_ONE : 0;
        acc = fp_add(fp_dbl(acc), bit);
        trace[q] = bit;
        trace[n + q] = acc;
    }
    return acc;
}
/* src/range/air.rs:91-98: enforce_double_and_add_step(result, cur, next, 1, 0, ONE) */
void cso_range_evaluate_transition(const uint64_t *cur, const uint64_t *next, const uint64_t *pv, uint64_t *res) {
    (void)pv;
    res[0] = res[1] = 0;
    field_enforce_double_and_add(res, cur, next, 1, 0, FP_ONE);
}

/* ---- Rescue hash-chain AIR (benches/rescue.rs) --------------------------------------------------- */
int cso_rescue_chain_build_trace(const uint64_t *seed7, uint32_t iterations, uint64_t *trace /*[14][8*iterations]*/) {
    const size_t n = (size_t)iterations * 8;
    fp st[14] = {0};
    memcpy(st, seed7, 7 * sizeof(fp));
    for (int c = 0; c < 14; c++) trace[(size_t)c * n] = st[c];
    for (size_t step = 0; step + 1 < n; step++) { /* :299-318 */
        if (step % 8 < 7) rescue_apply_round(st, step);
        else for (int i = 7; i < 14; i++) st[i] = 0;
        for (int c = 0; c < 14; c++) trace[(size_t)c * n + step + 1] = st[c];
    }
    return 0;
}
/* the off-circuit chain of benches/rescue.rs:104-123.  NOTE (reference quirk): it merges [values || result], so
 * from the second link on it differs from what the trace computes ([result || 0]); the bench only calls prove(). */
void cso_rescue_compute_hash_chain(const uint64_t *seed7, uint32_t length, uint64_t *out7) {
    fp values[7], result[7] = {0};
    memcpy(values, seed7, sizeof values);
    for (uint32_t i = 0; i < length; i++) { rescue_merge(values, result, result); memcpy(values, result, sizeof values); }
    memcpy(out7, result, sizeof result);
}
/* evaluate_transition :200-222; periodic = [cycle mask, 28 ark] */
void cso_rescue_chain_evaluate_transition(const uint64_t *cur, const uint64_t *next, const uint64_t *pv, uint64_t *res) {
    memset(res, 0, 14 * sizeof(fp));
    fp hash_flag = pv[0], copy_flag = c_not(pv[0]);
    rescue_enforce_round(res, cur, next, pv + 1, hash_flag);
    for (int i = 0; i < 7; i++) agg(res, i, copy_flag, c_are_equal(cur[i], next[i]));
    for (int i = 0; i < 7; i++) agg(res, 7 + i, copy_flag, next[7 + i]);
}
void cso_rescue_chain_periodic_columns(uint64_t *out /*[29][8]*/) {
    for (int i = 0; i < 8; i++) out[i] = i < 7 ? FP_ONE : 0;
    for (int j = 0; j < 28; j++) for (int i = 0; i < 8; i++) out[(1 + j) * 8 + i] = CS_ARK_MONT[i * 28 + j];
}

/* ---- SchnorrAir (src/schnorr/air.rs:41-300, src/schnorr/prover.rs:21-67) ------------------------------ */
void cso_schnorr_step(size_t step, const uint64_t *msg, const uint64_t *pkey, const uint8_t *s_bytes, const uint8_t *h_bytes, uint64_t *st);
void cso_schnorr_constraints(uint64_t *res, const uint64_t *cur, const uint64_t *next, const uint64_t *ark, uint64_t doubling_flag, uint64_t addition_flag,
                             const uint64_t *digest_flags, const uint64_t *pkey, uint64_t final_add_flag, uint64_t hash_flag, uint64_t copy_hash_flag,
                             const uint64_t *internal_inputs);
void cso_sign_message(uint64_t *rng_state, const uint64_t *msg28, uint64_t sk, uint64_t *rx6, uint8_t *s32);

/* messages: [n][28] = pkey(12) | 16 elements; trace: [56][512 n] */
int cso_schnorr_build_trace(uint32_t n_sig, const uint64_t *messages, const uint64_t *sig_rx, const uint8_t *sig_s, uint64_t *trace) {
    const size_t n = (size_t)n_sig * 512;
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t t = 0; t < n_sig; t++) {
        const fp *msg = messages + 28 * t;
        fp h[7], st[56];
        uint8_t h_bytes[32];
        cso_schnorr_hash_message(sig_rx + 6 * t, msg, h); /* build_sig_info, src/schnorr/trace.rs:127-142 */
        for (int i = 0; i < 4; i++) { uint64_t v = fp_to_u64(h[i]); for (int b = 0; b < 8; b++) h_bytes[8 * i + b] = v >> (8 * b); }
        memset(st, 0, sizeof st); /* init_sig_verification_state :18-30 */
        st[6] = FP_ONE; st[25] = FP_ONE;
        memcpy(st + 42, sig_rx + 6 * t, 6 * sizeof(fp));
        for (int c = 0; c < 56; c++) trace[(size_t)c * n + 512 * t] = st[c];
        for (size_t step = 0; step < 511; step++) {
            cso_schnorr_step(step, msg, msg, sig_s + 32 * t, h_bytes, st);
            for (int c = 0; c < 56; c++) trace[(size_t)c * n + 512 * t + step + 1] = st[c];
        }
    }
    return 0;
}
/* the public-input-dependent "periodic" columns of SchnorrAir (:228-290): pkey (12) constant over each 512-row
 * block, message chunks (7) at rows 8i+7 (i < 4) of each block; [19][512 n] */
void cso_schnorr_aux_columns(uint32_t n_sig, const uint64_t *messages, uint64_t *out) {
    const size_t n = (size_t)n_sig * 512;
    memset(out, 0, 19 * n * sizeof(fp));
    for (size_t t = 0; t < n_sig; t++) {
        const fp *msg = messages + 28 * t;
        for (int j = 0; j < 12; j++) for (int i = 0; i < 512; i++) out[(size_t)j * n + 512 * t + i] = msg[j];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 7; j++) out[(size_t)(12 + j) * n + 512 * t + 8 * i + 7] = msg[j + 7 * i];
    }
}
/* the input-independent periodic columns: 8 masks of src/schnorr/air.rs:335-391 then the 28 round constants; [36][512] */
void cso_schnorr_mask_columns(uint64_t *out) {
    memset(out, 0, 36 * 512 * sizeof(fp));
#define COL(c) (out + (size_t)(c) * 512)
    for (int i = 0; i < 511; i++) COL(0)[i] = FP_ONE;
    for (int i = 0; i < 510; i++) { COL(1)[i] = FP_ONE; COL(2)[i] = (i % 2 == 0) ? FP_ONE : 0; }
    const int lo[4] = {0, 126, 254, 382}, hi[4] = {126, 254, 382, 510};
    for (int k = 0; k < 4; k++) for (int i = lo[k]; i < hi[k]; i++) COL(3 + k)[i] = FP_ONE;
    for (int i = 0; i < 40; i++) COL(7)[i] = (i % 8) != 7 ? FP_ONE : 0;
    for (int j = 0; j < 28; j++) for (int i = 0; i < 512; i++) COL(8 + j)[i] = CS_ARK_MONT[(i % 8) * 28 + j];
#undef COL
}
/* SchnorrAir::evaluate_transition (:68-109) over LDE cosets: lde [nk][56][n], aux [nk][19][n] (LDE of the aux columns),
 * ptab [b][36][512]; out [nk][56][n] */
void cso_schnorr_evaluate_transitions(const uint64_t *lde, const uint64_t *aux, const uint64_t *ptab, uint64_t *out, unsigned log_n, uint32_t k0, uint32_t nk) {
    const size_t n = (size_t)1 << log_n;
#pragma omp parallel for schedule(static) collapse(2)
    for (uint32_t k = k0; k < k0 + nk; k++)
        for (size_t j = 0; j < n; j++) {
            fp cur[56], next[56], m[36], pk[12], inp[7], res[56];
            for (int c = 0; c < 56; c++) { cur[c] = lde[((size_t)(k - k0) * 56 + c) * n + j]; next[c] = lde[((size_t)(k - k0) * 56 + c) * n + (j + 1) % n]; }
            for (int c = 0; c < 36; c++) m[c] = ptab[((size_t)k * 36 + c) * 512 + j % 512];
            for (int c = 0; c < 12; c++) pk[c] = aux[((size_t)(k - k0) * 19 + c) * n + j];
            for (int c = 0; c < 7; c++) inp[c] = aux[((size_t)(k - k0) * 19 + 12 + c) * n + j];
            memset(res, 0, sizeof res);
            fp copy_hash = fp_mul(c_not(m[7]), m[0]), final_add = fp_mul(c_not(m[1]), m[0]), addition = fp_mul(c_not(m[2]), m[1]);
            cso_schnorr_constraints(res, cur, next, m + 8, m[2], addition, m + 3, pk, final_add, m[7], copy_hash, inp);
            for (int i = 0; i < 56; i++) out[((size_t)(k - k0) * 56 + i) * n + j] = res[i];
        }
}
/* the same at one point (what a verifier evaluates on the out-of-domain frame): m = the 36 mask / round-constant values, pk / inp =
 * values of the public-input columns */
void cso_schnorr_evaluate_transition_at(const uint64_t *cur, const uint64_t *next, const uint64_t *m, const uint64_t *pk, const uint64_t *inp, uint64_t *res) {
    memset(res, 0, 56 * sizeof(fp));
    fp copy_hash = fp_mul(c_not(m[7]), m[0]), final_add = fp_mul(c_not(m[1]), m[0]), addition = fp_mul(c_not(m[2]), m[1]);
    cso_schnorr_constraints(res, cur, next, m + 8, m[2], addition, m + 3, pk, final_add, m[7], copy_hash, inp);
}
/* schnorr::transition_constraint_degrees(num_sig, 512), src/schnorr/air.rs:533-585 */
void cso_schnorr_constraint_degrees(uint32_t n_sig, uint32_t *base, uint32_t *cycles) {
    const uint32_t bit_degree = n_sig == 1 ? 3 : 5;
    for (int i = 0; i < 6; i++) { base[i] = 5; cycles[i] = 2; }
    for (int i = 6; i < 18; i++) { base[i] = 4; cycles[i] = 2; }
    base[18] = 2; cycles[18] = 1;
    for (int i = 19; i < 37; i++) { base[i] = bit_degree; cycles[i] = 2; }
    base[37] = 2; cycles[37] = 1;
    for (int i = 38; i < 42; i++) { base[i] = 1; cycles[i] = 2; }
    for (int i = 42; i < 56; i++) { base[i] = 3; cycles[i] = 1; }
}
/* deterministic messages and signatures (counterpart of SchnorrExample::new, src/schnorr/mod.rs:86-141) */
int cso_schnorr_witness_generate(uint32_t n_sig, uint64_t seed, uint64_t *messages, uint64_t *sig_rx, uint8_t *sig_s) {
    uint64_t rng = seed;
    for (uint32_t t = 0; t < n_sig; t++) {
        fp *msg = messages + 28 * t;
        uint64_t z = (rng += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31;
        uint64_t sk = 1 + z % 8;
        cso_ecc_scalar_mul_affine(&sk, 1, CS_GENERATOR_MONT, msg);
        for (int i = 12; i < 28; i++) {
            z = (rng += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31;
            msg[i] = fp_from_u64(z);
        }
        uint64_t r2 = rng ^ 0xA5A5A5A5DEADBEEFULL;
        cso_sign_message(&r2, msg, sk, sig_rx + 6 * t, sig_s + 32 * t);
    }
    return 0;
}

/* ---- generic driver --------------------------------------------------------------------------------
 * evals: [nk][nc][n] transition evaluations on the LDE cosets k0.. (coset-major), lde: [nk][width][n].
 * Degree (base_i; cyc_i cycles of length cycle_len).  Single-step assertions only: (reg, last?, value).
 * Only cosets with k % (lde_b / ce_b) == 0 belong to the constraint-evaluation domain; others get 0. */
void cso_air_combine(const cso_air_desc *d, const uint64_t *lde, const uint64_t *evals, const uint64_t *t_alpha, const uint64_t *t_beta,
                     const uint64_t *b_alpha, const uint64_t *b_beta, uint64_t *out, unsigned log_n, unsigned log_b, uint32_t k0, uint32_t nk,
                     int all_cosets, const uint64_t *avals /* [nk][n_avals][n] LDE of the sequence-value polynomials, or NULL */, uint32_t n_avals) {
    const size_t n = (size_t)1 << log_n;
    const unsigned log_ce = d->log_ce_blowup, stride = 1u << (log_b - log_ce);
    const uint64_t ce = (uint64_t)n << log_ce;
    fp g = fp_from_u64(FP_LDE_OFFSET_CAN), wbn = fp_root_of_unity(log_n + log_b), wn = fp_root_of_unity(log_n);
#pragma omp parallel for schedule(static) collapse(2)
    for (uint32_t k = k0; k < k0 + nk; k++)
        for (size_t j = 0; j < n; j++) {
            fp *o = out + (size_t)(k - k0) * n + j;
            if ((k % stride) && !all_cosets) { *o = 0; continue; } /* all_cosets: test hook, evaluates the same rational function off-domain */
            fp x = fp_mul(fp_mul(g, fp_pow(wbn, k)), fp_pow(wn, j)), acc = 0;
            for (uint32_t i = 0; i < d->n_constraints; i++) {
                uint64_t ev_deg = d->base[i] * (n - 1) + (d->cycle_len ? d->cycles[i] * (n / d->cycle_len) * (d->cycle_len - 1) : 0);
                uint64_t adj = CSTARK_CONV_TRANSITION_ADJUSTMENT(ce, n, ev_deg);
                fp c = evals[((size_t)(k - k0) * d->n_constraints + i) * n + j];
                acc = fp_add(acc, fp_mul(c, fp_add(t_alpha[i], fp_mul(t_beta[i], fp_pow(x, adj)))));
            }
            acc = fp_mul(acc, fp_inv(fp_mul(fp_sub(fp_pow(x, n), FP_ONE), fp_inv(fp_sub(x, fp_inv(wn))))));
            /* boundary constraints: (T_reg(x) - c_a(x)) (alpha_a + beta_a x^(ce - 1 + m - (n - 1))) / (x^m - w^(first * m)),
             * m = number of asserted steps; assertions with the same divisor could be summed first (same value) */
            for (uint32_t a = 0; a < d->n_assertions; a++) {
                uint64_t first = d->a_stride ? d->a_first[a] : (d->a_last[a] ? n - 1 : 0);
                uint64_t m = (d->a_stride && d->a_stride[a]) ? n / d->a_stride[a] : 1;
                fp tv = lde[((size_t)(k - k0) * d->width + d->a_reg[a]) * n + j];
                fp cv = (d->a_seq && d->a_seq[a] >= 0) ? avals[((size_t)(k - k0) * n_avals + d->a_seq[a]) * n + j] : d->a_value[a];
                fp xb = fp_pow(x, CSTARK_CONV_BOUNDARY_ADJUSTMENT(ce, n, m));
                fp z = fp_sub(fp_pow(x, m), fp_pow(wn, (first * m) % n));
                acc = fp_add(acc, fp_mul(fp_mul(fp_sub(tv, cv), fp_add(b_alpha[a], fp_mul(b_beta[a], xb))), fp_inv(z)));
            }
            *o = acc;
        }
}
/* Coefficients (in x, zero-padded to n) of the value polynomials of sequence assertions: values [n_seq][m] are asserted at
 * steps first_step + k * (n / m); c(x) = P(x * w^-first_step) with P = interpolation of the values over the m-th roots. */
void cso_sequence_value_polys(const uint64_t *values, uint32_t n_seq, uint32_t m, uint32_t first_step, unsigned log_n, uint64_t *out) {
    const size_t n = (size_t)1 << log_n;
    unsigned log_m = 0;
    while ((1u << log_m) < m) log_m++;
    fp winv = fp_inv(fp_root_of_unity(log_n));
    fp off = fp_pow(winv, first_step);
    memset(out, 0, (size_t)n_seq * n * sizeof(fp));
    for (uint32_t s = 0; s < n_seq; s++) {
        fp *o = out + (size_t)s * n;
        memcpy(o, values + (size_t)s * m, m * sizeof(fp));
        if (m > 1) cso_intt(o, log_m);
        fp sc = FP_ONE;
        for (uint32_t t = 0; t < m; t++) { o[t] = fp_mul(o[t], sc); sc = fp_mul(sc, off); }
    }
}

typedef void (*transition_fn)(const uint64_t *, const uint64_t *, const uint64_t *, uint64_t *);
/* all transition constraints of a small AIR over LDE cosets; ptab: [b][n_periodic][cycle_len] (or NULL) */
void cso_air_evaluate_transitions(int air, const uint64_t *lde, const uint64_t *ptab, uint64_t *out, uint32_t width, uint32_t nc, uint32_t np,
                                  uint32_t cycle_len, unsigned log_n, uint32_t k0, uint32_t nk) {
    const size_t n = (size_t)1 << log_n;
    transition_fn fn = air == CSTARK_AIR_MERKLE_UPDATE ? cso_merkle_evaluate_transition
                       : air == CSTARK_AIR_RANGE ? cso_range_evaluate_transition : cso_rescue_chain_evaluate_transition;
#pragma omp parallel for schedule(static) collapse(2)
    for (uint32_t k = k0; k < k0 + nk; k++)
        for (size_t j = 0; j < n; j++) {
            fp cur[128], next[128], pv[64], res[128];
            for (uint32_t c = 0; c < width; c++) {
                cur[c] = lde[((size_t)(k - k0) * width + c) * n + j];
                next[c] = lde[((size_t)(k - k0) * width + c) * n + (j + 1) % n];
            }
            for (uint32_t c = 0; c < np; c++) pv[c] = ptab[((size_t)k * np + c) * cycle_len + j % cycle_len];
            fn(cur, next, pv, res);
            for (uint32_t i = 0; i < nc; i++) out[((size_t)(k - k0) * nc + i) * n + j] = res[i];
        }
}
/* periodic table of arbitrary columns: cols [np][cycle_len] -> out [b][np][cycle_len] over the LDE cosets */
void cso_periodic_table(const uint64_t *cols, uint32_t np, unsigned log_cycle, unsigned log_n, unsigned log_b, uint64_t *out) {
    size_t C = (size_t)1 << log_cycle, b = (size_t)1 << log_b, n = (size_t)1 << log_n;
    fp *co = malloc(np * C * sizeof(fp));
    memcpy(co, cols, np * C * sizeof(fp));
    cso_interpolate_columns(co, np, log_cycle);
    fp g = fp_from_u64(FP_LDE_OFFSET_CAN), wbn = fp_root_of_unity(log_n + log_b);
    for (size_t k = 0; k < b; k++) {
        fp off = fp_pow(fp_mul(g, fp_pow(wbn, k)), n / C);
        for (uint32_t c = 0; c < np; c++) {
            fp *o = out + (k * np + c) * C, s = FP_ONE;
            for (size_t m = 0; m < C; m++) { o[m] = fp_mul(co[c * C + m], s); s = fp_mul(s, off); }
            cso_ntt(o, log_cycle);
        }
    }
    free(co);
}
