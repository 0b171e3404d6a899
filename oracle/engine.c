/* ORACLE (test infrastructure, not product code).
 *
 * CPU restatement of the ENGINE stages the reference obtains from its un-vendored dependency
 * (winterfell fork, Cargo.toml:20 @8e37310; invoked through `prover.prove(trace)`, src/lib.rs:140):
 * polynomial interpolation / low-degree extension, Blake3-256 row hashing (HashFunction::Blake3_256 is
 * selected at src/lib.rs:82), the Merkle commitment, and the constraint-evaluation driver that calls
 * back into Air::evaluate_transition (src/air.rs:114-173) and get_assertions (:175-184).
 *
 * The source of that dependency is not on this machine, so these functions restate the PUBLISHED
 * algorithms [UPSTREAM-RECALL winterfell v0.3]:
 *   - trace domain <w_n>, LDE domain g*<w_bn> with g = field generator, evaluations in natural order;
 *   - leaf i = Blake3(row i of the LDE, elements as little-endian bytes of their memory form);
 *   - node = Blake3(left || right), binary tree, nodes[1] = root;
 *   - transition constraints grouped by evaluation degree, merged as sum_i (alpha_i + beta_i x^adj) C_i(x),
 *     adj = (ce_domain_size - 1 + n - 1) - evaluation_degree, divided by (x^n - 1)/(x - w^(n-1));
 *   - boundary constraints (T_r(x) - v)(alpha + beta x^(ce_size - n + 1)) / (x - w^step).
 * PARITY: BLAKE3 is pinned by the official test vectors (tests/test_oracle_engine.py); the NTT by the
 * naive DFT; every convention above that cannot be checked without the fork is "parity unpinned".
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "fp.h"

/* ---- NTT ---------------------------------------------------------------------------------------- */
static void bit_reverse(fp *a, size_t n) {
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { fp t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}
/* in-place radix-2 DIT, natural in / natural out: a[k] <- sum_j a[j] w^(jk) */
static void ntt_core(fp *a, unsigned log_n, fp w) {
    size_t n = (size_t)1 << log_n;
    bit_reverse(a, n);
    for (unsigned s = 1; s <= log_n; s++) {
        size_t m = (size_t)1 << s, half = m >> 1;
        fp wm = w;
        for (unsigned i = s; i < log_n; i++) wm = fp_sqr(wm); /* w^(n/m) */
        for (size_t k = 0; k < n; k += m) {
            fp tw = FP_ONE;
            for (size_t j = 0; j < half; j++) {
                fp u = a[k + j], v = fp_mul(a[k + j + half], tw);
                a[k + j] = fp_add(u, v);
                a[k + j + half] = fp_sub(u, v);
                tw = fp_mul(tw, wm);
            }
        }
    }
}
void cso_ntt(uint64_t *a, unsigned log_n) { ntt_core(a, log_n, fp_root_of_unity(log_n)); }
void cso_intt(uint64_t *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    ntt_core(a, log_n, fp_inv(fp_root_of_unity(log_n)));
    fp ninv = fp_inv(fp_from_u64(n));
    for (size_t i = 0; i < n; i++) a[i] = fp_mul(a[i], ninv);
}
/* O(n^2) reference transform for the tests */
void cso_dft_naive(const uint64_t *a, uint64_t *out, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    fp w = fp_root_of_unity(log_n);
    for (size_t k = 0; k < n; k++) {
        fp wk = fp_pow(w, k), x = FP_ONE, acc = 0;
        for (size_t j = 0; j < n; j++) { acc = fp_add(acc, fp_mul(a[j], x)); x = fp_mul(x, wk); }
        out[k] = acc;
    }
}
/* columns of evaluations over <w_n> -> coefficients (engine: interpolate_poly) */
void cso_interpolate_columns(uint64_t *cols, uint32_t width, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t c = 0; c < width; c++) cso_intt(cols + (size_t)c * n, log_n);
}
/* coefficients -> evaluations over offset * w_{bn}^k * <w_n>, coset-major (see include/cstark.h) */
void cso_lde_columns(const uint64_t *coeffs, uint64_t *lde, uint32_t width, unsigned log_n, unsigned log_b, uint64_t offset,
                     uint32_t k0, uint32_t nk) {
    size_t n = (size_t)1 << log_n;
    fp wbn = fp_root_of_unity(log_n + log_b);
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (uint32_t k = k0; k < k0 + nk; k++)
        for (uint32_t c = 0; c < width; c++) {
            fp shift = fp_mul(offset, fp_pow(wbn, k)), s = FP_ONE;
            fp *out = lde + ((size_t)(k - k0) * width + c) * n;
            const fp *in = coeffs + (size_t)c * n;
            for (size_t m = 0; m < n; m++) { out[m] = fp_mul(in[m], s); s = fp_mul(s, shift); }
            cso_ntt(out, log_n);
        }
}
uint64_t cso_fp_generator(void) { return fp_from_u64(FP_LDE_OFFSET_CAN); } /* used as the offset of the evaluation domains */
/* the engine conventions of include/cstark_conventions.h for the Python half of the oracle (verifier.py, prover.py) */
void cso_conventions(int64_t *out /*[16]*/) {
    out[0] = CSTARK_CONV_FIELD_GENERATOR; out[1] = CSTARK_CONV_TWO_ADIC_ROOT_EXP; out[2] = CSTARK_CONV_LDE_OFFSET;
    out[3] = CSTARK_CONV_HASHED_ELEMENT_BYTES_MONTGOMERY; out[4] = CSTARK_CONV_TRANSITION_EXEMPTIONS;
    out[5] = CSTARK_CONV_COIN_FIRST_COUNTER; out[6] = CSTARK_CONV_COIN_REJECT_ABOVE_P; out[7] = CSTARK_CONV_QUERY_DEDUP;
    out[8] = CSTARK_CONV_DEEP_DRAWS_PER_REGISTER;
    out[9] = CSTARK_CONV_E2_C0; out[10] = CSTARK_CONV_E2_C1; out[11] = CSTARK_CONV_E3_C0; out[12] = CSTARK_CONV_E3_C1; out[13] = CSTARK_CONV_E3_C2;
    out[14] = 0; out[15] = 0;
}
/* degree adjustments (CSTARK_CONV_*_ADJUSTMENT) for the Python half */
uint64_t cso_transition_adjustment(uint64_t ce_size, uint64_t n, uint64_t eval_degree) { return CSTARK_CONV_TRANSITION_ADJUSTMENT(ce_size, n, eval_degree); }
uint64_t cso_boundary_adjustment(uint64_t ce_size, uint64_t n, uint64_t m) { return CSTARK_CONV_BOUNDARY_ADJUSTMENT(ce_size, n, m); }

/* ---- BLAKE3 (public specification) -------------------------------------------------------------- */
static const uint32_t B3_IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
static const uint8_t B3_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum { B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_PARENT = 4, B3_ROOT = 8 };
static inline uint32_t rotr32(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }
static inline void b3_g(uint32_t *s, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
    s[a] = s[a] + s[b] + mx; s[d] = rotr32(s[d] ^ s[a], 16);
    s[c] = s[c] + s[d];      s[b] = rotr32(s[b] ^ s[c], 12);
    s[a] = s[a] + s[b] + my; s[d] = rotr32(s[d] ^ s[a], 8);
    s[c] = s[c] + s[d];      s[b] = rotr32(s[b] ^ s[c], 7);
}
static void b3_compress(const uint32_t cv[8], const uint8_t block[64], uint64_t counter, uint32_t block_len, uint32_t flags, uint32_t out[16]) {
    uint32_t m[16], s[16];
    for (int i = 0; i < 16; i++) m[i] = (uint32_t)block[4 * i] | (uint32_t)block[4 * i + 1] << 8 | (uint32_t)block[4 * i + 2] << 16 | (uint32_t)block[4 * i + 3] << 24;
    for (int i = 0; i < 8; i++) s[i] = cv[i];
    for (int i = 0; i < 4; i++) s[8 + i] = B3_IV[i];
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = block_len; s[15] = flags;
    for (int r = 0; r < 7; r++) {
        b3_g(s, 0, 4, 8, 12, m[0], m[1]);  b3_g(s, 1, 5, 9, 13, m[2], m[3]);
        b3_g(s, 2, 6, 10, 14, m[4], m[5]); b3_g(s, 3, 7, 11, 15, m[6], m[7]);
        b3_g(s, 0, 5, 10, 15, m[8], m[9]); b3_g(s, 1, 6, 11, 12, m[10], m[11]);
        b3_g(s, 2, 7, 8, 13, m[12], m[13]); b3_g(s, 3, 4, 9, 14, m[14], m[15]);
        uint32_t t[16];
        for (int i = 0; i < 16; i++) t[i] = m[B3_PERM[i]];
        memcpy(m, t, sizeof m);
    }
    for (int i = 0; i < 8; i++) { out[i] = s[i] ^ s[i + 8]; out[i + 8] = s[i + 8] ^ cv[i]; }
}
/* chaining value of one chunk (<= 1024 bytes); `root` sets the ROOT flag on its last block */
static void b3_chunk_cv(const uint8_t *in, size_t len, uint64_t chunk_index, int root, uint32_t cv_out[8]) {
    uint32_t cv[8], out[16];
    memcpy(cv, B3_IV, sizeof cv);
    size_t nblocks = len == 0 ? 1 : (len + 63) / 64;
    for (size_t b = 0; b < nblocks; b++) {
        uint8_t block[64] = {0};
        size_t blen = (b + 1 < nblocks) ? 64 : len - 64 * b;
        memcpy(block, in + 64 * b, blen);
        uint32_t flags = (b == 0 ? B3_CHUNK_START : 0) | (b + 1 == nblocks ? B3_CHUNK_END | (root ? B3_ROOT : 0) : 0);
        b3_compress(cv, block, chunk_index, (uint32_t)blen, flags, out);
        memcpy(cv, out, sizeof cv);
    }
    memcpy(cv_out, cv, sizeof cv);
}
static void b3_parent_cv(const uint32_t l[8], const uint32_t r[8], int root, uint32_t cv_out[8]) {
    uint8_t block[64];
    uint32_t out[16];
    for (int i = 0; i < 8; i++) for (int b = 0; b < 4; b++) { block[4 * i + b] = l[i] >> (8 * b); block[32 + 4 * i + b] = r[i] >> (8 * b); }
    b3_compress(B3_IV, block, 0, 64, B3_PARENT | (root ? B3_ROOT : 0), out);
    memcpy(cv_out, out, 32);
}
/* subtree of `len` bytes starting at chunk index `chunk0`; left subtree = largest power-of-two number of chunks */
static void b3_subtree(const uint8_t *in, size_t len, uint64_t chunk0, int root, uint32_t cv_out[8]) {
    if (len <= 1024) { b3_chunk_cv(in, len, chunk0, root, cv_out); return; }
    size_t chunks = (len + 1023) / 1024, left = 1;
    while (left * 2 < chunks) left *= 2;
    uint32_t l[8], r[8];
    b3_subtree(in, left * 1024, chunk0, 0, l);
    b3_subtree(in + left * 1024, len - left * 1024, chunk0 + left, 0, r);
    b3_parent_cv(l, r, root, cv_out);
}
void cso_blake3(const uint8_t *in, size_t len, uint8_t out[32]) {
    uint32_t cv[8];
    b3_subtree(in, len, 0, 1, cv);
    for (int i = 0; i < 8; i++) for (int b = 0; b < 4; b++) out[4 * i + b] = cv[i] >> (8 * b);
}

/* row hashing (engine: hash_elements over each LDE row) -> leaf i = b*j + k */
void cso_hash_rows(const uint64_t *lde, uint8_t *leaves, uint32_t width, unsigned log_n, unsigned log_b, uint32_t k0, uint32_t nk) {
    size_t n = (size_t)1 << log_n, b = (size_t)1 << log_b;
#pragma omp parallel for schedule(static) collapse(2)
    for (uint32_t k = k0; k < k0 + nk; k++)
        for (size_t j = 0; j < n; j++) {
            uint8_t buf[8 * 256];
            for (uint32_t c = 0; c < width; c++) {
                uint64_t v = lde[((size_t)(k - k0) * width + c) * n + j];
#if !CSTARK_CONV_HASHED_ELEMENT_BYTES_MONTGOMERY
                v = fp_to_u64(v); /* canonical little-endian bytes */
#endif
                for (int t = 0; t < 8; t++) buf[8 * c + t] = v >> (8 * t);
            }
            cso_blake3(buf, 8 * (size_t)width, leaves + 32 * (b * j + k));
        }
}
/* nodes: 2*L digests, leaves at [L, 2L); nodes[i] = Blake3(nodes[2i] || nodes[2i+1]); nodes[0] zero */
void cso_merkle_build(uint8_t *nodes, unsigned log_leaves) {
    size_t L = (size_t)1 << log_leaves;
    memset(nodes, 0, 32);
    for (size_t lvl = L >> 1; lvl >= 1; lvl >>= 1) {
#pragma omp parallel for schedule(static)
        for (size_t i = lvl; i < 2 * lvl; i++) cso_blake3(nodes + 64 * i, 64, nodes + 32 * i);
    }
}

/* ---- constraint-evaluation driver ---------------------------------------------------------------- */
/* periodic table: out[k][48][1024] = periodic polynomial c evaluated at (g w_{bn}^k)^(n/1024) * w_1024^jj */
void cso_tx_periodic_table(unsigned depth, unsigned log_n, unsigned log_b, uint64_t *out) {
    const unsigned LOGC = 10;
    size_t C = (size_t)1 << LOGC, b = (size_t)1 << log_b, n = (size_t)1 << log_n;
    fp *cols = malloc(48 * C * sizeof(fp));
    cso_tx_periodic_columns(depth, cols);
    cso_interpolate_columns(cols, 48, LOGC);
    fp g = fp_from_u64(FP_LDE_OFFSET_CAN), wbn = fp_root_of_unity(log_n + log_b);
    for (size_t k = 0; k < b; k++) {
        fp off = fp_pow(fp_mul(g, fp_pow(wbn, k)), n / C), s;
        for (int c = 0; c < 48; c++) {
            fp *o = out + (k * 48 + c) * C;
            s = FP_ONE;
            for (size_t m = 0; m < C; m++) { o[m] = fp_mul(cols[c * C + m], s); s = fp_mul(s, off); }
            cso_ntt(o, LOGC);
        }
    }
    free(cols);
}

/* degree adjustment of each transition constraint: (ce_size - 1 + n - 1) - evaluation_degree */
void cso_tx_degree_adjustments(unsigned log_n, unsigned log_b, uint64_t *adj /*115*/) {
    uint32_t base[115], cyc[115];
    cso_tx_constraint_degrees(base, cyc);
    uint64_t n = (uint64_t)1 << log_n, ce = n << log_b;
    for (int i = 0; i < 115; i++) {
        uint64_t ev = base[i] * (n - 1) + cyc[i] * (n / 1024) * 1023;
        adj[i] = CSTARK_CONV_TRANSITION_ADJUSTMENT(ce, n, ev);
    }
}

/* all 115 transition evaluations over cosets [k0,k0+nk): out[(k-k0)][115][n] */
void cso_tx_evaluate_transitions(const uint64_t *lde, uint64_t *out, unsigned depth, unsigned log_n, unsigned log_b, uint32_t k0, uint32_t nk) {
    size_t n = (size_t)1 << log_n, b = (size_t)1 << log_b;
    fp *ptab = malloc(b * 48 * 1024 * sizeof(fp));
    cso_tx_periodic_table(depth, log_n, log_b, ptab);
#pragma omp parallel for schedule(static) collapse(2)
    for (uint32_t k = k0; k < k0 + nk; k++)
        for (size_t j = 0; j < n; j++) {
            fp cur[94], next[94], pv[48], res[115];
            const fp *base = lde + (size_t)(k - k0) * 94 * n;
            for (int c = 0; c < 94; c++) { cur[c] = base[c * n + j]; next[c] = base[c * n + (j + 1) % n]; } /* row i + b of the LDE */
            for (int c = 0; c < 48; c++) pv[c] = ptab[((size_t)k * 48 + c) * 1024 + j % 1024];
            cso_tx_evaluate_transition(cur, next, pv, res);
            for (int i = 0; i < 115; i++) out[((size_t)(k - k0) * 115 + i) * n + j] = res[i];
        }
    free(ptab);
}

/* combined constraint evaluations (transition + boundary, already divided by their divisors) */
void cso_tx_evaluate_constraints(const uint64_t *lde, const cstark_tx_coeffs *cf, const uint64_t pub_inputs[4], uint64_t *out,
                                 unsigned depth, unsigned log_n, unsigned log_b, uint32_t k0, uint32_t nk) {
    size_t n = (size_t)1 << log_n, b = (size_t)1 << log_b;
    fp *ptab = malloc(b * 48 * 1024 * sizeof(fp));
    cso_tx_periodic_table(depth, log_n, log_b, ptab);
    uint64_t adj[115];
    cso_tx_degree_adjustments(log_n, log_b, adj);
    const uint64_t badj = CSTARK_CONV_BOUNDARY_ADJUSTMENT(n << log_b, n, 1);
    fp g = fp_from_u64(FP_LDE_OFFSET_CAN), wbn = fp_root_of_unity(log_n + log_b), wn = fp_root_of_unity(log_n);
    fp w_last = fp_inv(wn); /* w^(n-1) */
#pragma omp parallel for schedule(static) collapse(2)
    for (uint32_t k = k0; k < k0 + nk; k++)
        for (size_t j = 0; j < n; j++) {
            fp cur[94], next[94], pv[48], res[115];
            const fp *base = lde + (size_t)(k - k0) * 94 * n;
            for (int c = 0; c < 94; c++) { cur[c] = base[c * n + j]; next[c] = base[c * n + (j + 1) % n]; }
            for (int c = 0; c < 48; c++) pv[c] = ptab[((size_t)k * 48 + c) * 1024 + j % 1024];
            cso_tx_evaluate_transition(cur, next, pv, res);
            fp x = fp_mul(fp_mul(g, fp_pow(wbn, k)), fp_pow(wn, j));
            fp acc = 0;
            for (int i = 0; i < 115; i++)
                acc = fp_add(acc, fp_mul(res[i], fp_add(cf->t_alpha[i], fp_mul(cf->t_beta[i], fp_pow(x, adj[i])))));
            /* divisor (x^n - 1) / (x - w^(n-1)) */
            fp zt = fp_mul(fp_sub(fp_pow(x, n), FP_ONE), fp_inv(fp_sub(x, w_last)));
            acc = fp_mul(acc, fp_inv(zt));
            /* boundary: registers 58,59 at the first step, 58,59 at the last step (src/air.rs:175-184) */
            fp xb = fp_pow(x, badj), first = 0, last = 0;
            for (int a = 0; a < 2; a++) {
                first = fp_add(first, fp_mul(fp_sub(cur[58 + a], pub_inputs[a]), fp_add(cf->b_alpha[a], fp_mul(cf->b_beta[a], xb))));
                last = fp_add(last, fp_mul(fp_sub(cur[58 + a], pub_inputs[2 + a]), fp_add(cf->b_alpha[2 + a], fp_mul(cf->b_beta[2 + a], xb))));
            }
            acc = fp_add(acc, fp_mul(first, fp_inv(fp_sub(x, FP_ONE))));
            acc = fp_add(acc, fp_mul(last, fp_inv(fp_sub(x, w_last))));
            out[(size_t)(k - k0) * n + j] = acc;
        }
    free(ptab);
}

/* ---- out-of-domain evaluation (what the verifier recomputes; used here to pin the driver) ---------- */
uint64_t cso_poly_eval(const uint64_t *co, size_t n, uint64_t x) {
    fp acc = 0;
    for (size_t i = n; i-- > 0;) acc = fp_add(fp_mul(acc, x), co[i]);
    return acc;
}
/* value of the combined constraint expression at an arbitrary point z from an evaluation frame (cur = T(z), next = T(z w)):
 * what the verifier computes from the out-of-domain frame (engine: evaluate_constraints on the OOD frame) */
uint64_t cso_tx_combined_from_frame(const uint64_t *cur, const uint64_t *next, const cstark_tx_coeffs *cf, const uint64_t pub_inputs[4],
                                    unsigned depth, unsigned log_n, unsigned log_b, uint64_t z) {
    size_t n = (size_t)1 << log_n;
    fp wn = fp_root_of_unity(log_n), w_last = fp_inv(wn);
    fp pv[48], res[115];
    fp *cols = malloc(48 * 1024 * sizeof(fp));
    cso_tx_periodic_columns(depth, cols);
    cso_interpolate_columns(cols, 48, 10);
    fp y = fp_pow(z, n / 1024);
    for (int c = 0; c < 48; c++) pv[c] = cso_poly_eval(cols + (size_t)c * 1024, 1024, y);
    free(cols);
    cso_tx_evaluate_transition(cur, next, pv, res);
    uint64_t adj[115];
    cso_tx_degree_adjustments(log_n, log_b, adj);
    fp acc = 0;
    for (int i = 0; i < 115; i++) acc = fp_add(acc, fp_mul(res[i], fp_add(cf->t_alpha[i], fp_mul(cf->t_beta[i], fp_pow(z, adj[i])))));
    acc = fp_mul(acc, fp_inv(fp_mul(fp_sub(fp_pow(z, n), FP_ONE), fp_inv(fp_sub(z, w_last)))));
    fp xb = fp_pow(z, CSTARK_CONV_BOUNDARY_ADJUSTMENT(n << log_b, n, 1)), first = 0, last = 0;
    for (int a = 0; a < 2; a++) {
        first = fp_add(first, fp_mul(fp_sub(cur[58 + a], pub_inputs[a]), fp_add(cf->b_alpha[a], fp_mul(cf->b_beta[a], xb))));
        last = fp_add(last, fp_mul(fp_sub(cur[58 + a], pub_inputs[2 + a]), fp_add(cf->b_alpha[2 + a], fp_mul(cf->b_beta[2 + a], xb))));
    }
    acc = fp_add(acc, fp_mul(first, fp_inv(fp_sub(z, FP_ONE))));
    return fp_add(acc, fp_mul(last, fp_inv(fp_sub(z, w_last))));
}
/* the same from the trace polynomials */
uint64_t cso_tx_combined_at(const uint64_t *trace_coeffs, const cstark_tx_coeffs *cf, const uint64_t pub_inputs[4],
                            unsigned depth, unsigned log_n, unsigned log_b, uint64_t z) {
    size_t n = (size_t)1 << log_n;
    fp cur[94], next[94];
    fp zn = fp_mul(z, fp_root_of_unity(log_n));
    for (int c = 0; c < 94; c++) {
        cur[c] = cso_poly_eval(trace_coeffs + (size_t)c * n, n, z);
        next[c] = cso_poly_eval(trace_coeffs + (size_t)c * n, n, zn);
    }
    return cso_tx_combined_from_frame(cur, next, cf, pub_inputs, depth, log_n, log_b, z);
}


/* ---- composition polynomial (engine: ConstraintEvaluationTable::into_poly, CompositionPoly, build_commitment) -------
 * [UPSTREAM-RECALL winterfell v0.3]  combined: [b][n] coset-major evaluations over g<w_{bn}> (b = ce blowup = LDE blowup).
 * H(x) = interpolate_poly_with_offset(evaluations, g); its b*n coefficients are split into b column polynomials
 * H(x) = sum_i x^i H_i(x^b)  (coefficient m goes to column m mod b at position m div b).
 * out_cols: [b][n] coefficients. */
void cso_composition_columns(const uint64_t *combined, uint64_t *out_cols, unsigned log_n, unsigned log_b) {
    const size_t n = (size_t)1 << log_n, b = (size_t)1 << log_b, N = n * b;
    fp *h = malloc(N * sizeof(fp));
    for (size_t k = 0; k < b; k++) for (size_t j = 0; j < n; j++) h[b * j + k] = combined[k * n + j]; /* natural order */
    cso_intt(h, log_n + log_b);
    fp ginv = fp_inv(fp_from_u64(FP_LDE_OFFSET_CAN)), s = FP_ONE;
    for (size_t m = 0; m < N; m++) { h[m] = fp_mul(h[m], s); s = fp_mul(s, ginv); }
    for (size_t m = 0; m < N; m++) out_cols[(m % b) * n + m / b] = h[m];
    free(h);
}


/* ---- out-of-domain frame and DEEP composition (engine steps 5 of SURVEY 3.1) [UPSTREAM-RECALL winterfell v0.3] ---------
 * DEEP(x) = [ sum_c alpha_c (T_c(x) - T_c(z)) / (x - z) + beta_c (T_c(x) - T_c(z w)) / (x - z w)
 *           + sum_i delta_i (H_i(x) - H_i(z^b)) / (x - z^b) ] * (deg_a + deg_b x)
 * over the LDE domain (coset-major).  trace_lde [nk][W][n], comp_lde [nk][nb][n]; ood_trace [2][W] = T(z), T(z w);
 * ood_comp [nb] = H_i(z^nb). */
void cso_deep_composition(const uint64_t *trace_lde, const uint64_t *comp_lde, uint32_t width, uint32_t nb, uint64_t z,
                          const uint64_t *ood_trace, const uint64_t *ood_comp, const uint64_t *alpha, const uint64_t *beta,
                          const uint64_t *delta, uint64_t deg_a, uint64_t deg_b, uint64_t *out, unsigned log_n, unsigned log_b,
                          uint32_t k0, uint32_t nk) {
    const size_t n = (size_t)1 << log_n;
    fp g = fp_from_u64(FP_LDE_OFFSET_CAN), wbn = fp_root_of_unity(log_n + log_b), wn = fp_root_of_unity(log_n);
    fp zw = fp_mul(z, wn), zb = fp_pow(z, nb);
#pragma omp parallel for schedule(static) collapse(2)
    for (uint32_t k = k0; k < k0 + nk; k++)
        for (size_t j = 0; j < n; j++) {
            fp x = fp_mul(fp_mul(g, fp_pow(wbn, k)), fp_pow(wn, j));
            fp i1 = fp_inv(fp_sub(x, z)), i2 = fp_inv(fp_sub(x, zw)), i3 = fp_inv(fp_sub(x, zb));
            fp s1 = 0, s2 = 0, s3 = 0;
            for (uint32_t c = 0; c < width; c++) {
                fp t = trace_lde[((size_t)(k - k0) * width + c) * n + j];
                s1 = fp_add(s1, fp_mul(alpha[c], fp_sub(t, ood_trace[c])));
                s2 = fp_add(s2, fp_mul(beta[c], fp_sub(t, ood_trace[width + c])));
            }
            for (uint32_t i = 0; i < nb; i++)
                s3 = fp_add(s3, fp_mul(delta[i], fp_sub(comp_lde[((size_t)(k - k0) * nb + i) * n + j], ood_comp[i])));
            fp acc = fp_add(fp_add(fp_mul(s1, i1), fp_mul(s2, i2)), fp_mul(s3, i3));
            out[(size_t)(k - k0) * n + j] = fp_mul(acc, fp_add(deg_a, fp_mul(deg_b, x)));
        }
}
/* values of `width` coefficient columns at `npts` points: out[p][c] */
void cso_evaluate_polys_at(const uint64_t *coeffs, uint32_t width, unsigned log_n, const uint64_t *points, uint32_t npts, uint64_t *out) {
    const size_t n = (size_t)1 << log_n;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (uint32_t p = 0; p < npts; p++)
        for (uint32_t c = 0; c < width; c++) out[(size_t)p * width + c] = cso_poly_eval(coeffs + (size_t)c * n, n, points[p]);
}


/* ---- FRI layer folding (folding factor f = 2^log_f: 4, 8, 16) [UPSTREAM-RECALL winterfell-fri v0.3 apply_drp] ----------------
 * evals: N evaluations of p over offset * <w_N> in natural order.  Output: N/f evaluations over offset^f * <w_{N/f}> of
 * p'(y) = sum_k alpha^k p_k(y) where p(x) = sum_k x^k p_k(x^f): row i = { evals[i + t N/f] } are p at x_i zeta^t
 * (zeta = w_N^(N/f)), the degree-(f-1) polynomial through them is evaluated at alpha:
 * (1/f) sum_k (alpha / x_i)^k sum_t v_t zeta^(-t k).  FriOptions::folding_factor: examples/state-transition.rs:46-47 (-f). */
void cso_fri_fold(const uint64_t *evals, uint64_t *out, unsigned log_n, unsigned log_f, uint64_t offset, uint64_t alpha) {
    const size_t N = (size_t)1 << log_n, F = (size_t)1 << log_f, Q = N / F;
    fp w = fp_root_of_unity(log_n), zeta_inv = fp_inv(fp_pow(w, Q)), invf = fp_inv(fp_from_u64(F));
    fp winv = fp_inv(w), oinv = fp_inv(offset);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < Q; i++) {
        fp xinv = fp_mul(oinv, fp_pow(winv, i));      /* 1 / x_i */
        fp r = fp_mul(alpha, xinv), rk = FP_ONE, acc = 0; /* (alpha / x)^k */
        for (size_t k = 0; k < F; k++) {
            fp zk = fp_pow(zeta_inv, k), zt = FP_ONE, s = 0; /* sum_t v_t zeta^(-t k) */
            for (size_t t = 0; t < F; t++) { s = fp_add(s, fp_mul(evals[i + t * Q], zt)); zt = fp_mul(zt, zk); }
            acc = fp_add(acc, fp_mul(rk, s));
            rk = fp_mul(rk, r);
        }
        out[i] = fp_mul(acc, invf);
    }
}
void cso_fri_fold4(const uint64_t *evals, uint64_t *out, unsigned log_n, uint64_t offset, uint64_t alpha) {
    cso_fri_fold(evals, out, log_n, 2, offset, alpha);
}

/* ---- SHA3-256 (FIPS 202): HashFunction::Sha3_256 of the reference's ProofOptions (examples/state-transition.rs:67-71) ------ */
static const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
    0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KECCAK_ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
static inline uint64_t rotl64(uint64_t x, int r) { return r ? (x << r) | (x >> (64 - r)) : x; }
static void keccak_f(uint64_t s[25]) { /* lane (x, y) at s[x + 5 y] */
    for (int rnd = 0; rnd < 24; rnd++) {
        uint64_t c[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; x++) {
            uint64_t d = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
            for (int y = 0; y < 5; y++) s[x + 5 * y] ^= d;
        }
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(s[x + 5 * y], KECCAK_ROT[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        s[0] ^= KECCAK_RC[rnd];
    }
}
void cso_sha3_256(const uint8_t *in, size_t len, uint8_t out[32]) {
    uint64_t s[25] = {0};
    uint8_t blk[136];
    while (len >= 136) {
        for (int i = 0; i < 17; i++) { uint64_t w = 0; for (int b = 0; b < 8; b++) w |= (uint64_t)in[8 * i + b] << (8 * b); s[i] ^= w; }
        keccak_f(s);
        in += 136; len -= 136;
    }
    memset(blk, 0, sizeof blk);
    memcpy(blk, in, len);
    blk[len] ^= 0x06;
    blk[135] ^= 0x80;
    for (int i = 0; i < 17; i++) { uint64_t w = 0; for (int b = 0; b < 8; b++) w |= (uint64_t)blk[8 * i + b] << (8 * b); s[i] ^= w; }
    keccak_f(s);
    for (int i = 0; i < 4; i++) for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(s[i] >> (8 * b));
}
/* the generic digest: hash_fn 0 = Blake3_256, 1 = Sha3_256 */
void cso_digest(int hash_fn, const uint8_t *in, size_t len, uint8_t out[32]) {
    if (hash_fn == 1) cso_sha3_256(in, len, out); else cso_blake3(in, len, out);
}
void cso_hash_rows_fn(int hash_fn, const uint64_t *lde, uint8_t *leaves, uint32_t width, unsigned log_n, unsigned log_b, uint32_t k0, uint32_t nk) {
    size_t n = (size_t)1 << log_n, b = (size_t)1 << log_b;
#pragma omp parallel for schedule(static) collapse(2)
    for (uint32_t k = k0; k < k0 + nk; k++)
        for (size_t j = 0; j < n; j++) {
            uint8_t buf[8 * 256];
            for (uint32_t c = 0; c < width; c++) {
                uint64_t v = lde[((size_t)(k - k0) * width + c) * n + j];
#if !CSTARK_CONV_HASHED_ELEMENT_BYTES_MONTGOMERY
                v = fp_to_u64(v); /* canonical little-endian bytes */
#endif
                for (int t = 0; t < 8; t++) buf[8 * c + t] = v >> (8 * t);
            }
            cso_digest(hash_fn, buf, 8 * (size_t)width, leaves + 32 * (b * j + k));
        }
}
void cso_merkle_build_fn(int hash_fn, uint8_t *nodes, unsigned log_leaves) {
    size_t L = (size_t)1 << log_leaves;
    memset(nodes, 0, 32);
    for (size_t lvl = L >> 1; lvl >= 1; lvl >>= 1) {
#pragma omp parallel for schedule(static)
        for (size_t i = lvl; i < 2 * lvl; i++) cso_digest(hash_fn, nodes + 64 * i, 64, nodes + 32 * i);
    }
}
