/* ORACLE (test infrastructure, not product code).
 *
 * FieldExtension::Quadratic / ::Cubic for the engine stages after the constraint evaluation.  [UPSTREAM-RECALL winterfell v0.3; the
 * extensions the fork defines for f63 are not in the reference tree -- PARITY UNPINNED.  Assumed: the two polynomials of the
 * reference's own curve tower (src/utils/ecc.rs:407-648): E2 = F_p[u] / (u^2 - 2u - 2) and E3 = F_p[v] / (v^3 + v + 1) (irreducible
 * over F_p because it is irreducible over F_p2).]  An element is m = 2 or 3 consecutive base elements (coefficients of 1, x, x^2).
 * The execution trace stays in the base field; random coefficients, the out-of-domain point, the DEEP composition and FRI live
 * in the extension.  Every coefficient multiplies a base-field constraint value, so the merged constraint evaluations are m
 * independent base-field combinations (one per component).
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "gadgets.h"

typedef struct { fp c[3]; } E; /* unused high coefficients are zero */

static inline E e_zero(void) { E r = {{0, 0, 0}}; return r; }
static inline E e_one(void) { E r = {{FP_ONE, 0, 0}}; return r; }
static inline E e_load(const uint64_t *p, int m) { E r = e_zero(); for (int i = 0; i < m; i++) r.c[i] = p[i]; return r; }
static inline void e_store(uint64_t *p, E x, int m) { for (int i = 0; i < m; i++) p[i] = x.c[i]; }
static inline E e_add(E x, E y) { E r; for (int i = 0; i < 3; i++) r.c[i] = fp_add(x.c[i], y.c[i]); return r; }
static inline E e_sub(E x, E y) { E r; for (int i = 0; i < 3; i++) r.c[i] = fp_sub(x.c[i], y.c[i]); return r; }
static inline E e_scale(E x, fp s) { E r; for (int i = 0; i < 3; i++) r.c[i] = fp_mul(x.c[i], s); return r; }
static inline E e_sub_base(fp x, E z) { E r = {{fp_sub(x, z.c[0]), fp_neg(z.c[1]), fp_neg(z.c[2])}}; return r; } /* x - z */
static E e_mul(E x, E y, int m) {
    E r = e_zero();
    if (m == 2) { /* u^2 = C1 u + C0 (include/cstark_conventions.h; assumed 2u + 2) */
        fp bd = fp_mul(x.c[1], y.c[1]);
        r.c[0] = fp_add(fp_mul(x.c[0], y.c[0]), fp_mul_small(bd, CSTARK_CONV_E2_C0));
        r.c[1] = fp_add(fp_add(fp_mul(x.c[0], y.c[1]), fp_mul(x.c[1], y.c[0])), fp_mul_small(bd, CSTARK_CONV_E2_C1));
        return r;
    }
    /* v^3 = C2 v^2 + C1 v + C0 (assumed -v - 1), v^4 = (C2^2 + C1) v^2 + (C2 C1 + C0) v + C2 C0 */
    const int C0 = CSTARK_CONV_E3_C0, C1 = CSTARK_CONV_E3_C1, C2 = CSTARK_CONV_E3_C2;
    fp d0 = fp_mul(x.c[0], y.c[0]);
    fp d1 = fp_add(fp_mul(x.c[0], y.c[1]), fp_mul(x.c[1], y.c[0]));
    fp d2 = fp_add(fp_add(fp_mul(x.c[0], y.c[2]), fp_mul(x.c[1], y.c[1])), fp_mul(x.c[2], y.c[0]));
    fp d3 = fp_add(fp_mul(x.c[1], y.c[2]), fp_mul(x.c[2], y.c[1]));
    fp d4 = fp_mul(x.c[2], y.c[2]);
    r.c[0] = fp_add(fp_add(d0, fp_mul_small(d3, C0)), fp_mul_small(d4, C2 * C0));
    r.c[1] = fp_add(fp_add(d1, fp_mul_small(d3, C1)), fp_mul_small(d4, C2 * C1 + C0));
    r.c[2] = fp_add(fp_add(d2, fp_mul_small(d3, C2)), fp_mul_small(d4, C2 * C2 + C1));
    return r;
}
static E e_inv(E x, int m) {
    E r = e_zero();
    if (m == 2) { /* 1/(a + b u) = (a + C1 b - b u) / (a (a + C1 b) - C0 b^2) */
        fp a = x.c[0], b = x.c[1];
        fp s = fp_add(a, fp_mul_small(b, CSTARK_CONV_E2_C1));
        fp t = fp_inv(fp_sub(fp_mul(a, s), fp_mul_small(fp_sqr(b), CSTARK_CONV_E2_C0)));
        r.c[0] = fp_mul(s, t);
        r.c[1] = fp_mul(fp_neg(b), t);
        return r;
    }
    /* first column of the adjugate of the matrix of "multiply by a + b v + c v^2" (columns e, e v, e v^2) over its determinant; for
     * v^3 + v + 1 this is the cubic layer of ecc.rs:551-591 */
    const int C0 = CSTARK_CONV_E3_C0, C1 = CSTARK_CONV_E3_C1, C2 = CSTARK_CONV_E3_C2;
    fp a = x.c[0], b = x.c[1], c = x.c[2];
    fp y0 = fp_mul_small(c, C0), y1 = fp_add(a, fp_mul_small(c, C1)), y2 = fp_add(b, fp_mul_small(c, C2));
    fp w0 = fp_mul_small(y2, C0), w1 = fp_add(y0, fp_mul_small(y2, C1)), w2 = fp_add(y1, fp_mul_small(y2, C2));
    fp r0 = fp_sub(fp_mul(y1, w2), fp_mul(y2, w1));
    fp r1 = fp_sub(fp_mul(c, w1), fp_mul(b, w2));
    fp r2 = fp_sub(fp_mul(b, y2), fp_mul(c, y1));
    fp nrm = fp_add(fp_add(fp_mul(a, r0), fp_mul(y0, r1)), fp_mul(w0, r2));
    fp t = fp_inv(nrm);
    r.c[0] = fp_mul(r0, t); r.c[1] = fp_mul(r1, t); r.c[2] = fp_mul(r2, t);
    return r;
}

/* values of `width` base-coefficient columns at one point of the extension: out[c][m] */
void cso_evaluate_polys_at_ext(const uint64_t *coeffs, uint32_t width, unsigned log_n, const uint64_t *zp, uint64_t *out, int m) {
    const size_t n = (size_t)1 << log_n;
    const E z = e_load(zp, m);
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t c = 0; c < width; c++) {
        E acc = e_zero();
        for (size_t k = n; k-- > 0;) {
            acc = e_mul(acc, z, m);
            acc.c[0] = fp_add(acc.c[0], coeffs[(size_t)c * n + k]);
        }
        e_store(out + (size_t)m * c, acc, m);
    }
}

/* DEEP composition over the extension.  trace_lde [b][W][n] base; comp_lde [b][m nb][n]: column m i + k = component k of
 * composition column i; ood_trace = T(z)[W] | T(z w)[W], ood_comp = H_i(z^nb), coefficient arrays alpha, beta [W], delta [nb],
 * deg_a, deg_b: all as m-tuples.  out [m][b][n]: component-major. */
void cso_deep_composition_ext(const uint64_t *trace_lde, const uint64_t *comp_lde, uint32_t width, uint32_t nb, const uint64_t *zp,
                              const uint64_t *ood_trace, const uint64_t *ood_comp, const uint64_t *alpha, const uint64_t *beta,
                              const uint64_t *delta, const uint64_t *deg_ap, const uint64_t *deg_bp, uint64_t *out, unsigned log_n, unsigned log_b,
                              int m) {
    const size_t n = (size_t)1 << log_n, b = (size_t)1 << log_b;
    const fp g = fp_from_u64(FP_LDE_OFFSET_CAN), wbn = fp_root_of_unity(log_n + log_b), wn = fp_root_of_unity(log_n);
    const E z = e_load(zp, m), zw = e_scale(z, wn);
    E zb = e_one();
    for (uint32_t i = 0; i < nb; i++) zb = e_mul(zb, z, m);
    const E da = e_load(deg_ap, m), db = e_load(deg_bp, m);
#pragma omp parallel for schedule(static) collapse(2)
    for (size_t k = 0; k < b; k++)
        for (size_t j = 0; j < n; j++) {
            const fp x = fp_mul(fp_mul(g, fp_pow(wbn, k)), fp_pow(wn, j));
            const E i1 = e_inv(e_sub_base(x, z), m), i2 = e_inv(e_sub_base(x, zw), m), i3 = e_inv(e_sub_base(x, zb), m);
            E s1 = e_zero(), s2 = s1, s3 = s1;
            for (uint32_t c = 0; c < width; c++) {
                const fp t = trace_lde[(k * width + c) * n + j];
                s1 = e_add(s1, e_mul(e_load(alpha + (size_t)m * c, m), e_sub_base(t, e_load(ood_trace + (size_t)m * c, m)), m));
                s2 = e_add(s2, e_mul(e_load(beta + (size_t)m * c, m), e_sub_base(t, e_load(ood_trace + (size_t)m * (width + c), m)), m));
            }
            for (uint32_t i = 0; i < nb; i++) {
                E h = e_zero();
                for (int q = 0; q < m; q++) h.c[q] = comp_lde[(k * m * nb + (size_t)m * i + q) * n + j];
                s3 = e_add(s3, e_mul(e_load(delta + (size_t)m * i, m), e_sub(h, e_load(ood_comp + (size_t)m * i, m)), m));
            }
            E acc = e_add(e_add(e_mul(s1, i1, m), e_mul(s2, i2, m)), e_mul(s3, i3, m));
            acc = e_mul(acc, e_add(da, e_scale(db, x)), m);
            for (int q = 0; q < m; q++) out[((size_t)q * b + k) * n + j] = acc.c[q];
        }
}

/* FRI folding by f = 2^log_f over the extension: evals [m][N] component-major over offset * <w_N>; out [m][N/f] */
void cso_fri_fold_ext(const uint64_t *evals, uint64_t *out, unsigned log_n, unsigned log_f, uint64_t offset, const uint64_t *alphap, int m) {
    const size_t N = (size_t)1 << log_n, F = (size_t)1 << log_f, Q = N / F;
    const fp w = fp_root_of_unity(log_n), zeta_inv = fp_inv(fp_pow(w, Q)), invf = fp_inv(fp_from_u64(F));
    const fp winv = fp_inv(w), oinv = fp_inv(offset);
    const E alpha = e_load(alphap, m);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < Q; i++) {
        const fp xinv = fp_mul(oinv, fp_pow(winv, i));
        const E r = e_scale(alpha, xinv);
        E rk = e_one(), acc = e_zero();
        for (size_t k = 0; k < F; k++) {
            const fp zk = fp_pow(zeta_inv, k);
            fp zt = FP_ONE;
            E s = e_zero();
            for (size_t t = 0; t < F; t++) {
                E v = e_zero();
                for (int q = 0; q < m; q++) v.c[q] = evals[(size_t)q * N + i + t * Q];
                s = e_add(s, e_scale(v, zt));
                zt = fp_mul(zt, zk);
            }
            acc = e_add(acc, e_mul(rk, s, m));
            rk = e_mul(rk, r, m);
        }
        acc = e_scale(acc, invf);
        for (int q = 0; q < m; q++) out[(size_t)q * Q + i] = acc.c[q];
    }
}
void cso_fri_fold4_ext(const uint64_t *evals, uint64_t *out, unsigned log_n, uint64_t offset, const uint64_t *alphap, int m) {
    cso_fri_fold_ext(evals, out, log_n, 2, offset, alphap, m);
}
