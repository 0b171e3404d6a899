/* ORACLE (test infrastructure, not product code).
 *
 * FieldExtension::Quadratic for the engine stages after the constraint evaluation.  [UPSTREAM-RECALL winterfell v0.3; the
 * extension the fork defines for f63 is not in the reference tree -- PARITY UNPINNED.  Assumed: E = F_p[u] / (u^2 - 2u - 2), the
 * quadratic extension the reference itself uses as the base of its curve tower (src/utils/ecc.rs:407-466); an element is the
 * pair (a, b) = a + b u, stored as two consecutive base elements.]  The execution trace stays in the base field; random
 * coefficients, the out-of-domain point, the DEEP composition and FRI live in E.  Because every coefficient multiplies a
 * base-field constraint value, the merged constraint evaluations are two independent base-field combinations (components a, b).
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "gadgets.h"

typedef fp2 E;
static inline E e_make(fp a, fp b) { E r = {{a, b}}; return r; }
static inline E e_scale(E x, fp s) { return e_make(fp_mul(x.c[0], s), fp_mul(x.c[1], s)); }
static inline E e_sub_base(fp x, E z) { return e_make(fp_sub(x, z.c[0]), fp_neg(z.c[1])); } /* x - z, x in the base field */

/* values of `width` base-coefficient columns at one point of E: out[c] = (a, b) */
void cso_evaluate_polys_at_ext(const uint64_t *coeffs, uint32_t width, unsigned log_n, const uint64_t *z2, uint64_t *out) {
    const size_t n = (size_t)1 << log_n;
    const E z = e_make(z2[0], z2[1]);
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t c = 0; c < width; c++) {
        E acc = e_make(0, 0);
        for (size_t m = n; m-- > 0;) {
            acc = fp2_mul(acc, z);
            acc.c[0] = fp_add(acc.c[0], coeffs[(size_t)c * n + m]);
        }
        out[2 * c] = acc.c[0]; out[2 * c + 1] = acc.c[1];
    }
}

/* DEEP composition over E.  trace_lde [b][W][n] base; comp_lde [b][2 nb][n]: column 2i + k = component k of composition column i;
 * ood_trace = T(z)[W] | T(z w)[W] as pairs, ood_comp = H_i(z^nb) as pairs; alpha, beta [W], delta [nb], deg_a, deg_b in E.
 * out [2][b][n]: component-major. */
void cso_deep_composition_ext(const uint64_t *trace_lde, const uint64_t *comp_lde, uint32_t width, uint32_t nb, const uint64_t *z2,
                              const uint64_t *ood_trace, const uint64_t *ood_comp, const uint64_t *alpha, const uint64_t *beta,
                              const uint64_t *delta, const uint64_t *deg_a2, const uint64_t *deg_b2, uint64_t *out, unsigned log_n, unsigned log_b) {
    const size_t n = (size_t)1 << log_n, b = (size_t)1 << log_b;
    const fp g = fp_from_u64(FP_GENERATOR_CAN), wbn = fp_root_of_unity(log_n + log_b), wn = fp_root_of_unity(log_n);
    const E z = e_make(z2[0], z2[1]), zw = e_scale(z, wn);
    E zb = e_make(FP_ONE, 0);
    for (uint32_t i = 0; i < nb; i++) zb = fp2_mul(zb, z);
    const E da = e_make(deg_a2[0], deg_a2[1]), db = e_make(deg_b2[0], deg_b2[1]);
#pragma omp parallel for schedule(static) collapse(2)
    for (size_t k = 0; k < b; k++)
        for (size_t j = 0; j < n; j++) {
            const fp x = fp_mul(fp_mul(g, fp_pow(wbn, k)), fp_pow(wn, j));
            const E i1 = fp2_inv(e_sub_base(x, z)), i2 = fp2_inv(e_sub_base(x, zw)), i3 = fp2_inv(e_sub_base(x, zb));
            E s1 = e_make(0, 0), s2 = s1, s3 = s1;
            for (uint32_t c = 0; c < width; c++) {
                const fp t = trace_lde[(k * width + c) * n + j];
                const E a = e_make(alpha[2 * c], alpha[2 * c + 1]), bt = e_make(beta[2 * c], beta[2 * c + 1]);
                s1 = fp2_add(s1, fp2_mul(a, e_sub_base(t, e_make(ood_trace[2 * c], ood_trace[2 * c + 1]))));
                s2 = fp2_add(s2, fp2_mul(bt, e_sub_base(t, e_make(ood_trace[2 * (width + c)], ood_trace[2 * (width + c) + 1]))));
            }
            for (uint32_t i = 0; i < nb; i++) {
                const E h = e_make(comp_lde[(k * 2 * nb + 2 * i) * n + j], comp_lde[(k * 2 * nb + 2 * i + 1) * n + j]);
                const E dl = e_make(delta[2 * i], delta[2 * i + 1]);
                s3 = fp2_add(s3, fp2_mul(dl, fp2_sub(h, e_make(ood_comp[2 * i], ood_comp[2 * i + 1]))));
            }
            E acc = fp2_add(fp2_add(fp2_mul(s1, i1), fp2_mul(s2, i2)), fp2_mul(s3, i3));
            acc = fp2_mul(acc, fp2_add(da, e_scale(db, x)));
            out[k * n + j] = acc.c[0];
            out[(b + k) * n + j] = acc.c[1];
        }
}

/* FRI folding by 4 over E: evals [2][N] component-major over offset * <w_N>; out [2][N/4] */
void cso_fri_fold4_ext(const uint64_t *evals, uint64_t *out, unsigned log_n, uint64_t offset, const uint64_t *alpha2) {
    const size_t N = (size_t)1 << log_n, Q = N / 4;
    const fp w = fp_root_of_unity(log_n), zeta_inv = fp_inv(fp_pow(w, Q)), inv4 = fp_inv(fp_from_u64(4));
    const fp winv = fp_inv(w), oinv = fp_inv(offset);
    const E alpha = e_make(alpha2[0], alpha2[1]);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < Q; i++) {
        const fp xinv = fp_mul(oinv, fp_pow(winv, i));
        const E r = e_scale(alpha, xinv);
        E rk = e_make(FP_ONE, 0), acc = e_make(0, 0);
        for (int k = 0; k < 4; k++) {
            const fp zk = fp_pow(zeta_inv, k);
            fp zt = FP_ONE;
            E s = e_make(0, 0);
            for (int t = 0; t < 4; t++) {
                s = fp2_add(s, e_scale(e_make(evals[i + t * Q], evals[N + i + t * Q]), zt));
                zt = fp_mul(zt, zk);
            }
            acc = fp2_add(acc, fp2_mul(rk, s));
            rk = fp2_mul(rk, r);
        }
        acc = e_scale(acc, inv4);
        out[i] = acc.c[0];
        out[Q + i] = acc.c[1];
    }
}
