/* ORACLE (test infrastructure, not product code).
 *
 * CPU restatement of the reference's L0 gadgets: Rescue-Prime over f63, F_p2/F_p6 tower and the
 * Cheetah-curve point formulas, field double-and-add, and the constraint helper semantics.
 * Each function cites the reference lines it follows (paths relative to /root/reference).
 */
#ifndef CS_ORACLE_GADGETS_H
#define CS_ORACLE_GADGETS_H
#include <string.h>
#include "fp.h"
#include "constants_gen.h"

#define RESCUE_STATE 14
#define RESCUE_RATE 7
#define RESCUE_ROUNDS 7
#define HASH_CYCLE 8
#define COORD 6
#define AFFINE 12
#define PROJ 18

/* ---- constraint helpers: src/utils/mod.rs:28-68 -------------------------------------------- */
static inline fp c_are_equal(fp a, fp b) { return fp_sub(a, b); }
static inline fp c_is_binary(fp a) { return fp_sub(fp_mul(a, a), a); }
static inline fp c_not(fp a) { return fp_sub(FP_ONE, a); }
/* EvaluationResult::agg_constraint: result[i] += flag * value */
static inline void agg(fp *result, int i, fp flag, fp value) { result[i] = fp_add(result[i], fp_mul(flag, value)); }

/* ---- Rescue-Prime: src/utils/rescue.rs ------------------------------------------------------ */
static inline void rescue_sbox(fp *s) { /* :327-333, alpha = 3 */
    for (int i = 0; i < RESCUE_STATE; i++) s[i] = fp_mul(s[i], fp_sqr(s[i]));
}
static inline void rescue_inv_sbox(fp *s) { /* :337-341 */
    for (int i = 0; i < RESCUE_STATE; i++) s[i] = fp_pow(s[i], CS_INV_ALPHA);
}
static inline void rescue_matvec(const uint64_t *m, fp *s) { /* :345-375 */
    fp out[RESCUE_STATE];
    for (int i = 0; i < RESCUE_STATE; i++) {
        fp acc = 0;
        for (int j = 0; j < RESCUE_STATE; j++) acc = fp_add(acc, fp_mul(m[i * RESCUE_STATE + j], s[j]));
        out[i] = acc;
    }
    memcpy(s, out, sizeof out);
}
/* one round, constants ARK[step % 8]: :246-263 */
static inline void rescue_apply_round(fp *s, size_t step) {
    const uint64_t *ark = CS_ARK_MONT + (step % HASH_CYCLE) * 2 * RESCUE_STATE;
    rescue_sbox(s);
    rescue_matvec(CS_MDS_MONT, s);
    for (int i = 0; i < RESCUE_STATE; i++) s[i] = fp_add(s[i], ark[i]);
    rescue_inv_sbox(s);
    rescue_matvec(CS_MDS_MONT, s);
    for (int i = 0; i < RESCUE_STATE; i++) s[i] = fp_add(s[i], ark[RESCUE_STATE + i]);
}
static inline void rescue_apply_permutation(fp *s) { /* :236-241 */
    for (int i = 0; i < RESCUE_ROUNDS; i++) rescue_apply_round(s, i);
}
/* Rescue63::merge :143-152 */
static inline void rescue_merge(const fp *a, const fp *b, fp *out) {
    fp s[RESCUE_STATE];
    memcpy(s, a, 7 * sizeof(fp));
    memcpy(s + 7, b, 7 * sizeof(fp));
    rescue_apply_permutation(s);
    memcpy(out, s, 7 * sizeof(fp));
}
/* Rescue63::digest :108-130 (no padding) */
static inline void rescue_digest(const fp *data, size_t n, fp *out) {
    fp s[RESCUE_STATE] = {0};
    size_t i = 0;
    for (size_t k = 0; k < n; k++) {
        s[i] = fp_add(s[i], data[k]);
        if (++i % RESCUE_RATE == 0) { rescue_apply_permutation(s); i = 0; }
    }
    if (i > 0) rescue_apply_permutation(s);
    memcpy(out, s, 7 * sizeof(fp));
}
/* enforce_round :269-300; ark = the 28 periodic round-constant values at this point */
static inline void rescue_enforce_round(fp *result, const fp *cur, const fp *next, const fp *ark, fp flag) {
    fp s1[RESCUE_STATE], s2[RESCUE_STATE];
    memcpy(s1, cur, sizeof s1);
    rescue_sbox(s1);
    rescue_matvec(CS_MDS_MONT, s1);
    for (int i = 0; i < RESCUE_STATE; i++) s1[i] = fp_add(s1[i], ark[i]);
    for (int i = 0; i < RESCUE_STATE; i++) s2[i] = fp_sub(next[i], ark[RESCUE_STATE + i]);
    rescue_matvec(CS_INV_MDS_MONT, s2);
    rescue_sbox(s2);
    for (int i = 0; i < RESCUE_STATE; i++) agg(result, i, flag, c_are_equal(s2[i], s1[i]));
}

/* ---- F_p2 = F_p[u]/(u^2 - 2u - 2): src/utils/ecc.rs:407-466 --------------------------------- */
typedef struct { fp c[2]; } fp2;
typedef struct { fp c[6]; } fp6;

static inline fp2 fp2_add(fp2 a, fp2 b) { fp2 r = {{fp_add(a.c[0], b.c[0]), fp_add(a.c[1], b.c[1])}}; return r; }
static inline fp2 fp2_sub(fp2 a, fp2 b) { fp2 r = {{fp_sub(a.c[0], b.c[0]), fp_sub(a.c[1], b.c[1])}}; return r; }
static inline fp2 fp2_dbl(fp2 a) { return fp2_add(a, a); }
static inline fp2 fp2_neg(fp2 a) { fp2 r = {{fp_neg(a.c[0]), fp_neg(a.c[1])}}; return r; }
static inline fp2 fp2_mul(fp2 a, fp2 b) { /* :424-439 */
    fp aa = fp_mul(a.c[0], b.c[0]), bb = fp_mul(a.c[1], b.c[1]);
    fp t = fp_mul(fp_sub(a.c[0], a.c[1]), fp_sub(b.c[1], b.c[0]));
    fp c0 = fp_add(fp_dbl(bb), aa);
    fp2 r = {{c0, fp_add(fp_add(bb, c0), t)}};
    return r;
}
static inline fp2 fp2_sqr(fp2 a) { /* :407-421 */
    fp aa = fp_sqr(a.c[0]), bb = fp_sqr(a.c[1]);
    fp t = fp_sqr(fp_sub(a.c[0], a.c[1]));
    fp c0 = fp_add(fp_dbl(bb), aa);
    fp2 r = {{c0, fp_sub(fp_add(bb, c0), t)}};
    return r;
}
static inline fp2 fp2_inv(fp2 a) { /* :442-446 */
    fp t = fp_inv(fp_sub(fp_add(fp_sqr(a.c[0]), fp_mul(fp_dbl(a.c[0]), a.c[1])), fp_dbl(fp_sqr(a.c[1]))));
    fp2 r = {{fp_mul(fp_add(a.c[0], fp_dbl(a.c[1])), t), fp_mul(fp_neg(a.c[1]), t)}};
    return r;
}

/* ---- F_p6 = F_p2[v]/(v^3 + v + 1): src/utils/ecc.rs:468-648 --------------------------------- */
static inline fp2 f6c(const fp6 *a, int k) { fp2 r = {{a->c[2 * k], a->c[2 * k + 1]}}; return r; }
static inline fp6 f6pack(fp2 c0, fp2 c1, fp2 c2) {
    fp6 r = {{c0.c[0], c0.c[1], c1.c[0], c1.c[1], c2.c[0], c2.c[1]}};
    return r;
}
static inline fp6 fp6_add(fp6 a, fp6 b) { fp6 r; for (int i = 0; i < 6; i++) r.c[i] = fp_add(a.c[i], b.c[i]); return r; }
static inline fp6 fp6_sub(fp6 a, fp6 b) { fp6 r; for (int i = 0; i < 6; i++) r.c[i] = fp_sub(a.c[i], b.c[i]); return r; }
static inline fp6 fp6_dbl(fp6 a) { return fp6_add(a, a); }
static inline fp6 fp6_karatsuba_tail(fp2 aa, fp2 bb, fp2 cc, fp2 ab, fp2 ac, fp2 bc) { /* :534-547 */
    fp2 tmp = fp2_add(fp2_add(aa, bb), cc);
    fp2 c0 = fp2_sub(tmp, bc);
    fp2 c1 = fp2_sub(fp2_sub(ab, bc), aa);
    fp2 c2 = fp2_add(fp2_sub(fp2_sub(ac, tmp), cc), fp2_add(bb, bb));
    return f6pack(c0, c1, c2);
}
static inline fp6 fp6_mul(fp6 a, fp6 b) { /* :506-548 */
    fp2 a0 = f6c(&a, 0), a1 = f6c(&a, 1), a2 = f6c(&a, 2), b0 = f6c(&b, 0), b1 = f6c(&b, 1), b2 = f6c(&b, 2);
    return fp6_karatsuba_tail(fp2_mul(a0, b0), fp2_mul(a1, b1), fp2_mul(a2, b2),
                              fp2_mul(fp2_add(a0, a1), fp2_add(b0, b1)), fp2_mul(fp2_add(a0, a2), fp2_add(b0, b2)),
                              fp2_mul(fp2_add(a1, a2), fp2_add(b1, b2)));
}
static inline fp6 fp6_sqr(fp6 a) { /* :469-503 */
    fp2 a0 = f6c(&a, 0), a1 = f6c(&a, 1), a2 = f6c(&a, 2);
    return fp6_karatsuba_tail(fp2_sqr(a0), fp2_sqr(a1), fp2_sqr(a2), fp2_sqr(fp2_add(a0, a1)),
                              fp2_sqr(fp2_add(a0, a2)), fp2_sqr(fp2_add(a1, a2)));
}
static inline fp6 fp6_inv(fp6 a) { /* :551-591 */
    fp2 c0 = f6c(&a, 0), c1 = f6c(&a, 1), c2 = f6c(&a, 2);
    fp2 s0 = fp2_sqr(c0), s1 = fp2_sqr(c1), s2 = fp2_sqr(c2);
    fp2 t = fp2_mul(c0, fp2_add(s0, s1));
    t = fp2_sub(t, fp2_mul(c1, s1));
    t = fp2_add(t, fp2_mul(fp2_add(c0, fp2_sub(c2, c1)), s2));
    fp2 w = fp2_mul(fp2_add(fp2_dbl(c0), c0), c1);
    w = fp2_mul(fp2_sub(fp2_dbl(s0), w), c2);
    t = fp2_inv(fp2_sub(t, w));
    fp2 r0 = fp2_sub(fp2_add(fp2_add(s0, s1), s2), fp2_mul(fp2_sub(fp2_dbl(c0), c1), c2));
    r0 = fp2_mul(r0, t);
    fp2 r1 = fp2_mul(fp2_neg(fp2_add(fp2_mul(c0, c1), s2)), t);
    fp2 r2 = fp2_mul(fp2_add(fp2_sub(s1, fp2_mul(c0, c2)), s2), t);
    return f6pack(r0, r1, r2);
}
static inline fp6 fp6_load(const fp *p) { fp6 r; memcpy(r.c, p, sizeof r.c); return r; }
static inline void fp6_store(fp *p, fp6 a) { memcpy(p, a.c, sizeof a.c); }
static inline fp6 fp6_b3(void) { return fp6_load(CS_B3_MONT); }

/* ---- curve y^2 = x^3 + x + B, complete projective formulas: src/utils/ecc.rs:186-404 --------- */
static inline void ecc_double(fp *st) { /* compute_double :186-242 */
    fp6 X = fp6_load(st), Y = fp6_load(st + 6), Z = fp6_load(st + 12), b3 = fp6_b3();
    fp6 t0 = fp6_sqr(X), t1 = fp6_sqr(Y), t2 = fp6_sqr(Z);
    fp6 t3 = fp6_dbl(fp6_mul(X, Y));
    fp6 z3 = fp6_dbl(fp6_mul(X, Z));
    fp6 y3 = fp6_add(z3, fp6_mul(b3, t2));
    fp6 x3 = fp6_sub(t1, y3);
    y3 = fp6_add(t1, y3);
    y3 = fp6_mul(x3, y3);
    x3 = fp6_mul(t3, x3);
    z3 = fp6_mul(b3, z3);
    t3 = fp6_add(fp6_sub(t0, t2), z3);
    t0 = fp6_add(fp6_add(fp6_dbl(t0), t0), t2);
    t0 = fp6_mul(t0, t3);
    y3 = fp6_add(y3, t0);
    t2 = fp6_dbl(fp6_mul(Y, Z));
    t0 = fp6_mul(t2, t3);
    x3 = fp6_sub(x3, t0);
    z3 = fp6_dbl(fp6_dbl(fp6_mul(t2, t1)));
    fp6_store(st, x3); fp6_store(st + 6, y3); fp6_store(st + 12, z3);
}
static inline void ecc_add(fp *st, const fp *pt) { /* compute_add :256-328 (pt projective) */
    fp6 X1 = fp6_load(st), Y1 = fp6_load(st + 6), Z1 = fp6_load(st + 12);
    fp6 X2 = fp6_load(pt), Y2 = fp6_load(pt + 6), Z2 = fp6_load(pt + 12), b3 = fp6_b3();
    fp6 t0 = fp6_mul(X1, X2), t1 = fp6_mul(Y1, Y2), t2 = fp6_mul(Z1, Z2);
    fp6 t3 = fp6_sub(fp6_mul(fp6_add(X1, Y1), fp6_add(X2, Y2)), fp6_add(t0, t1));
    fp6 t4 = fp6_sub(fp6_mul(fp6_add(X1, Z1), fp6_add(X2, Z2)), fp6_add(t0, t2));
    fp6 t5 = fp6_sub(fp6_mul(fp6_add(Y1, Z1), fp6_add(Y2, Z2)), fp6_add(t1, t2));
    fp6 z3 = fp6_add(fp6_mul(b3, t2), t4);
    fp6 x3 = fp6_sub(t1, z3);
    z3 = fp6_add(t1, z3);
    fp6 y3 = fp6_mul(x3, z3);
    t1 = fp6_add(fp6_add(fp6_dbl(t0), t0), t2);
    t4 = fp6_add(fp6_mul(b3, t4), fp6_sub(t0, t2));
    y3 = fp6_add(y3, fp6_mul(t1, t4));
    x3 = fp6_sub(fp6_mul(t3, x3), fp6_mul(t5, t4));
    z3 = fp6_add(fp6_mul(t5, z3), fp6_mul(t3, t1));
    fp6_store(st, x3); fp6_store(st + 6, y3); fp6_store(st + 12, z3);
}
static inline void ecc_add_mixed(fp *st, const fp *pt) { /* compute_add_mixed :343-404 (pt affine) */
    fp6 X1 = fp6_load(st), Y1 = fp6_load(st + 6), Z1 = fp6_load(st + 12);
    fp6 X2 = fp6_load(pt), Y2 = fp6_load(pt + 6), b3 = fp6_b3();
    fp6 t0 = fp6_mul(X1, X2), t1 = fp6_mul(Y1, Y2);
    fp6 t3 = fp6_sub(fp6_mul(fp6_add(X2, Y2), fp6_add(X1, Y1)), fp6_add(t0, t1));
    fp6 t4 = fp6_add(fp6_mul(X2, Z1), X1);
    fp6 t5 = fp6_add(fp6_mul(Y2, Z1), Y1);
    fp6 z3 = fp6_add(fp6_mul(Z1, b3), t4);
    fp6 x3 = fp6_sub(t1, z3);
    z3 = fp6_add(t1, z3);
    fp6 y3 = fp6_mul(x3, z3);
    t1 = fp6_add(fp6_add(fp6_dbl(t0), t0), Z1);
    t4 = fp6_add(fp6_mul(t4, b3), fp6_sub(t0, Z1));
    y3 = fp6_add(y3, fp6_mul(t1, t4));
    x3 = fp6_sub(fp6_mul(t3, x3), fp6_mul(t5, t4));
    z3 = fp6_add(fp6_mul(t5, z3), fp6_mul(t3, t1));
    fp6_store(st, x3); fp6_store(st + 6, y3); fp6_store(st + 12, z3);
}
/* trace appliers :51-67: addition only when the bit register (index 18) holds ONE */
static inline void ecc_apply_addition_mixed(fp *st, const fp *pt) { if (st[PROJ] == FP_ONE) ecc_add_mixed(st, pt); }
static inline void ecc_apply_addition(fp *st, const fp *pt) { if (st[PROJ] == FP_ONE) ecc_add(st, pt); }

/* constraint gadgets :73-172 */
static inline void ecc_enforce_doubling(fp *result, const fp *cur, const fp *next, fp flag) {
    fp s1[PROJ];
    memcpy(s1, cur, sizeof s1);
    ecc_double(s1);
    for (int i = 0; i < PROJ; i++) agg(result, i, flag, c_are_equal(next[i], s1[i]));
    agg(result, PROJ, flag, c_is_binary(cur[PROJ]));
}
static inline void ecc_enforce_addition_mixed(fp *result, const fp *cur, const fp *next, const fp *pt, fp flag) {
    fp s1[PROJ];
    memcpy(s1, cur, sizeof s1);
    ecc_add_mixed(s1, pt);
    fp bit = cur[PROJ];
    for (int i = 0; i < PROJ; i++)
        agg(result, i, flag, c_are_equal(next[i], fp_add(fp_mul(bit, s1[i]), fp_mul(c_not(bit), cur[i]))));
    agg(result, PROJ, flag, c_are_equal(cur[PROJ], next[PROJ]));
}
static inline void ecc_enforce_addition_reduce_x(fp *result, const fp *cur, const fp *next, const fp *pt, fp flag) {
    fp s1[PROJ];
    memcpy(s1, cur, sizeof s1);
    ecc_add(s1, pt);
    fp6 xz = fp6_mul(fp6_load(next), fp6_load(s1 + AFFINE));
    for (int i = 0; i < COORD; i++) agg(result, i, flag, c_are_equal(xz.c[i], s1[i]));
    for (int i = COORD; i < PROJ; i++) agg(result, i, flag, c_are_equal(next[i], s1[i]));
}

/* ---- field double-and-add: src/utils/field.rs ------------------------------------------------ */
static inline void field_apply_double_and_add(fp *st, int value_pos, int bit_pos) { /* :16-22 */
    st[value_pos] = fp_add(fp_dbl(st[value_pos]), st[bit_pos]);
}
static inline void field_enforce_double_and_add(fp *result, const fp *cur, const fp *next, int vp, int bp, fp flag) { /* :31-50 */
    agg(result, vp, flag, c_are_equal(next[vp], fp_add(fp_dbl(cur[vp]), next[bp])));
    agg(result, bp, flag, c_is_binary(next[bp]));
}
static inline void field_enforce_double_and_add_constrained(fp *result, const fp *cur, const fp *next, int vp, int bp, fp flag) { /* :54-70 */
    agg(result, vp, flag, c_are_equal(next[vp], fp_add(fp_dbl(cur[vp]), next[bp])));
}
#endif
