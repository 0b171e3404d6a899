// K6 -- constraint evaluation for the state-transition AIR over the LDE / constraint-evaluation domain.
//
// Per point x = g * w_{bn}^k * w_n^j this computes what the engine's ConstraintEvaluator obtains by calling
// back into the reference:
//   Air::evaluate_transition            /root/reference/src/air.rs:114-173
//   evaluate_constraints                src/air.rs:383-610  (merkle::init  src/merkle/init/air.rs:159-202,
//                                       merkle::update src/merkle/update/air.rs:215-369,
//                                       schnorr src/schnorr/air.rs:394-531, range src/utils/field.rs:31-50)
//   gadgets                             src/utils/rescue.rs:269-300, src/utils/ecc.rs:73-172
// and then the driver's merge [UPSTREAM-RECALL winterfell v0.3]: sum_i (alpha_i + beta_i x^adj_i) C_i(x)
// divided by the transition divisor, plus the four boundary terms of get_assertions (src/air.rs:175-184).
//
// Mapping: ONE LANE PER EVALUATION POINT.  With the coset-major column layout, lane j reads element j of a
// column, so every load of the 94 current-row and 94 next-row registers is a contiguous 512-byte line per
// wave and there is no cross-lane traffic at all.  The kernel is integer-VALU bound (~4.4 k modular products
// per point), not HBM bound; see DESIGN.md.
//
// Exact field arithmetic lets the evaluation be reorganised without changing any value:
//  * the Rescue round gadget is evaluated once per register window (5 windows) although the reference calls
//    it 9 times -- merkle::init and merkle::update constrain the same registers with different flags;
//  * result slots are never materialised in the fused kernel: each flag*value is folded straight into the
//    random linear combination (the index aliasing of src/constants.rs:56-68 is reproduced because every
//    contribution still uses the coefficient of the slot the reference adds it to).
#include <type_traits>
#include "constraints.h"
#include "mds_mfma.cuh"
#include "rounds_layout.h"
#include "rescue.cuh"
#include "tower.cuh"

namespace cs {
namespace {

constexpr int NT = 128; // threads per workgroup

// register / result / periodic-column indices (src/merkle/constants.rs:33-63, src/constants.rs:35-116)
enum {
    S_INIT = 0, S_UPD = 15, R_INIT = 29, R_UPD = 44, PREV_ROOT = 58, S_KEY = 65, R_KEY = 77, DELTA_COPY = 89, SIGMA_COPY = 90,
    NONCE_COPY = 91, DELTA_BIT = 56, DELTA_ACC = 57, SIGMA_BIT = 92, SIGMA_ACC = 93,
    VALUE_RES = 65, BALANCE_RES = 90, NONCE_UPD_RES = 91, INT_ROOT_RES = 92, PREV_MATCH_RES = 99, S_KEY_RES = 101, R_KEY_RES = 103,
    DELTA_COPY_RES = 105, SIGMA_COPY_RES = 106, NONCE_COPY_RES = 107, DELTA_RANGE_RES = 108, SIGMA_RANGE_RES = 109,
    P_SETUP = 0, P_MERKLE = 1, P_HASH_INPUT = 2, P_FINISH = 3, P_HASH = 4, P_SCHNORR = 5, P_SCALAR_MULT = 6, P_DOUBLING = 7,
    P_DIGEST = 8, P_SCHNORR_HASH = 12, P_HASH_INTERNAL = 13, P_RANGE_STEP = 17, P_RANGE_FINISH = 18, P_VALUE_COPY = 19, P_ARK = 20
};

// one evaluation frame: strided views of the current / next LDE rows and of the periodic values
struct Frame {
    const fp *cur_p, *next_p, *per_p; // element c at cur_p[c * n], per_p[c * pcycle]
    size_t n;
    unsigned pcycle;                  // rows per periodic cycle (1024 for the composite AIR)
    __device__ __forceinline__ fp cur(int c) const { return cur_p[(size_t)c * n]; }
    __device__ __forceinline__ fp next(int c) const { return next_p[(size_t)c * n]; }
    __device__ __forceinline__ fp pv(int c) const { return per_p[(size_t)c * pcycle]; }
};

// ---- accumulators ---------------------------------------------------------------------------------
// Parity/debug: every result slot is kept, in global memory: out[i * n + j] += flag * value.
struct AccAll {
    fp *out; // points at element j of slot 0
    size_t n;
    __device__ __forceinline__ void add(int i, fp flag, fp val) { out[(size_t)i * n] = fp_add(out[(size_t)i * n], fp_mul(flag, val)); }
};
__device__ __forceinline__ fp c_not(fp a) { return fp_sub(FP_ONE, a); }
__device__ __forceinline__ fp c_is_binary(fp a) { return fp_sub(fp_sqr(a), a); }

// sum_j m[j] * x[j] for a row of 14 constants: seven products per fold of the 128-bit accumulator, one reduction
__device__ __forceinline__ fp dot14_rows(const fp *__restrict__ m, const fp (&x)[14]) {
    Acc128 a = acc_zero();
#pragma unroll
    for (int j = 0; j < 7; j++) acc_mad(a, m[j], x[j]);
    acc_fold(a);
#pragma unroll
    for (int j = 7; j < 14; j++) acc_mad(a, m[j], x[j]);
    acc_fold(a);
    return acc_reduce(a);
}

// ---- Rescue round gadget (rescue.rs:269-300) on the 14-register window starting at `reg`; the same 14
// differences feed up to two (result base, flag) pairs.
template <class Acc>
__device__ __forceinline__ void enforce_round(Acc &acc, const Frame &f, int reg, int res_a, fp flag_a, int res_b, fp flag_b, bool two) {
    fp cube[14], d[14];
#pragma unroll
    for (int j = 0; j < 14; j++) {
        cube[j] = fp_cube(f.cur(reg + j));
        d[j] = fp_sub(f.next(reg + j), f.pv(P_ARK + 14 + j));
    }
#pragma unroll 1
    for (int i = 0; i < 14; i++) {
        // rows of the two matrices against the 14 cubes / differences: 128-bit accumulation, one reduction per row (dot14 below)
        const fp s1 = fp_add(f.pv(P_ARK + i), dot14_rows(c_mds + i * 14, cube));
        const fp s2 = dot14_rows(c_inv_mds + i * 14, d);
        const fp diff = fp_sub(fp_cube(s2), s1);
        acc.add(res_a + i, flag_a, diff);
        if (two) acc.add(res_b + i, flag_b, diff);
    }
}

// ---- curve gadgets (ecc.rs:73-172).  F_p6 products are real calls to keep the kernel inside the
// instruction cache: the five gadgets contain 69 of them.
#ifdef CS_EC_NOINLINE
__device__ __noinline__ Fp6 mul6(const Fp6 a, const Fp6 b) { return fp6_mul(a, b); }
#else
__device__ __forceinline__ Fp6 mul6(const Fp6 &a, const Fp6 &b) { return fp6_mul(a, b); }
#endif
// the parity/debug kernel keeps everything in one launch and must stay compact: always a call there
__device__ __noinline__ Fp6 mul6_call(const Fp6 a, const Fp6 b) { return fp6_mul(a, b); }

template <bool CALL>
__device__ __forceinline__ Fp6 MUL6_SEL(const Fp6 &a, const Fp6 &b) {
    if constexpr (CALL) return mul6_call(a, b);
    else return mul6(a, b);
}

struct Point { Fp6 x, y, z; };

__device__ __forceinline__ Fp6 load6(const Frame &f, int reg, bool next) {
    Fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = next ? f.next(reg + i) : f.cur(reg + i);
    return r;
}
__device__ __forceinline__ Fp6 const6(const fp *p) { return fp6_load(p); }

template <bool CALL>
__device__ __forceinline__ Point ec_double(const Point &p) {
#define mul6 MUL6_SEL<CALL> // complete doubling, ecc.rs:177-242
    const Fp6 b3 = const6(c_b3);
#ifdef CS_EC_SQR
    Fp6 t0 = fp6_sqr(p.x), t1 = fp6_sqr(p.y), t2 = fp6_sqr(p.z);
#else
    Fp6 t0 = mul6(p.x, p.x), t1 = mul6(p.y, p.y), t2 = mul6(p.z, p.z);
#endif
    Fp6 t3 = fp6_dbl(mul6(p.x, p.y));
    Fp6 z3 = fp6_dbl(mul6(p.x, p.z));
    Fp6 y3 = fp6_add(z3, mul6(b3, t2));
    Fp6 x3 = fp6_sub(t1, y3);
    y3 = fp6_add(t1, y3);
    y3 = mul6(x3, y3);
    x3 = mul6(t3, x3);
    z3 = mul6(b3, z3);
    t3 = fp6_add(fp6_sub(t0, t2), z3);
    t0 = fp6_add(fp6_add(fp6_dbl(t0), t0), t2);
    y3 = fp6_add(y3, mul6(t0, t3));
    t2 = fp6_dbl(mul6(p.y, p.z));
    x3 = fp6_sub(x3, mul6(t2, t3));
    z3 = fp6_dbl(fp6_dbl(mul6(t2, t1)));
    return {x3, y3, z3};
#undef mul6
}
template <bool CALL>
__device__ __forceinline__ Point ec_add_mixed(const Point &p, const Fp6 &qx, const Fp6 &qy) {
#define mul6 MUL6_SEL<CALL> // ecc.rs:330-404
    const Fp6 b3 = const6(c_b3);
    Fp6 t0 = mul6(p.x, qx), t1 = mul6(p.y, qy);
    Fp6 t3 = fp6_sub(mul6(fp6_add(qx, qy), fp6_add(p.x, p.y)), fp6_add(t0, t1));
    Fp6 t4 = fp6_add(mul6(qx, p.z), p.x);
    Fp6 t5 = fp6_add(mul6(qy, p.z), p.y);
    Fp6 z3 = fp6_add(mul6(p.z, b3), t4);
    Fp6 x3 = fp6_sub(t1, z3);
    z3 = fp6_add(t1, z3);
    Fp6 y3 = mul6(x3, z3);
    t1 = fp6_add(fp6_add(fp6_dbl(t0), t0), p.z);
    t4 = fp6_add(mul6(t4, b3), fp6_sub(t0, p.z));
    y3 = fp6_add(y3, mul6(t1, t4));
    x3 = fp6_sub(mul6(t3, x3), mul6(t5, t4));
    z3 = fp6_add(mul6(t5, z3), mul6(t3, t1));
    return {x3, y3, z3};
#undef mul6
}
template <bool CALL>
__device__ __forceinline__ Point ec_add(const Point &p, const Point &q) {
#define mul6 MUL6_SEL<CALL> // ecc.rs:244-328
    const Fp6 b3 = const6(c_b3);
    Fp6 t0 = mul6(p.x, q.x), t1 = mul6(p.y, q.y), t2 = mul6(p.z, q.z);
    Fp6 t3 = fp6_sub(mul6(fp6_add(p.x, p.y), fp6_add(q.x, q.y)), fp6_add(t0, t1));
    Fp6 t4 = fp6_sub(mul6(fp6_add(p.x, p.z), fp6_add(q.x, q.z)), fp6_add(t0, t2));
    Fp6 t5 = fp6_sub(mul6(fp6_add(p.y, p.z), fp6_add(q.y, q.z)), fp6_add(t1, t2));
    Fp6 z3 = fp6_add(mul6(b3, t2), t4);
    Fp6 x3 = fp6_sub(t1, z3);
    z3 = fp6_add(t1, z3);
    Fp6 y3 = mul6(x3, z3);
    t1 = fp6_add(fp6_add(fp6_dbl(t0), t0), t2);
    t4 = fp6_add(mul6(b3, t4), fp6_sub(t0, t2));
    y3 = fp6_add(y3, mul6(t1, t4));
    x3 = fp6_sub(mul6(t3, x3), mul6(t5, t4));
    z3 = fp6_add(mul6(t5, z3), mul6(t3, t1));
    return {x3, y3, z3};
#undef mul6
}

// doubling + conditional mixed addition constraints for the point at registers [reg, reg+19)
template <class Acc>
__device__ __forceinline__ void enforce_scalar_mult_step(Acc &acc, const Frame &f, int reg, const Fp6 &qx, const Fp6 &qy, fp doubling, fp addition) {
    const Point p = {load6(f, reg, false), load6(f, reg + 6, false), load6(f, reg + 12, false)};
    const fp bit = f.cur(reg + 18);
    {   // enforce_point_doubling ecc.rs:73-98
        const Point d = ec_double<true>(p);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            acc.add(reg + i, doubling, fp_sub(f.next(reg + i), d.x.c[i]));
            acc.add(reg + 6 + i, doubling, fp_sub(f.next(reg + 6 + i), d.y.c[i]));
            acc.add(reg + 12 + i, doubling, fp_sub(f.next(reg + 12 + i), d.z.c[i]));
        }
        acc.add(reg + 18, doubling, c_is_binary(bit));
    }
    {   // enforce_point_addition_mixed ecc.rs:102-138: next = bit * (p + q) + (1 - bit) * p
        const Point a = ec_add_mixed<true>(p, qx, qy);
        const fp nb = c_not(bit);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            acc.add(reg + i, addition, fp_sub(f.next(reg + i), fp_add(fp_mul(bit, a.x.c[i]), fp_mul(nb, p.x.c[i]))));
            acc.add(reg + 6 + i, addition, fp_sub(f.next(reg + 6 + i), fp_add(fp_mul(bit, a.y.c[i]), fp_mul(nb, p.y.c[i]))));
            acc.add(reg + 12 + i, addition, fp_sub(f.next(reg + 12 + i), fp_add(fp_mul(bit, a.z.c[i]), fp_mul(nb, p.z.c[i]))));
        }
        acc.add(reg + 18, addition, fp_sub(bit, f.next(reg + 18)));
    }
}

// src/merkle/update/air.rs:291-369 without its two enforce_round calls (done by the caller per window)
template <class Acc>
__device__ __forceinline__ void merkle_auth_rest(Acc &acc, const Frame &f, int base, fp tx_hash, fp hash_input, fp hash_flag) {
    const fp hash_copy = fp_mul(tx_hash, c_not(fp_add(hash_flag, hash_input)));
    const fp hash_init = fp_mul(tx_hash, hash_input);
    const fp bit = f.next(base + 14), not_bit = c_not(bit);
    acc.add(base + 14, tx_hash, c_is_binary(bit));
#pragma unroll 1
    for (int k = 0; k < 2; k++) {
        const int b = base + 15 * k;
#pragma unroll 1
        for (int i = 0; i < 7; i++) {
            const fp ci = f.cur(b + i), d = fp_sub(ci, f.next(b + i));
            acc.add(b + i, hash_copy, d);
            acc.add(b + i, hash_init, fp_mul(not_bit, d));
            acc.add(b + 7 + i, hash_init, fp_mul(bit, fp_sub(ci, f.next(b + 7 + i))));
        }
    }
#pragma unroll 1
    for (int i = 0; i < 7; i++) acc.add(base + i, hash_init, fp_mul(bit, fp_sub(f.next(base + 15 + i), f.next(base + i))));
#pragma unroll 1
    for (int i = 7; i < 14; i++) acc.add(base + i, hash_init, fp_mul(not_bit, fp_sub(f.next(base + 15 + i), f.next(base + i))));
}

// All 115 transition constraints at one point (src/air.rs:114-173, :383-610).
template <class Acc>
__device__ __forceinline__ void evaluate_transition(Acc &acc, const Frame &f) {
    const fp setup = f.pv(P_SETUP), tx_hash = f.pv(P_MERKLE), hash_input = f.pv(P_HASH_INPUT), finish = f.pv(P_FINISH), hash_flag = f.pv(P_HASH);
    const fp schnorr_mask = f.pv(P_SCHNORR), scalar_mult = f.pv(P_SCALAR_MULT), doubling = f.pv(P_DOUBLING), schnorr_hash = f.pv(P_SCHNORR_HASH);
    const fp range_flag = f.pv(P_RANGE_STEP), range_finish = f.pv(P_RANGE_FINISH), copy_values = f.pv(P_VALUE_COPY);
    const fp copy_hash = fp_mul(c_not(schnorr_hash), schnorr_mask);
    const fp final_add = fp_mul(c_not(scalar_mult), schnorr_mask);
    const fp addition = fp_mul(c_not(doubling), scalar_mult);

    // Rescue windows: merkle::init (setup flag, shifted result bases, src/merkle/init/air.rs:166-201) and
    // merkle::update (hash flag, src/merkle/update/air.rs:316-322) share their register windows.
    enforce_round(acc, f, S_INIT, S_INIT, setup, S_INIT, hash_flag, true);
    enforce_round(acc, f, S_UPD, S_UPD - 1, setup, S_UPD, hash_flag, true);
    enforce_round(acc, f, R_INIT, R_INIT - 1, setup, R_INIT, hash_flag, true);
    enforce_round(acc, f, R_UPD, R_UPD - 2, setup, R_UPD, hash_flag, true);
    enforce_round(acc, f, 42, 42, schnorr_hash, 0, 0, false); // src/schnorr/air.rs:488-494

    // transaction setup (src/air.rs:406-453)
#pragma unroll 1
    for (int i = 0; i < 12; i++) {
        acc.add(VALUE_RES + i, setup, fp_sub(f.cur(S_INIT + i), f.cur(S_UPD + i)));
        acc.add(VALUE_RES + 12 + i, setup, fp_sub(f.cur(R_INIT + i), f.cur(R_UPD + i)));
    }
    acc.add(VALUE_RES + 24, setup, fp_sub(f.cur(R_INIT + 13), f.cur(R_UPD + 13)));
    const fp s_spent = fp_sub(f.cur(S_INIT + 12), f.cur(S_UPD + 12));
    acc.add(BALANCE_RES, setup, fp_sub(s_spent, fp_sub(f.cur(R_UPD + 12), f.cur(R_INIT + 12))));
    acc.add(NONCE_UPD_RES, setup, fp_sub(f.cur(S_UPD + 13), fp_add(f.cur(S_INIT + 13), FP_ONE)));
    // key / delta / sigma / nonce copies (src/air.rs:456-529; result indices alias as in the reference)
#pragma unroll 1
    for (int o = 0; o < 12; o++) {
        const fp sk = f.next(S_KEY + o), rk = f.next(R_KEY + o);
        acc.add(S_KEY_RES + o, setup, fp_sub(sk, f.cur(S_INIT + o)));
        acc.add(R_KEY_RES + o, setup, fp_sub(rk, f.cur(R_INIT + o)));
        acc.add(S_KEY_RES + o, copy_values, fp_sub(sk, f.cur(S_KEY + o)));
        acc.add(R_KEY_RES + o, copy_values, fp_sub(rk, f.cur(R_KEY + o)));
    }
    acc.add(DELTA_COPY_RES, setup, fp_sub(f.next(DELTA_COPY), s_spent));
    acc.add(SIGMA_COPY_RES, setup, fp_sub(f.next(SIGMA_COPY), f.cur(S_UPD + 12)));
    acc.add(NONCE_COPY_RES, setup, fp_sub(f.next(NONCE_COPY), f.cur(S_INIT + 13)));
    acc.add(DELTA_COPY_RES, copy_values, fp_sub(f.next(DELTA_COPY), f.cur(DELTA_COPY)));
    acc.add(SIGMA_COPY_RES, copy_values, fp_sub(f.next(SIGMA_COPY), f.cur(SIGMA_COPY)));
    acc.add(NONCE_COPY_RES, copy_values, fp_sub(f.next(NONCE_COPY), f.cur(NONCE_COPY)));

    // merkle::update (src/merkle/update/air.rs:215-289)
    merkle_auth_rest(acc, f, S_INIT, tx_hash, hash_input, hash_flag);
    merkle_auth_rest(acc, f, R_INIT, tx_hash, hash_input, hash_flag);
    const fp not_finish = c_not(finish);
#pragma unroll 1
    for (int i = 0; i < 7; i++) {
        const fp nr = f.next(PREV_ROOT + i), cr = f.cur(PREV_ROOT + i);
        acc.add(PREV_ROOT + i, not_finish, fp_sub(nr, cr));
        acc.add(PREV_ROOT + i, finish, fp_sub(nr, f.next(R_UPD + i)));
        acc.add(INT_ROOT_RES + i, finish, fp_sub(f.cur(S_UPD + i), f.cur(R_INIT + i)));
        acc.add(PREV_MATCH_RES + i, finish, fp_sub(f.next(S_INIT + i), cr));
    }

    // schnorr (src/schnorr/air.rs:394-531) on registers / results [0, 56)
    {
        const Fp6 gx = const6(c_generator), gy = const6(c_generator + 6);
        enforce_scalar_mult_step(acc, f, 0, gx, gy, doubling, addition);
        const Fp6 px = load6(f, S_KEY, true), py = load6(f, S_KEY + 6, true); // pkey = next[S_KEY..], src/air.rs:575
        enforce_scalar_mult_step(acc, f, 19, px, py, doubling, addition);
    }
#pragma unroll 1
    for (int i = 0; i < 4; i++) { // h-limb double-and-add, accumulator copies (:451-484)
        const fp dflag = f.pv(P_DIGEST + i);
        const fp c = f.cur(41 - i), nx = f.next(41 - i);
        acc.add(41 - i, fp_mul(dflag, doubling), fp_sub(nx, fp_add(fp_dbl(c), f.next(37))));
        acc.add(41 - i, fp_mul(c_not(dflag), doubling), fp_sub(c, nx));
        acc.add(38 + i, addition, fp_sub(f.cur(38 + i), f.next(38 + i)));
        acc.add(38 + i, final_add, fp_sub(f.cur(38 + i), f.cur(42 + i))); // h == hash output (:521-530)
    }
    // enforce_hash_copy (:309-330) with the internal inputs of src/air.rs:543-565
#pragma unroll 1
    for (int i = 0; i < 7; i++) {
        acc.add(42 + i, copy_hash, fp_sub(f.cur(42 + i), f.next(42 + i)));
        fp inp = 0;
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
            const int m = k * 7 + i;
            const fp cell = m < 12 ? f.next(S_KEY + m) : m < 24 ? f.next(R_KEY + m - 12) : m == 24 ? f.next(DELTA_COPY) : m == 25 ? f.next(NONCE_COPY) : 0;
            inp = fp_add(inp, fp_mul(f.pv(P_HASH_INTERNAL + k), cell));
        }
        acc.add(49 + i, copy_hash, fp_sub(f.next(49 + i), inp));
    }
    {   // final addition S + h*P with X reduced to affine (ecc.rs:146-172)
        const Point s = {load6(f, 0, false), load6(f, 6, false), load6(f, 12, false)};
        const Point hp = {load6(f, 19, false), load6(f, 25, false), load6(f, 31, false)};
        const Point r = ec_add<true>(s, hp);
        const Fp6 xz = mul6_call(load6(f, 0, true), r.z);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            acc.add(i, final_add, fp_sub(xz.c[i], r.x.c[i]));
            acc.add(6 + i, final_add, fp_sub(f.next(6 + i), r.y.c[i]));
            acc.add(12 + i, final_add, fp_sub(f.next(12 + i), r.z.c[i]));
        }
    }
    // range proofs (src/air.rs:583-609); SIGMA_RANGE_RES re-checks the delta registers, as in the reference
    {
        const fp db = f.next(DELTA_BIT), sb = f.next(SIGMA_BIT);
        acc.add(DELTA_ACC, range_flag, fp_sub(f.next(DELTA_ACC), fp_add(fp_dbl(f.cur(DELTA_ACC)), db)));
        acc.add(DELTA_BIT, range_flag, c_is_binary(db));
        acc.add(SIGMA_ACC, range_flag, fp_sub(f.next(SIGMA_ACC), fp_add(fp_dbl(f.cur(SIGMA_ACC)), sb)));
        acc.add(SIGMA_BIT, range_flag, c_is_binary(sb));
        const fp dr = fp_sub(f.next(DELTA_ACC), f.next(DELTA_COPY));
        acc.add(DELTA_RANGE_RES, range_finish, dr);
        acc.add(SIGMA_RANGE_RES, range_finish, dr);
    }
}

__device__ __forceinline__ Frame make_frame(const CeParams &p, unsigned kk, size_t j) {
    const size_t n = (size_t)1 << p.log_n;
    const fp *base = p.lde + (size_t)kk * 94 * n;
    Frame f;
    f.n = n;
    f.cur_p = base + j;
    f.next_p = base + ((j + 1) & (n - 1)); // LDE row i + blowup = (k, j + 1)
    f.per_p = p.ptab + (size_t)(p.k0 + kk) * 48 * 1024 + (j & 1023);
    f.pcycle = 1024;
    return f;
}

// grid = (n / NT, nk)
__global__ __launch_bounds__(NT) void k_eval_transitions(CeParams p) {
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)NT + threadIdx.x;
    const unsigned kk = blockIdx.y;
    const Frame f = make_frame(p, kk, j);
    AccAll acc{p.out + (size_t)kk * 115 * n + j, n}; // caller zero-fills
    evaluate_transition(acc, f);
}

// =====================================================================================================
// Production path: fused evaluation.  The work is split into four launches ("parts") so that each has a
// small instruction footprint and register budget:  ROUNDS (the five Rescue windows), EC0 / EC1 (double +
// mixed-add gadgets of s*G and h*P), REST (final addition, every linear constraint, boundary terms).
// Each part folds its constraints straight into the random linear combination
//     sum_i (alpha_i + beta_i x^adj_g(i)) * flag * value_i
// using one 128-bit lazy accumulator per (gadget, flag) section, multiplies by 1/Z(x) and adds its share to
// the output.  Exact arithmetic: the sum of the parts equals the reference's merged evaluation.
constexpr int FNT = 256;
#ifndef CS_LINA_UNROLL
#define CS_LINA_UNROLL 1
#endif
#ifndef CS_EC_CALL
#define CS_EC_CALL false // true: F_p6 products of the fused EC parts as calls (smaller code): measured slower, 2.57 vs 2.15 ms per part
#endif
#ifndef CS_LINS_UNROLL_A
#define CS_LINS_UNROLL_A 1
#endif
#ifndef CS_LINS_UNROLL_B
#define CS_LINS_UNROLL_B 7 // split linear groups, measured: A rolled 0.57 ms (unrolled 1.26); B with both loops unrolled 0.95 ms (rolled 1.14)
#endif
#ifndef CS_LINS_UNROLL_BLK
#define CS_LINS_UNROLL_BLK 2
#endif
#ifndef CS_ROUNDS_INV_UNROLL
#define CS_ROUNDS_INV_UNROLL 1
#endif
#ifndef CS_ROUNDS_UNROLL
#define CS_ROUNDS_UNROLL 1
#endif
constexpr int ROUNDS_UNROLL = CS_ROUNDS_UNROLL;
enum { PART_ROUNDS = 0, PART_DBL0, PART_ADD0, PART_DBL1, PART_ADD1, PART_FINAL, PART_LIN_A, PART_LIN_B, PART_LIN_C, NUM_PARTS };

// M = number of coefficient sets merged in one pass (1 for base-field proofs; 2 / 3 = the components of a quadratic / cubic
// extension proof, whose coefficients multiply the same base-field constraint values: the values are computed once).
// Tables every lane reads at the same address, written before the launch and never during it (coefficients, folded round
// tables): addressed through the constant address space, so that they stay scalar loads (s_load) whatever else the kernel does
// to memory (inline-asm waits, LDS-DMA) -- as vector loads they would also sit on the same in-order counter as the LDS-DMA.
#define CS_CONST __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const CS_CONST T *as_const(const T *p) { return (const CS_CONST T *)(uintptr_t)p; }

template <int M>
struct Fused {
    const CS_CONST fp *coefs; // uniform: M blocks of CE_COEF_WORDS (alpha[115] | beta[115] | boundary)
    const fp *xp;    // LDS [5][FNT]: x^adj_g of this lane's point
    Acc128 s[M];
    int cnt;
    fp total[M];
    __device__ __forceinline__ fp coef(int c, int i) const {
        const CS_CONST fp *a = coefs + c * CE_COEF_WORDS;
        return fp_add(a[i], fp_mul(a[115 + i], xp[tx_degree_group(i) * FNT + threadIdx.x]));
    }
    __device__ __forceinline__ void begin() {
#pragma unroll
        for (int c = 0; c < M; c++) s[c] = acc_zero();
        cnt = 0;
    }
    __device__ __forceinline__ void term(int i, fp v) {
        ++cnt;
#pragma unroll
        for (int c = 0; c < M; c++) {
            acc_mad(s[c], coef(c, i), v);
            if (cnt == 7) acc_fold(s[c]);
        }
        if (cnt == 7) cnt = 0;
    }
    __device__ __forceinline__ void end(fp flag) {
#pragma unroll
        for (int c = 0; c < M; c++) {
            acc_fold(s[c]);
            total[c] = fp_add(total[c], fp_mul(flag, acc_reduce(s[c])));
        }
    }
    __device__ __forceinline__ void add(int c, fp v) { total[c] = fp_add(total[c], v); }
};

// sum_j m[j] * x[j] for a row of 14 uniform constants, one reduction
__device__ __forceinline__ fp dot14(const fp *__restrict__ m, const fp (&x)[14]) {
    Acc128 a = acc_zero();
#pragma unroll
    for (int j = 0; j < 7; j++) acc_mad(a, m[j], x[j]);
    acc_fold(a);
#pragma unroll
    for (int j = 7; j < 14; j++) acc_mad(a, m[j], x[j]);
    acc_fold(a);
    return acc_reduce(a);
}

// The same dot product against a row of 14 UNIFORM constants that were pre-split into three 21-bit limbs (4 dwords per
// constant, the 4th unused).  A 32-bit half of x times a limb is < 2^53, so 14 of them accumulate in a plain 64-bit
// v_mad_u64_u32 chain with no carry handling at all: 6 multiply-adds per term and nothing else (the carry-propagating
// 128-bit accumulation above costs ~21 VALU instructions per term).  The six column sums are recombined once:
//   V = sum_l 2^(21 l) (c0l + 2^32 c1l) < 14 p^2 < 2^128.
__device__ __forceinline__ fp dot14l(const CS_CONST uint32_t *m, const fp (&x)[14]) {
    uint64_t c00 = 0, c01 = 0, c02 = 0, c10 = 0, c11 = 0, c12 = 0;
#pragma unroll
    for (int j = 0; j < 14; j++) {
        const uint32_t x0 = (uint32_t)x[j], x1 = (uint32_t)(x[j] >> 32);
        const uint32_t m0 = m[4 * j], m1 = m[4 * j + 1], m2 = m[4 * j + 2];
        c00 = mad_u64_u32(x0, m0, c00); c01 = mad_u64_u32(x0, m1, c01); c02 = mad_u64_u32(x0, m2, c02);
        c10 = mad_u64_u32(x1, m0, c10); c11 = mad_u64_u32(x1, m1, c11); c12 = mad_u64_u32(x1, m2, c12);
    }
    typedef unsigned __int128 u128;
    const u128 v = ((u128)c00 + ((u128)c10 << 32)) + (((u128)c01 + ((u128)c11 << 32)) << 21) + (((u128)c02 + ((u128)c12 << 32)) << 42);
    Acc128 a{(uint64_t)v, (uint64_t)(v >> 64)};
    acc_fold(a);
    return acc_reduce(a);
}
__device__ __forceinline__ void split_limbs(fp v, uint32_t *out4) {
    out4[0] = (uint32_t)v & 0x1fffffu; out4[1] = (uint32_t)(v >> 21) & 0x1fffffu; out4[2] = (uint32_t)(v >> 42); out4[3] = 0;
}

// the five Rescue windows and the layout of CeParams::rtab: rounds_layout.h
__constant__ RoundWindow c_windows[5] = CS_ROUND_WINDOWS_INIT;
__constant__ int c_window_groups[5][2][3] = CS_WINDOW_GROUPS_INIT;
constexpr size_t ROUNDS_LDS = MT_BYTES + (FNT / 64) * 64 * mdsmfma::ROW_BYTES; // table + one staging image per wave
#ifdef CS_ROUNDS_MFMA
constexpr size_t ROUNDS_DYN_LDS = ROUNDS_LDS;
#else
constexpr size_t ROUNDS_DYN_LDS = 0;
#endif

// The forward half of the round gadget is linear in cube(cur): sum_i c_i (MDS cube + ark1)_i = (MDS^T c) . cube + c . ark1.
// With c_i = alpha_i + beta_i x^adj_g(i) this is one 14-term dot product per (alpha | beta restricted to a degree group) instead of
// one per result slot.  k_rounds_setup folds the coefficients of one proof through MDS (U) and through the round constants, whose
// extension has period 8 in j on every coset (A[k][j & 7]).  Exact arithmetic: the merged value is unchanged.
__global__ void k_rounds_setup(const fp *__restrict__ coef, const fp *__restrict__ ptab, fp *__restrict__ rtab) {
    coef += (size_t)blockIdx.y * CE_COEF_WORDS; // grid.y = coefficient set
    rtab += (size_t)blockIdx.y * CE_RTAB_WORDS;
    if (blockIdx.x >= RT_SECTIONS) { // the constant matrix of the inverse half as a matrix-core table (region zero-filled by the launcher)
        if (threadIdx.x < 16) mdsmfma::build_table_entry((uint8_t *)(rtab + RT_MT), c_inv_mds, 14, blockIdx.x - RT_SECTIONS, threadIdx.x);
        return;
    }
    const int sec = blockIdx.x, wdx = sec >> 3, fs = (sec >> 2) & 1, slot = sec & 3;
    const RoundWindow w = c_windows[wdx];
    const int res = fs ? w.res_b : w.res_a;
    const int grp = slot ? c_window_groups[wdx][fs][slot - 1] : -2;
    const bool used = (fs == 0 || w.flag_b >= 0) && grp != -1;
    __shared__ fp gam[14];
    if (threadIdx.x < 14) {
        const int i = res + threadIdx.x;
        gam[threadIdx.x] = !used ? 0 : slot == 0 ? coef[i] : (tx_degree_group(i) == grp ? coef[115 + i] : 0);
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 14) rtab[RT_G + sec * 14 + t] = gam[t];
    if (t < 14) {
        fp u = 0;
        for (int i = 0; i < 14; i++) u = fp_add(u, fp_mul(gam[i], c_mds[i * 14 + t]));
        split_limbs(u, (uint32_t *)(rtab + RT_UL) + (sec * 14 + t) * 4);
    }
    if (sec == 0) // limb form of the inverse matrix (VALU variant of the inverse half)
        for (int e = t; e < 196; e += blockDim.x) split_limbs(c_inv_mds[e], (uint32_t *)(rtab + RT_ML) + e * 4);
    if (t < 64) {
        const int k = t >> 3, r = t & 7;
        fp a = 0;
        for (int i = 0; i < 14; i++) a = fp_add(a, fp_mul(gam[i], ptab[((size_t)k * 48 + P_ARK + i) * 1024 + r]));
        rtab[RT_A + sec * 64 + t] = a;
    }
}

// Window image of one wave: 14 columns x 66 rows (rows j0 .. j0 + 65 of the wave's 64 points; 65 are used), filled by LDS-DMA
// (global_load_lds_dwordx4: lane l < 33 moves rows j0 + 2l, j0 + 2l + 1 of a column straight into LDS, no VGPR staging).  A lane's
// current row is element `lane`, its next row element `lane + 1`: every LDE cell is fetched from memory once.
constexpr int RW_ROWS = 66, RW_IMG = 14 * RW_ROWS;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
// rows: this lane's source rows (lane 32 of the last wave of a coset wraps to rows 0, 1); img: the wave's image (wave-uniform)
__device__ __forceinline__ void rounds_fetch_window(const fp *rows, size_t n, int reg, int lane, fp *img) {
    if (lane < 33) {
#pragma unroll
        for (int j = 0; j < 14; j++)
            __builtin_amdgcn_global_load_lds((glb_void *)(rows + (size_t)(reg + j) * n), (lds_void *)(img + j * RW_ROWS), 16, 0, 0);
    }
}
template <int M>
__device__ __forceinline__ void fused_rounds(Fused<M> &acc, const Frame &f, const fp *__restrict__ rtab, unsigned k, unsigned jr, const uint8_t *mtab,
                                             uint8_t *stage, const fp *ark2_lds, const fp *atab_lds, fp *img, const fp *colbase) {
    const int lane = threadIdx.x & 63;
    const fp *ark2 = ark2_lds + jr * 14;
    const fp flags[4] = {f.pv(P_SETUP), f.pv(P_HASH), f.pv(P_SCHNORR_HASH), fp_add(f.pv(P_SETUP), f.pv(P_HASH))};
    const fp *atab = atab_lds + jr; // [M][RT_SECTIONS][8]: this coset's A[section][j & 7]
    const CS_CONST uint32_t *ul = as_const((const uint32_t *)(rtab + RT_UL));
    // source rows of this lane's 16-byte piece: rows j0 + 2 lane, + 1 (j0 = the wave's first row); rows n, n + 1 wrap to 0, 1
    const fp *rows = f.cur_p + lane;
    if (lane == 32 && ((size_t)(f.cur_p - lane + 64 - colbase) & (f.n - 1)) == 0) rows -= f.n;
    rounds_fetch_window(rows, f.n, c_windows[0].reg, lane, img);
#pragma unroll 1
    for (int wdx = 0; wdx < 5; wdx++) {
        const RoundWindow w = c_windows[wdx];
        // register budget (168 VGPRs at 3 waves per SIMD): the 14 round constants of the inverse half are re-read per window
        // from LDS (their extension has period 8 in j: 8 x 14 values per workgroup) and the cubes of the forward half are
        // formed after the inverse half, not before.
        // The window's cells arrive by LDS-DMA (issued during the previous window's forward half); the DMA is the only vector
        // memory operation of the loop (tables: scalar loads or LDS), so this wait concerns nothing else.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fp d[14];
#pragma unroll
        for (int j = 0; j < 14; j++) d[j] = fp_sub(img[j * RW_ROWS + lane + 1], ark2[j]);
        // inverse half: cube(INV_MDS (next - ark2))_i against the coefficients of both flag sets.  Default: limb dot products on
        // the vector ALU.  With -DCS_ROUNDS_MFMA the 14x14 product runs on the matrix cores for the 64 points of the wave
        // (mds_mfma.cuh; the staging image is private to the wave, whose LDS operations execute in order, so no workgroup
        // barrier is involved) -- bit-identical, and measured at the SAME kernel time on MI355X as the limb products when both staged
        // their rows through registers (6.65 ms): the recombination of the byte diagonals and the serialisation of MFMA and dependent
        // VALU work inside a wave eat the gain.  Since the LDS-DMA window images the variant is slower (9.1 vs 5.8 ms): its tables and
        // staging images no longer fit beside them at three workgroups per CU.
#ifdef CS_ROUNDS_MFMA
        fp yv[14];
        {
            mdsmfma::stage_vector(stage, lane, d);
            __builtin_amdgcn_wave_barrier();
            mdsmfma::BFrags bf;
            mdsmfma::load_bfrags(bf, stage, lane);
            __builtin_amdgcn_wave_barrier();
            uint64_t *img = (uint64_t *)stage;
            constexpr int RW = mdsmfma::ROW_BYTES / 8;
#pragma unroll
            for (int T = 0; T < 7; T++) { // lane (n, g) gets output 2T + g of points n and 32 + n: hand them back to the points' own lanes
                fp y2[2];
                mdsmfma::tile_product(mtab, 7, T, bf, lane, y2);
                img[(lane & 31) * RW + 2 * T + (lane >> 5)] = y2[0];
                img[(32 + (lane & 31)) * RW + 2 * T + (lane >> 5)] = y2[1];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 14; i++) yv[i] = img[lane * RW + i];
            __builtin_amdgcn_wave_barrier();
        }
#endif
        Acc128 sa[M], sb[M];
#pragma unroll
        for (int c = 0; c < M; c++) sa[c] = sb[c] = acc_zero();
#ifdef CS_ROUNDS_MFMA
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const fp s2 = fp_cube(yv[i]);
#else
        const CS_CONST uint32_t *ml = as_const((const uint32_t *)(rtab + RT_ML));
#pragma unroll CS_ROUNDS_INV_UNROLL
        for (int i = 0; i < 14; i++) {
            const fp s2 = fp_cube(dot14l(ml + i * 56, d));
#endif
#pragma unroll
            for (int c = 0; c < M; c++) {
                acc_mad(sa[c], acc.coef(c, w.res_a + i), s2);
                if (w.flag_b >= 0) acc_mad(sb[c], acc.coef(c, w.res_b + i), s2);
                if (i == 6) { acc_fold(sa[c]); acc_fold(sb[c]); }
            }
        }
#pragma unroll
        for (int c = 0; c < M; c++) { acc_fold(sa[c]); acc_fold(sb[c]); }
        // forward half through the folded vectors
        fp cube[14];
#pragma unroll
        for (int j = 0; j < 14; j++) cube[j] = fp_cube(img[j * RW_ROWS + lane]);
        if (wdx < 4) { // the image is free again: fetch the next window behind the forward half
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            rounds_fetch_window(rows, f.n, c_windows[wdx + 1].reg, lane, img);
        }
#pragma unroll 1
        for (int fs = 0; fs < 2; fs++) {
            if (fs == 1 && w.flag_b < 0) break;
            const int sec = (wdx * 2 + fs) * 4;
            const int fl = fs ? w.flag_b : w.flag_a;
            const fp flag = fl == 0 ? flags[0] : fl == 1 ? flags[1] : fl == 2 ? flags[2] : flags[3];
#pragma unroll
            for (int c = 0; c < M; c++) {
                const CS_CONST uint32_t *ulc = ul + (size_t)c * CE_RTAB_WORDS * 2;
                const fp *atc = atab + c * RT_SECTIONS * 8;
                fp fwd = fp_add(dot14l(ulc + sec * 56, cube), atc[sec * 8]);
#pragma unroll 1
                for (int sl = 1; sl < 4; sl++) {
                    const int g = c_window_groups[wdx][fs][sl - 1];
                    if (g < 0) break;
                    const fp v = fp_add(dot14l(ulc + (sec + sl) * 56, cube), atc[(sec + sl) * 8]);
                    fwd = fp_add(fwd, fp_mul(acc.xp[g * FNT + threadIdx.x], v));
                }
                const fp inv_side = acc_reduce(fs ? sb[c] : sa[c]);
                acc.add(c, fp_mul(flag, fp_sub(inv_side, fwd)));
            }
        }
    }
}

// =====================================================================================================
// Split evaluation of the Rescue windows.  Every Rescue term flag(x) * (forward_i - inverse_i^3) is a polynomial of degree
// <= 3 (n - 1) + (n - 1) < 4n (trace columns: degree < n; periodic flags and round constants: degree < n), whatever degree class its
// result slot is declared in.  The merged contribution of the windows is therefore
//     R_alpha(x) + sum_g x^adj_g R_beta,g(x),   R_alpha = sum_i alpha_i F_i,  R_beta,g = sum_{i in g} beta_i F_i   (g = 0, 1, 2)
// with four polynomials of degree < 4n: they are evaluated on the 4n-point sub-domain of the even cosets only (half the work of
// the windows), interpolated there, extended to the odd cosets by transforms of four columns (capi.hip), and recombined with
// x^adj_g and the transition divisor at every point (k_rounds_finish).  Exact arithmetic: the same merged evaluations.
// out = [6 polynomials][4 even cosets][n] (alpha, beta of groups 0..4; this kernel initialises all six).  grid = (n / FNT, 4)
// Polynomials that share their flag and differ only in the power x^adj_g they are multiplied with are MERGED when the sum still has
// degree < 4n:  x^adj_g S_g + x^adj_h S_h = x^adj_g (S_g + x^(adj_h - adj_g) S_h), and adj_h - adj_g is the difference of the declared
// evaluation degrees -- (n - 1) between neighbouring groups 2, 3, 4 and between groups 0, 1.  First family: beta of groups 2, 3, 4
// in one table (degrees <= 4n - 3 - n/1024 after the lift; groups 0 and 1 hold Rescue terms of degree ~4n and stay apart); addition
// family: beta of groups 0, 1 in one table (cubic terms: 4 (n - 1) after the lift).  11 tables instead of 14 to interpolate, extend
// and read back.  x^(n-1) at a point = shift_k^(n-1) w_n^(-j).
__device__ __forceinline__ fp split_lift(const CeParams &p, unsigned k, size_t j) { // x^(n-1) at point j of coset k
    const size_t n = (size_t)1 << p.log_n;
    return fp_mul(p.coset[(size_t)k * CE_COSET_CONSTS + 8], p.w[(n - j) & (n - 1)]);
}
// tables per coefficient set: first family 4 (alpha, beta of groups 0, 1, beta of groups 2..4 merged; Rescue windows + linear groups,
// flags inside) | doubling 3 (alpha, beta of groups 0, 1) | addition 2 (alpha, beta of groups 0, 1 merged) | addition x bit 2 (alpha,
// beta of group 0) | final addition 2 (alpha, beta of groups 0, 1 merged; below)
constexpr int SPLIT_TABLES = 13, SPLIT_FAM0 = 4, SPLIT_FINAL = 11;
// M coefficient sets (the components of an extension proof): the windows' values are computed once, every set has its own tables
// (rtab + c * CE_RTAB_WORDS) and its own block of SPLIT_TABLES output polynomials (out + c * SPLIT_TABLES * 4 n).
template <int M>
#ifndef CS_ROUNDS_SPLIT_WAVES
#define CS_ROUNDS_SPLIT_WAVES 3 // measured: 2 waves 3.86 ms, 3 waves 3.15 ms, 4 waves (spills) 10.3 ms
#endif
__global__ __launch_bounds__(FNT, M == 1 ? CS_ROUNDS_SPLIT_WAVES : 2) void k_rounds_split(CeParams p, fp *__restrict__ out) {
    __shared__ fp ark2_lds[8 * 14];
    __shared__ __attribute__((aligned(16))) fp img_lds[(FNT / 64) * RW_IMG];
    __shared__ fp atab_lds[M * RT_SECTIONS * 8];
    __shared__ fp s2_lds[M == 1 ? 1 : (FNT / 64) * 14 * 64]; // several sets: the window's cubes, one column per lane
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)FNT + threadIdx.x;
    // Even cosets of this table: blockIdx.y-th even coset of the window p.lde holds (cosets [p.k0, ...), p.k0 even; all eight on one
    // GPU).  kk = its index in p.lde, kc = its index among the four even cosets of the domain (table row), ka = 2 kc = its LDE coset.
    const unsigned kk = 2 * blockIdx.y, kc = (p.k0 >> 1) + blockIdx.y, ka = 2 * kc;
    const Frame f = make_frame(p, kk, j);
    if (threadIdx.x < 8 * 14) {
        const unsigned r = threadIdx.x / 14, c = threadIdx.x % 14;
        ark2_lds[threadIdx.x] = p.ptab[((size_t)ka * 48 + P_ARK + 14 + c) * 1024 + ((blockIdx.x * (size_t)FNT + r) & 1023)];
    }
    for (unsigned e = threadIdx.x; e < M * RT_SECTIONS * 8; e += FNT) {
        const unsigned c = e / (RT_SECTIONS * 8), r = e % (RT_SECTIONS * 8);
        atab_lds[e] = p.rtab[(size_t)c * CE_RTAB_WORDS + RT_A + (r >> 3) * 64 + ka * 8 + (r & 7)];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned jr = (unsigned)(j & 7);
    fp *img = img_lds + (threadIdx.x >> 6) * RW_IMG;
    const fp *ark2 = ark2_lds + jr * 14, *atab = atab_lds + jr;
    const fp flags[4] = {f.pv(P_SETUP), f.pv(P_HASH), f.pv(P_SCHNORR_HASH), fp_add(f.pv(P_SETUP), f.pv(P_HASH))};
    const CS_CONST uint32_t *ul = as_const((const uint32_t *)(p.rtab + RT_UL)), *ml = as_const((const uint32_t *)(p.rtab + RT_ML));
    const CS_CONST fp *gt = as_const((const fp *)(p.rtab + RT_G));
    const fp *colbase = p.lde + (size_t)kk * 94 * n;
    const fp *rows = f.cur_p + lane;
    if (lane == 32 && ((size_t)(f.cur_p - lane + 64 - colbase) & (n - 1)) == 0) rows -= n;
    rounds_fetch_window(rows, n, c_windows[0].reg, lane, img);
    fp tot[M][4]; // per set: R_alpha, R_beta of groups 0, 1, 2
#pragma unroll
    for (int c = 0; c < M; c++)
#pragma unroll
        for (int q = 0; q < 4; q++) tot[c][q] = 0;
#pragma unroll 1
    for (int wdx = 0; wdx < 5; wdx++) {
        const RoundWindow w = c_windows[wdx];
        const int g0[2] = {c_window_groups[wdx][0][0], c_window_groups[wdx][1][0]}, g1[2] = {c_window_groups[wdx][0][1], c_window_groups[wdx][1][1]},
                  g2[2] = {c_window_groups[wdx][0][2], c_window_groups[wdx][1][2]};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fp d[14];
#pragma unroll
        for (int jj = 0; jj < 14; jj++) d[jj] = fp_sub(img[jj * RW_ROWS + lane + 1], ark2[jj]);
        // inverse half: cube(INV_MDS d)_i against the coefficient vector of every section (alpha, beta of each group present).
        // One set: accumulated as the cubes are produced.  Several sets: the 14 cubes are parked in a wave-private LDS column and
        // every set runs its own pass over them (8 accumulators live at a time instead of 8 M: no spills).
        Acc128 s[2][4];
        fp *s2w = s2_lds + (threadIdx.x >> 6) * 14 * 64 + lane;
        if (M == 1) {
#pragma unroll
            for (int fs = 0; fs < 2; fs++)
#pragma unroll
                for (int sl = 0; sl < 4; sl++) s[fs][sl] = acc_zero();
        }
#pragma unroll 1
        for (int i = 0; i < 14; i++) {
            const fp s2 = fp_cube(dot14l(ml + i * 56, d));
            if (M > 1) { s2w[i * 64] = s2; continue; }
#pragma unroll
            for (int fs = 0; fs < 2; fs++) {
                if (fs == 1 && w.flag_b < 0) continue;
                const CS_CONST fp *gs = gt + (wdx * 2 + fs) * 4 * 14 + i;
                acc_mad(s[fs][0], gs[0], s2);
                if (g0[fs] >= 0) acc_mad(s[fs][1], gs[14], s2);
                if (g1[fs] >= 0) acc_mad(s[fs][2], gs[28], s2);
                if (g2[fs] >= 0) acc_mad(s[fs][3], gs[42], s2);
                if (i == 6) {
#pragma unroll
                    for (int sl = 0; sl < 4; sl++) acc_fold(s[fs][sl]);
                }
            }
        }
        fp cube[14];
#pragma unroll
        for (int jj = 0; jj < 14; jj++) cube[jj] = fp_cube(img[jj * RW_ROWS + lane]);
        if (wdx < 4) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            rounds_fetch_window(rows, n, c_windows[wdx + 1].reg, lane, img);
        }
#pragma unroll
        for (int c = 0; c < M; c++) {
            if (M > 1) {
#pragma unroll
                for (int fs = 0; fs < 2; fs++)
#pragma unroll
                    for (int sl = 0; sl < 4; sl++) s[fs][sl] = acc_zero();
#pragma unroll 1
                for (int i = 0; i < 14; i++) {
                    const fp s2 = s2w[i * 64];
#pragma unroll
                    for (int fs = 0; fs < 2; fs++) {
                        if (fs == 1 && w.flag_b < 0) continue;
                        const CS_CONST fp *gs = gt + (size_t)c * CE_RTAB_WORDS + (wdx * 2 + fs) * 4 * 14 + i;
                        acc_mad(s[fs][0], gs[0], s2);
                        if (g0[fs] >= 0) acc_mad(s[fs][1], gs[14], s2);
                        if (g1[fs] >= 0) acc_mad(s[fs][2], gs[28], s2);
                        if (g2[fs] >= 0) acc_mad(s[fs][3], gs[42], s2);
                        if (i == 6) {
#pragma unroll
                            for (int sl = 0; sl < 4; sl++) acc_fold(s[fs][sl]);
                        }
                    }
                }
            }
#pragma unroll
            for (int fs = 0; fs < 2; fs++) {
                if (fs == 1 && w.flag_b < 0) continue;
                const int sec = (wdx * 2 + fs) * 4;
                const int fl = fs ? w.flag_b : w.flag_a;
                const fp flag = fl == 0 ? flags[0] : fl == 1 ? flags[1] : fl == 2 ? flags[2] : flags[3];
#pragma unroll
                for (int sl = 0; sl < 4; sl++) {
                    const int g = sl == 0 ? -2 : sl == 1 ? g0[fs] : sl == 2 ? g1[fs] : g2[fs];
                    if (g == -1) continue;
                    acc_fold(s[fs][sl]);
                    const fp fwd = fp_add(dot14l(ul + (size_t)c * CE_RTAB_WORDS * 2 + (sec + sl) * 56, cube), atab[(c * RT_SECTIONS + sec + sl) * 8]);
                    const fp v = fp_mul(flag, fp_sub(acc_reduce(s[fs][sl]), fwd));
                    if (sl == 0) tot[c][0] = fp_add(tot[c][0], v);
                    else {
                        if (g == 0) tot[c][1] = fp_add(tot[c][1], v);
                        if (g == 1) tot[c][2] = fp_add(tot[c][2], v);
                        if (g == 2) tot[c][3] = fp_add(tot[c][3], v);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < M; c++)
#pragma unroll
        for (int q = 0; q < 4; q++) // table 3: group 2 here; the linear groups add groups 3, 4 lifted by x^(n-1), x^(2n-2)
            out[(((size_t)c * SPLIT_TABLES + q) * 4 + kc) * n + j] = tot[c][q];
}

// grid = (n / 256, 8): out[k][j] = [ (R_a + sum_g x^adj_g R_b,g) + doubling(x) (D_a + sum_g x^adj_g D_b,g) + addition(x) (A_a + ...) ]
//                                  * (x - w^(n-1)) / (x^n - 1).
// (+ the fourth family with the "flag" -addition(x) * register 37, see k_ec_split; + the boundary terms).  The eleven polynomials'
// values come from the split evaluations (even cosets, [11][4][n]: the first family of four -- alpha, beta of groups 0, 1, beta of
// groups 2..4 merged -- then doubling (3), addition (2: beta of groups 0, 1 merged), addition x bit (2)) or from their extension
// (odd cosets, [4 cosets][11][n]).  ADDS to p.out.
template <int M>
__global__ __launch_bounds__(256) void k_split_finish(CeParams p, const fp *__restrict__ even, const fp *__restrict__ odd, const fp *__restrict__ hi) {
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)256 + threadIdx.x;
    const unsigned k = blockIdx.y;
    const fp *cc = p.coset + (size_t)k * CE_COSET_CONSTS;
    const fp x = fp_mul(cc[0], p.w[j]);
    fp xp[5];
#pragma unroll
    for (int g = 0; g < 5; g++) xp[g] = fp_mul(cc[2 + g], p.w[(j * p.adj_mod_n[g]) & (n - 1)]);
    const fp *per = p.ptab + (size_t)k * 48 * 1024 + (j & 1023);
    const fp doubling = per[(size_t)P_DOUBLING * 1024], scalar_mult = per[(size_t)P_SCALAR_MULT * 1024];
    const fp addition = fp_mul(c_not(doubling), scalar_mult);
    const fp *col = p.lde + (size_t)k * 94 * n + j;
    const fp bit37 = col[(size_t)37 * n], r58 = col[(size_t)58 * n], r59 = col[(size_t)59 * n];
    const fp divisor = fp_mul(fp_sub(x, p.w_last), cc[1]);
    const fp xb = fp_mul(cc[7], p.w[(j * p.badj_mod_n) & (n - 1)]);
    const fp *bi = p.binv + (size_t)k * 2 * n + j;
    const fp bi0 = bi[0], bi1 = bi[n];
    constexpr int T = M * SPLIT_TABLES; // tables per coset in `odd`
#pragma unroll
    for (int c = 0; c < M; c++) {
        auto value = [&](int tb) {
            const int t = c * SPLIT_TABLES + tb;
            return (k & 1) ? odd[((size_t)(k >> 1) * T + t) * n + j] : even[((size_t)t * 4 + (k >> 1)) * n + j];
        };
        fp total = value(0);
#pragma unroll
        for (int g = 0; g < 3; g++) total = fp_add(total, fp_mul(value(1 + g), xp[g])); // table 3 = groups 2, 3, 4 merged relative to x^adj_2
        const fp dbl = fp_add(value(4), fp_add(fp_mul(value(5), xp[0]), fp_mul(value(6), xp[1])));
        const fp add = fp_add(value(7), fp_mul(value(8), xp[0]));                      // table 8 = groups 0, 1 merged relative to x^adj_0
        const fp addbit = fp_add(value(9), fp_mul(value(10), xp[0]));
        total = fp_add(total, fp_mul(doubling, dbl));
        total = fp_add(total, fp_mul(addition, fp_sub(add, fp_mul(bit37, addbit))));
        {   // final addition: Q = T on the even cosets, T - 2 H on the odd ones (k_final_split); flag = (1 - scalar_mult) schnorr
            fp fa = value(SPLIT_FINAL), fb = value(SPLIT_FINAL + 1);
            if (k & 1) {
                const fp *h = hi + (((size_t)(k >> 1) * M + c) * 2) * n + j;
                fa = fp_sub(fa, fp_dbl(h[0]));
                fb = fp_sub(fb, fp_dbl(h[n]));
            }
            total = fp_add(total, fp_mul(fp_mul(c_not(scalar_mult), per[(size_t)P_SCHNORR * 1024]), fp_add(fa, fp_mul(fb, xp[0]))));
        }
        fp t = fp_mul(total, divisor);
        // boundary constraints on registers 58, 59 at the first and last step (src/air.rs:175-184), as in the last linear group
        const fp *ba = p.coef + c * CE_COEF_WORDS + 230, *bb = ba + 4;
        const fp first = fp_add(fp_mul(fp_sub(r58, p.pubd ? p.pubd[0] : p.pub[0]), fp_add(ba[0], fp_mul(bb[0], xb))), fp_mul(fp_sub(r59, p.pubd ? p.pubd[1] : p.pub[1]), fp_add(ba[1], fp_mul(bb[1], xb))));
        const fp last = fp_add(fp_mul(fp_sub(r58, p.pubd ? p.pubd[7] : p.pub[2]), fp_add(ba[2], fp_mul(bb[2], xb))), fp_mul(fp_sub(r59, p.pubd ? p.pubd[8] : p.pub[3]), fp_add(ba[3], fp_mul(bb[3], xb))));
        t = fp_add(t, fp_add(fp_mul(first, bi0), fp_mul(last, bi1)));
        fp *o = (c == 0 ? p.out : p.out_ext[c == 0 ? 0 : c - 1]) + (size_t)k * n + j;
        *o = t;
    }
}

// The recombination for ONE RANK of a proof sharded by LDE coset (cstark_tx_shard_evaluate): the rank holds cosets [p.k0, p.k0 + nk),
// nk = 2 p.nkc, and evaluated the split polynomials on ITS even cosets only; `odd` / `hi` are the extensions of those PARTIAL tables
// (the other ranks' even cosets taken as zero).  Interpolation, extension and this recombination are linear in the even-coset values,
// so the sum of the ranks' rows for an odd coset is the value the single-GPU k_split_finish writes.  Rows of out[p.nkc + 4][n]:
//   y < p.nkc   own even coset p.k0 + 2 y: complete value (its tables are complete here)
//   y >= p.nkc  odd coset 2 (y - p.nkc) + 1: this rank's share; the boundary terms (registers 58, 59 of the extended trace) only on
//               the rank that holds the coset.  bit37 = register 37 extended to ALL cosets, [8][n]: the one trace column the
//               recombination itself reads (the "flag" of the public-key addition), every rank extends it from its coefficients.
__global__ __launch_bounds__(256) void k_split_finish_shard(CeParams p, const fp *__restrict__ even, const fp *__restrict__ odd, const fp *__restrict__ hi,
                                                            const fp *__restrict__ bit37_all, fp *__restrict__ out) {
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)256 + threadIdx.x;
    const unsigned y = blockIdx.y;
    const bool is_odd = y >= p.nkc;
    const unsigned k = is_odd ? 2 * (y - p.nkc) + 1 : p.k0 + 2 * y;
    const bool mine = k >= p.k0 && k < p.k0 + 2 * p.nkc;
    const fp *cc = p.coset + (size_t)k * CE_COSET_CONSTS;
    const fp x = fp_mul(cc[0], p.w[j]);
    fp xp[3];
#pragma unroll
    for (int g = 0; g < 3; g++) xp[g] = fp_mul(cc[2 + g], p.w[(j * p.adj_mod_n[g]) & (n - 1)]);
    const fp *per = p.ptab + (size_t)k * 48 * 1024 + (j & 1023);
    const fp doubling = per[(size_t)P_DOUBLING * 1024], scalar_mult = per[(size_t)P_SCALAR_MULT * 1024];
    const fp addition = fp_mul(c_not(doubling), scalar_mult);
    const fp bit37 = bit37_all[(size_t)k * n + j];
    const fp divisor = fp_mul(fp_sub(x, p.w_last), cc[1]);
    constexpr int T = SPLIT_TABLES;
    auto value = [&](int t) { return is_odd ? odd[((size_t)(k >> 1) * T + t) * n + j] : even[((size_t)t * 4 + (k >> 1)) * n + j]; };
    fp total = value(0);
#pragma unroll
    for (int g = 0; g < 3; g++) total = fp_add(total, fp_mul(value(1 + g), xp[g]));
    const fp dbl = fp_add(value(4), fp_add(fp_mul(value(5), xp[0]), fp_mul(value(6), xp[1])));
    const fp add = fp_add(value(7), fp_mul(value(8), xp[0]));
    const fp addbit = fp_add(value(9), fp_mul(value(10), xp[0]));
    total = fp_add(total, fp_mul(doubling, dbl));
    total = fp_add(total, fp_mul(addition, fp_sub(add, fp_mul(bit37, addbit))));
    {
        fp fa = value(SPLIT_FINAL), fb = value(SPLIT_FINAL + 1);
        if (is_odd) {
            const fp *h = hi + ((size_t)(k >> 1) * 2) * n + j;
            fa = fp_sub(fa, fp_dbl(h[0]));
            fb = fp_sub(fb, fp_dbl(h[n]));
        }
        total = fp_add(total, fp_mul(fp_mul(c_not(scalar_mult), per[(size_t)P_SCHNORR * 1024]), fp_add(fa, fp_mul(fb, xp[0]))));
    }
    fp t = fp_mul(total, divisor);
    if (mine) { // boundary constraints on registers 58, 59 (src/air.rs:175-184): the rank that holds the coset adds them
        const fp *col = p.lde + (size_t)(k - p.k0) * 94 * n + j;
        const fp r58 = col[(size_t)58 * n], r59 = col[(size_t)59 * n];
        const fp xb = fp_mul(cc[7], p.w[(j * p.badj_mod_n) & (n - 1)]);
        const fp *bi = p.binv + (size_t)k * 2 * n + j;
        const fp *ba = p.coef + 230, *bb = ba + 4;
        const fp first = fp_add(fp_mul(fp_sub(r58, p.pubd ? p.pubd[0] : p.pub[0]), fp_add(ba[0], fp_mul(bb[0], xb))), fp_mul(fp_sub(r59, p.pubd ? p.pubd[1] : p.pub[1]), fp_add(ba[1], fp_mul(bb[1], xb))));
        const fp last = fp_add(fp_mul(fp_sub(r58, p.pubd ? p.pubd[7] : p.pub[2]), fp_add(ba[2], fp_mul(bb[2], xb))), fp_mul(fp_sub(r59, p.pubd ? p.pubd[8] : p.pub[3]), fp_add(ba[3], fp_mul(bb[3], xb))));
        t = fp_add(t, fp_add(fp_mul(first, bi[0]), fp_mul(last, bi[n])));
    }
    out[(size_t)y * n + j] = t;
}
// parts = the ranks' rows side by side, [world][nkc + 4][n] (world = 4 / nkc): merged evaluations of all eight cosets, out[8][n]
__global__ __launch_bounds__(256) void k_shard_combine(const fp *__restrict__ parts, fp *__restrict__ out, unsigned log_n, unsigned nkc) {
    const size_t n = (size_t)1 << log_n;
    const size_t j = blockIdx.x * (size_t)256 + threadIdx.x;
    const unsigned k = blockIdx.y, rows = nkc + 4, world = 4 / nkc;
    fp v;
    if (!(k & 1)) {
        const unsigned kc = k >> 1, r = kc / nkc;
        v = parts[((size_t)r * rows + kc % nkc) * n + j];
    } else {
        v = 0;
        for (unsigned r = 0; r < world; r++) v = fp_add(v, parts[((size_t)r * rows + nkc + (k >> 1)) * n + j]);
    }
    out[(size_t)k * n + j] = v;
}

// doubling gadget for the point at registers [reg, reg + 19)  (ecc.rs:73-98)
template <class A>
__device__ __forceinline__ void fused_doubling(A &acc, const Frame &f, int reg, fp doubling) {
    const Point p = {load6(f, reg, false), load6(f, reg + 6, false), load6(f, reg + 12, false)};
    const Point d = ec_double<CS_EC_CALL>(p);
    acc.begin();
#pragma unroll
    for (int i = 0; i < 6; i++) {
        acc.term(reg + i, fp_sub(f.next(reg + i), d.x.c[i]));
        acc.term(reg + 6 + i, fp_sub(f.next(reg + 6 + i), d.y.c[i]));
        acc.term(reg + 12 + i, fp_sub(f.next(reg + 12 + i), d.z.c[i]));
    }
    acc.term(reg + 18, c_is_binary(f.cur(reg + 18)));
    acc.end(doubling);
}
// conditional mixed addition gadget (ecc.rs:102-138):
//   next - (bit * (p+q) + (1-bit) * p)  =  (next - p) - bit * ((p+q) - p)
template <class A>
__device__ __forceinline__ void fused_addition(A &acc, const Frame &f, int reg, const Fp6 &qx, const Fp6 &qy, fp addition) {
    const Point p = {load6(f, reg, false), load6(f, reg + 6, false), load6(f, reg + 12, false)};
    const fp bit = f.cur(reg + 18);
    const Point a = ec_add_mixed<CS_EC_CALL>(p, qx, qy);
    acc.begin();
#pragma unroll
    for (int i = 0; i < 6; i++) {
        acc.term(reg + i, fp_sub(fp_sub(f.next(reg + i), f.cur(reg + i)), fp_mul(bit, fp_sub(a.x.c[i], f.cur(reg + i)))));
        acc.term(reg + 6 + i, fp_sub(fp_sub(f.next(reg + 6 + i), f.cur(reg + 6 + i)), fp_mul(bit, fp_sub(a.y.c[i], f.cur(reg + 6 + i)))));
        acc.term(reg + 12 + i, fp_sub(fp_sub(f.next(reg + 12 + i), f.cur(reg + 12 + i)), fp_mul(bit, fp_sub(a.z.c[i], f.cur(reg + 12 + i)))));
    }
    acc.term(reg + 18, fp_sub(bit, f.next(reg + 18)));
    acc.end(addition);
}
// final addition S + h*P with X reduced to affine (ecc.rs:146-172)
template <class A>
__device__ __forceinline__ void fused_final_addition(A &acc, const Frame &f, fp final_add) {
    const Point sp = {load6(f, 0, false), load6(f, 6, false), load6(f, 12, false)};
    const Point hp = {load6(f, 19, false), load6(f, 25, false), load6(f, 31, false)};
    const Point r = ec_add<CS_EC_CALL>(sp, hp);
    const Fp6 xz = mul6(load6(f, 0, true), r.z);
    acc.begin();
#pragma unroll
    for (int i = 0; i < 6; i++) {
        acc.term(i, fp_sub(xz.c[i], r.x.c[i]));
        acc.term(6 + i, fp_sub(f.next(6 + i), r.y.c[i]));
        acc.term(12 + i, fp_sub(f.next(12 + i), r.z.c[i]));
    }
    acc.end(final_add);
}

// ---- split evaluation of the curve gadgets whose terms have degree < 4n WITHOUT their flag ----------------------------------
// Doubling: next - double(cur) has degree 4 (n - 1) in x (the complete doubling formulas are quartic in the coordinates); mixed
// addition of the constant generator: (next - cur) - bit (add(cur, G) - cur) has degree 3 (n - 1) (the formulas are quadratic in
// the coordinates when the second point is a constant).  With the periodic flag factored out of the section,
//     flag(x) * [ S_alpha(x) + sum_g x^adj_g S_beta,g(x) ],   S_alpha = sum_i alpha_i term_i,  S_beta,g = sum_{i in g} beta_i term_i,
// the S are polynomials of degree < 4n: evaluated on the even cosets only, extended like the Rescue-window polynomials and
// multiplied by the flag at every point in k_split_finish.  (The addition of the public key and the final addition reach degree
// 5 (n - 1) and stay on all eight cosets.)  Accumulator with the interface of Fused: slots of the curve registers are in groups 0..2.
// The two bit registers (slots 18 and 37, the only curve slots of group 2) are left out here: their terms are quadratic at most and
// join the flags-inside family in lin_c_split, which keeps the curve families at three (alpha, beta of groups 0, 1) / two tables.
template <int M>
struct SplitAcc {
    const CS_CONST fp *coefs; // M blocks of CE_COEF_WORDS: alpha[115] | beta[115] | ...
    Acc128 sa[M], sb[M][2];
    int ca, cb[2];
    __device__ __forceinline__ void begin() {
        ca = cb[0] = cb[1] = 0;
#pragma unroll
        for (int c = 0; c < M; c++) sa[c] = sb[c][0] = sb[c][1] = acc_zero();
    }
    __device__ __forceinline__ void term(int i, fp v) { // i is a compile-time constant after unrolling
        if (i == 18 || i == 37) return;
        const bool fa = ++ca == 7;
#pragma unroll
        for (int c = 0; c < M; c++) {
            acc_mad(sa[c], coefs[c * CE_COEF_WORDS + i], v);
            if (fa) acc_fold(sa[c]);
        }
        if (fa) ca = 0;
        const int g = tx_degree_group(i);
#pragma unroll
        for (int q = 0; q < 2; q++)
            if (g == q) {
                const bool fb = ++cb[q] == 7;
#pragma unroll
                for (int c = 0; c < M; c++) {
                    acc_mad(sb[c][q], coefs[c * CE_COEF_WORDS + 115 + i], v);
                    if (fb) acc_fold(sb[c][q]);
                }
                if (fb) cb[q] = 0;
            }
    }
    __device__ __forceinline__ void end(fp) {}
    __device__ __forceinline__ fp result(int c, int q) { // 0: alpha, 1, 2: beta of groups 0, 1
        Acc128 &a = q == 0 ? sa[c] : sb[c][q == 0 ? 0 : q - 1];
        acc_fold(a);
        return acc_reduce(a);
    }
};
// The addition of the public key (a variable point: add(cur, P) is quartic, bit * add quintic) splits once more:
//     (next - cur) - bit (add(cur, P) - cur)  =  L - bit * Q,   L = next - cur (degree n - 1),  Q = add(cur, P) - cur (degree 4 (n - 1)):
// sum coef_i L_i joins the addition family, sum coef_i Q_i is a family of its own whose "flag" is -addition(x) * bit(x), the bit
// being register 37 of the extended trace itself.
// out = [3 (2 for the quartic half above)][4 even cosets][n] of this flag's family; ACCUMULATE: add to what an earlier part of the same
// family wrote.  grid = (n / FNT, 4)
template <int PART, bool ACCUMULATE, int M>
#ifndef CS_EC_WAVES
#define CS_EC_WAVES 2
#endif
__global__ __launch_bounds__(FNT, CS_EC_WAVES) void k_ec_split(CeParams p, fp *__restrict__ out, fp *__restrict__ out_linear) {
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)FNT + threadIdx.x;
    const unsigned kc = (p.k0 >> 1) + blockIdx.y; // own even coset blockIdx.y of the window = even coset kc of the domain (k_rounds_split)
    const Frame f = make_frame(p, 2 * blockIdx.y, j);
    SplitAcc<M> acc;
    acc.coefs = as_const(p.coef);
    constexpr size_t SET = (size_t)SPLIT_TABLES * 4; // tables (x n words) between the blocks of two coefficient sets
    if (PART == PART_DBL0) fused_doubling(acc, f, 0, (fp)0);
    if (PART == PART_DBL1) fused_doubling(acc, f, 19, (fp)0);
    if (PART == PART_ADD0) fused_addition(acc, f, 0, const6(c_generator), const6(c_generator + 6), (fp)0);
    if (PART == PART_ADD1) {
        acc.begin(); // the linear half first: its four sums leave the registers before the curve arithmetic starts
#pragma unroll
        for (int i = 0; i < 18; i++) acc.term(19 + i, fp_sub(f.next(19 + i), f.cur(19 + i)));
#pragma unroll
        for (int c = 0; c < M; c++)
#pragma unroll
            for (int q = 0; q < 2; q++) { // registers 19..36 are all in group 0
                fp *o = out_linear + (c * SET + (size_t)q * 4 + kc) * n + j;
                *o = fp_add(*o, acc.result(c, q));
            }
        const Point pt = {load6(f, 19, false), load6(f, 25, false), load6(f, 31, false)};
        const Point a = ec_add_mixed<CS_EC_CALL>(pt, load6(f, S_KEY, true), load6(f, S_KEY + 6, true)); // pkey = next[S_KEY..], src/air.rs:575
        acc.begin();
#pragma unroll
        for (int i = 0; i < 6; i++) {
            acc.term(19 + i, fp_sub(a.x.c[i], pt.x.c[i]));
            acc.term(25 + i, fp_sub(a.y.c[i], pt.y.c[i]));
            acc.term(31 + i, fp_sub(a.z.c[i], pt.z.c[i]));
        }
    }
    // the h*P registers 19..36 are all in group 0 (PART_DBL1 adds zero for group 1); the addition family keeps beta of groups 0, 1 in ONE
    // table, group 1 lifted by x^(n-1) (cubic terms: the sum stays below degree 4n)
    constexpr int NQ = (PART == PART_ADD1 || PART == PART_ADD0) ? 2 : 3;
    const fp xd1 = PART == PART_ADD0 ? split_lift(p, 2 * kc, j) : 0;
#pragma unroll
    for (int c = 0; c < M; c++)
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            fp *o = out + (c * SET + (size_t)q * 4 + kc) * n + j;
            fp v = acc.result(c, q);
            if (PART == PART_ADD0 && q == 1) v = fp_add(v, fp_mul(xd1, acc.result(c, 2)));
            *o = ACCUMULATE ? fp_add(*o, v) : v;
        }
}

// ---- final addition on FIVE cosets instead of eight --------------------------------------------------------------------------
// Its two sums Q (alpha; beta of groups 0, 1 merged with the lift) have degree <= 5 (n - 1): as polynomials in y = x / g,
// Q = sum_{t < 5n} a_t y^t.  On the even cosets y^(4n) = 1, so the values there are those of T(y) = sum_{t < 4n} (a_t + [t < n] a_{4n+t}) y^t,
// a polynomial of degree < 4n that the split machinery interpolates and extends like the others; on the odd cosets y^(4n) = -1 and
// Q = T - 2 H with H(y) = sum_{s < n} a_{4n+s} y^s.  H is pinned by ONE odd coset: the sums are evaluated directly on LDE coset 1
// (this kernel with coset = 1), H = (T - Q) / 2 there (k_final_hi), interpolated (n points) and extended to cosets 3, 5, 7.
// Curve arithmetic on 5 n instead of 8 n points for 24 more n-point transforms.
template <int M>
__global__ __launch_bounds__(FNT, CS_EC_WAVES) void k_final_split(CeParams p, fp *__restrict__ out, int coset) {
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)FNT + threadIdx.x;
    const unsigned kc = (p.k0 >> 1) + blockIdx.y, k = coset < 0 ? 2 * kc : (unsigned)coset; // k: LDE coset; p.lde holds it at k - p.k0
    const Frame f = make_frame(p, k - p.k0, j);
    SplitAcc<M> acc;
    acc.coefs = as_const(p.coef);
    fused_final_addition(acc, f, (fp)0);
    const fp xd1 = split_lift(p, k, j);
    constexpr size_t SET = (size_t)SPLIT_TABLES * 4;
#pragma unroll
    for (int c = 0; c < M; c++) {
        const fp a = acc.result(c, 0), b = fp_add(acc.result(c, 1), fp_mul(xd1, acc.result(c, 2)));
        if (coset < 0) {
            out[(c * SET + kc) * n + j] = a;
            out[(c * SET + 4 + kc) * n + j] = b;
        } else {
            out[((size_t)c * 2) * n + j] = a;
            out[((size_t)c * 2 + 1) * n + j] = b;
        }
    }
}
template <int M>
__global__ __launch_bounds__(256) void k_final_hi(CeParams p, const fp *__restrict__ odd, const fp *__restrict__ direct, fp *__restrict__ hi, fp half) {
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)256 + threadIdx.x;
    const unsigned q = blockIdx.y % 2, c = blockIdx.y / 2;
    const fp t = odd[((size_t)c * SPLIT_TABLES + SPLIT_FINAL + q) * n + j]; // LDE coset 1 = the first odd coset of `odd`
    hi[((size_t)c * 2 + q) * n + j] = fp_mul(fp_sub(t, direct[((size_t)c * 2 + q) * n + j]), half);
}

// A: Fused<M>, or any accumulator with its interface for M sets (AirSumView below: the standalone MerkleAir)
template <int M, class A>
__device__ __forceinline__ void fused_merkle_auth_rest_sets(A &acc, const Frame &f, int base, fp tx_hash, fp hash_copy, fp hash_init) {
    const fp bit = f.next(base + 14), not_bit = c_not(bit);
    acc.begin();
    acc.term(base + 14, c_is_binary(bit));
    acc.end(tx_hash);
    // (Current and next row of a column from ONE load -- next = DPP wave shift of the neighbour's current row, lane 63 fetching its
    // own -- was measured slower, 1.84 vs 1.69 ms: the exec-masked edge loads cost more than the saved traffic.  Fetching each
    // step's 4-5 columns once by LDS-DMA into wave-private images, as the Rescue-window part does, was bit-identical and no
    // faster either: 1.73 ms with one step per batch, 2.09 / 2.37 ms with two / three (LDS footprint -> occupancy).  This part is
    // bound by request latency x bytes in flight per CU, not by the byte count.)
    // Two sections accumulated side by side.  One pass over i = 0..6 touches every cell of the two Rescue states exactly once
    // (the current and next row of a column are loaded back to back, so the second load is served from cache; the earlier
    // form walked the columns three times and moved 2.7x the algorithmic bytes -- this part is bandwidth-bound):
    //   copy (no hash, no input):  cur[b + i] - next[b + i]                                     b = base, base + 15
    //   init (hash input row):     not_bit (cur[b + i] - next[b + i]),  bit (cur[b + i] - next[b + 7 + i]),
    //                              bit (next[base + 15 + i] - next[base + i]),  not_bit (next[base + 22 + i] - next[base + 7 + i])
    Acc128 s_copy[M], s_init[M];
#pragma unroll
    for (int c = 0; c < M; c++) s_copy[c] = s_init[c] = acc_zero();
#pragma unroll 1
    for (int i = 0; i < 7; i++) {
        const fp ca = f.cur(base + i), na0 = f.next(base + i), cb = f.cur(base + 15 + i), nb0 = f.next(base + 15 + i);
        const fp na7 = f.next(base + 7 + i), nb7 = f.next(base + 22 + i);
        const fp da = fp_sub(ca, na0), db = fp_sub(cb, nb0);
        const fp u0 = fp_add(fp_mul(not_bit, da), fp_mul(bit, fp_sub(nb0, na0)));               // slot base + i
        const fp u7 = fp_add(fp_mul(bit, fp_sub(ca, na7)), fp_mul(not_bit, fp_sub(nb7, na7)));  // slot base + 7 + i
        const fp u15 = fp_mul(not_bit, db), u22 = fp_mul(bit, fp_sub(cb, nb7));                 // slots base + 15 + i, base + 22 + i
#pragma unroll
        for (int c = 0; c < M; c++) {
            const fp c0 = acc.coef(c, base + i), c15 = acc.coef(c, base + 15 + i);
            acc_mad(s_copy[c], c0, da);
            acc_mad(s_copy[c], c15, db);
            acc_mad(s_init[c], c0, u0);
            acc_mad(s_init[c], acc.coef(c, base + 7 + i), u7);
            acc_mad(s_init[c], c15, u15);
            acc_mad(s_init[c], acc.coef(c, base + 22 + i), u22);
            acc_fold(s_init[c]);
            if (i == 2 || i == 5) acc_fold(s_copy[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < M; c++) {
        acc_fold(s_copy[c]);
        acc.add(c, fp_add(fp_mul(hash_copy, acc_reduce(s_copy[c])), fp_mul(hash_init, acc_reduce(s_init[c]))));
    }
}

template <int M>
__device__ __forceinline__ void fused_merkle_auth_rest(Fused<M> &acc, const Frame &f, int base, fp tx_hash, fp hash_copy, fp hash_init) {
    fused_merkle_auth_rest_sets<M>(acc, f, base, tx_hash, hash_copy, hash_init);
}

// setup + value-copy constraints (src/air.rs:406-529)
template <int M>
__device__ __forceinline__ void fused_linear_a(Fused<M> &acc, const Frame &f) {
    const fp setup = f.pv(P_SETUP), copy_values = f.pv(P_VALUE_COPY);
    // The two sections (flag setup, src/air.rs:406-503; flag copy_values, :506-529) share the key-copy columns: one pass,
    // two accumulators, every cell loaded once (this part is bandwidth-bound).
    Acc128 s_set[M], s_cp[M];
#pragma unroll
    for (int c = 0; c < M; c++) s_set[c] = s_cp[c] = acc_zero();
#pragma unroll CS_LINA_UNROLL
    for (int i = 0; i < 12; i++) {
        const fp si = f.cur(S_INIT + i), su = f.cur(S_UPD + i), ri = f.cur(R_INIT + i), ru = f.cur(R_UPD + i);
        const fp skn = f.next(S_KEY + i), skc = f.cur(S_KEY + i), rkn = f.next(R_KEY + i), rkc = f.cur(R_KEY + i);
        const fp v0 = fp_sub(si, su), v1 = fp_sub(ri, ru), v2 = fp_sub(skn, si), v3 = fp_sub(rkn, ri), w0 = fp_sub(skn, skc), w1 = fp_sub(rkn, rkc);
#pragma unroll
        for (int c = 0; c < M; c++) {
            const fp ks = acc.coef(c, S_KEY_RES + i), kr = acc.coef(c, R_KEY_RES + i);
            acc_mad(s_set[c], acc.coef(c, VALUE_RES + i), v0);
            acc_mad(s_set[c], acc.coef(c, VALUE_RES + 12 + i), v1);
            acc_mad(s_set[c], ks, v2);
            acc_mad(s_set[c], kr, v3);
            acc_fold(s_set[c]);
            acc_mad(s_cp[c], ks, w0);
            acc_mad(s_cp[c], kr, w1);
            if (i % 3 == 2) acc_fold(s_cp[c]);
        }
    }
    const fp s_spent = fp_sub(f.cur(S_INIT + 12), f.cur(S_UPD + 12));
    const fp nd = f.next(DELTA_COPY), ns = f.next(SIGMA_COPY), nn = f.next(NONCE_COPY);
    const fp t0 = fp_sub(f.cur(R_INIT + 13), f.cur(R_UPD + 13)), t1 = fp_sub(s_spent, fp_sub(f.cur(R_UPD + 12), f.cur(R_INIT + 12)));
    const fp t2 = fp_sub(f.cur(S_UPD + 13), fp_add(f.cur(S_INIT + 13), FP_ONE));
    const fp t3 = fp_sub(nd, s_spent), t4 = fp_sub(ns, f.cur(S_UPD + 12)), t5 = fp_sub(nn, f.cur(S_INIT + 13));
    const fp x0 = fp_sub(nd, f.cur(DELTA_COPY)), x1 = fp_sub(ns, f.cur(SIGMA_COPY)), x2 = fp_sub(nn, f.cur(NONCE_COPY));
#pragma unroll
    for (int c = 0; c < M; c++) {
        const fp kd = acc.coef(c, DELTA_COPY_RES), ksg = acc.coef(c, SIGMA_COPY_RES), kn = acc.coef(c, NONCE_COPY_RES);
        acc_mad(s_set[c], acc.coef(c, VALUE_RES + 24), t0);
        acc_mad(s_set[c], acc.coef(c, BALANCE_RES), t1);
        acc_mad(s_set[c], acc.coef(c, NONCE_UPD_RES), t2);
        acc_mad(s_set[c], kd, t3);
        acc_mad(s_set[c], ksg, t4);
        acc_mad(s_set[c], kn, t5);
        acc_fold(s_set[c]);
        acc_mad(s_cp[c], kd, x0);
        acc_mad(s_cp[c], ksg, x1);
        acc_mad(s_cp[c], kn, x2);
        acc_fold(s_cp[c]);
        acc.add(c, fp_add(fp_mul(setup, acc_reduce(s_set[c])), fp_mul(copy_values, acc_reduce(s_cp[c]))));
    }
}
// merkle::update without its rounds (src/merkle/update/air.rs:215-369)
template <int M>
__device__ __forceinline__ void fused_linear_b(Fused<M> &acc, const Frame &f) {
    const fp tx_hash = f.pv(P_MERKLE), hash_input = f.pv(P_HASH_INPUT), finish = f.pv(P_FINISH), hash_flag = f.pv(P_HASH);
    // ---- merkle::update without its rounds (src/merkle/update/air.rs:215-369)
    {
        const fp hash_copy = fp_mul(tx_hash, c_not(fp_add(hash_flag, hash_input)));
        const fp hash_init = fp_mul(tx_hash, hash_input);
#pragma unroll 1
        for (int blk = 0; blk < 2; blk++) fused_merkle_auth_rest(acc, f, blk == 0 ? S_INIT : R_INIT, tx_hash, hash_copy, hash_init);
    }
    {
        Acc128 s_nf[M], s_f[M];
#pragma unroll
        for (int c = 0; c < M; c++) s_nf[c] = s_f[c] = acc_zero();
#pragma unroll
        for (int i = 0; i < 7; i++) {
            const fp nr = f.next(PREV_ROOT + i), cr = f.cur(PREV_ROOT + i);
            const fp v0 = fp_sub(nr, cr), v1 = fp_sub(nr, f.next(R_UPD + i)), v2 = fp_sub(f.cur(S_UPD + i), f.cur(R_INIT + i)),
                     v3 = fp_sub(f.next(S_INIT + i), cr);
#pragma unroll
            for (int c = 0; c < M; c++) {
                const fp c0 = acc.coef(c, PREV_ROOT + i);
                acc_mad(s_nf[c], c0, v0);
                acc_mad(s_f[c], c0, v1);
                acc_mad(s_f[c], acc.coef(c, INT_ROOT_RES + i), v2);
                acc_mad(s_f[c], acc.coef(c, PREV_MATCH_RES + i), v3);
                if (i & 1) acc_fold(s_f[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < M; c++) {
            acc_fold(s_nf[c]);
            acc_fold(s_f[c]);
            acc.add(c, fp_add(fp_mul(c_not(finish), acc_reduce(s_nf[c])), fp_mul(finish, acc_reduce(s_f[c]))));
        }
    }
}
// schnorr linear parts, hash copy, range proofs (src/schnorr/air.rs:451-530, src/air.rs:543-609)
template <int M>
__device__ __forceinline__ void fused_linear_c(Fused<M> &acc, const Frame &f) {
    const fp schnorr_mask = f.pv(P_SCHNORR), scalar_mult = f.pv(P_SCALAR_MULT), doubling = f.pv(P_DOUBLING), schnorr_hash = f.pv(P_SCHNORR_HASH);
    const fp range_flag = f.pv(P_RANGE_STEP), range_finish = f.pv(P_RANGE_FINISH);
    const fp copy_hash = fp_mul(c_not(schnorr_hash), schnorr_mask);
    const fp final_add = fp_mul(c_not(scalar_mult), schnorr_mask);
    const fp addition = fp_mul(c_not(doubling), scalar_mult);
    // ---- schnorr linear parts (src/schnorr/air.rs:451-530)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const fp dflag = f.pv(P_DIGEST + i);
        const fp c = f.cur(41 - i), nx = f.next(41 - i);
        const fp u41 = fp_add(fp_mul(fp_mul(dflag, doubling), fp_sub(nx, fp_add(fp_dbl(c), f.next(37)))), fp_mul(fp_mul(c_not(dflag), doubling), fp_sub(c, nx)));
        const fp u38 = fp_add(fp_mul(addition, fp_sub(f.cur(38 + i), f.next(38 + i))), fp_mul(final_add, fp_sub(f.cur(38 + i), f.cur(42 + i))));
#pragma unroll
        for (int q = 0; q < M; q++) acc.add(q, fp_add(fp_mul(acc.coef(q, 41 - i), u41), fp_mul(acc.coef(q, 38 + i), u38)));
    }
    acc.begin(); // enforce_hash_copy (:309-330) with the internal inputs of src/air.rs:543-565
#pragma unroll
    for (int i = 0; i < 7; i++) {
        acc.term(42 + i, fp_sub(f.cur(42 + i), f.next(42 + i)));
        fp inp = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int m = k * 7 + i;
            const fp cell = m < 12 ? f.next(S_KEY + m) : m < 24 ? f.next(R_KEY + m - 12) : m == 24 ? f.next(DELTA_COPY) : m == 25 ? f.next(NONCE_COPY) : 0;
            inp = fp_add(inp, fp_mul(f.pv(P_HASH_INTERNAL + k), cell));
        }
        acc.term(49 + i, fp_sub(f.next(49 + i), inp));
    }
    acc.end(copy_hash);
    {   // range proofs (src/air.rs:583-609)
        const fp db = f.next(DELTA_BIT), sb = f.next(SIGMA_BIT);
        acc.begin();
        acc.term(DELTA_ACC, fp_sub(f.next(DELTA_ACC), fp_add(fp_dbl(f.cur(DELTA_ACC)), db)));
        acc.term(DELTA_BIT, c_is_binary(db));
        acc.term(SIGMA_ACC, fp_sub(f.next(SIGMA_ACC), fp_add(fp_dbl(f.cur(SIGMA_ACC)), sb)));
        acc.term(SIGMA_BIT, c_is_binary(sb));
        acc.end(range_flag);
        const fp dr = fp_sub(f.next(DELTA_ACC), f.next(DELTA_COPY));
        acc.begin();
        acc.term(DELTA_RANGE_RES, dr);
        acc.term(SIGMA_RANGE_RES, dr);
        acc.end(range_finish);
    }
}

// ---- split evaluation of the linear groups (flags inside: every term has degree < 4n, see the degree table in DESIGN.md) ---------
// Section accumulator for slots of any degree group: alpha-weighted sum and one beta-weighted sum per group.  The slot may be a
// run-time value (rolled loops): its group is then a uniform condition.
constexpr unsigned G0 = 1, G1 = 2, G2 = 4, G3 = 8, G4 = 16;
// GM: the groups this section's slots can be in -- only those get a beta accumulator (four registers each) and a test in term().
template <unsigned GM>
struct SectionAccT {
    const CS_CONST fp *coefs;
    Acc128 sa, sb[5];
    int ca, cb[5];
    __device__ __forceinline__ void begin() {
        sa = acc_zero(); ca = 0;
#pragma unroll
        for (int g = 0; g < 5; g++)
            if ((GM >> g) & 1u) { sb[g] = acc_zero(); cb[g] = 0; }
    }
    __device__ __forceinline__ void term(int i, fp v) {
        acc_mad(sa, coefs[i], v);
        if (++ca == 7) { acc_fold(sa); ca = 0; }
        const int g = tx_degree_group(i);
#pragma unroll
        for (int q = 0; q < 5; q++)
            if (((GM >> q) & 1u) && g == q) {
                acc_mad(sb[q], coefs[115 + i], v);
                if (++cb[q] == 7) { acc_fold(sb[q]); cb[q] = 0; }
            }
    }
    // the sums so far are computed HERE: an empty volatile statement the optimiser cannot move arithmetic across (a scheduling barrier alone
    // only binds the machine scheduler; the passes before it still sink a section's arithmetic towards the flush)
    static __device__ __forceinline__ void pin128(const Acc128 &a) { // inputs only: no new value, so no copies
        asm volatile("" : : "v"((uint32_t)a.lo), "v"((uint32_t)(a.lo >> 32)), "v"((uint32_t)a.hi), "v"((uint32_t)(a.hi >> 32)));
    }
    __device__ __forceinline__ void pin() {
        pin128(sa);
#pragma unroll
        for (int g = 0; g < 5; g++)
            if ((GM >> g) & 1u) pin128(sb[g]);
    }
    // tot[0] += flag * alpha sum, tot[1 + g] += flag * beta sum of group g, for the groups in GMASK (the others received nothing)
    template <unsigned GMASK>
    __device__ __forceinline__ void flush(fp flag, fp (&tot)[6]) {
        static_assert((GMASK & ~GM) == 0, "flushed group without an accumulator");
        acc_fold(sa);
        tot[0] = fp_add(tot[0], fp_mul(flag, acc_reduce(sa)));
#pragma unroll
        for (int g = 0; g < 5; g++)
            if ((GMASK >> g) & 1u) {
                acc_fold(sb[g]);
                tot[1 + g] = fp_add(tot[1 + g], fp_mul(flag, acc_reduce(sb[g])));
            }
    }
};
using SectionAcc = SectionAccT<31u>;

// setup + value-copy constraints: every slot is in group 4
__device__ __forceinline__ void lin_a_split(const CS_CONST fp *coefs, const Frame &f, fp (&tot)[6]) {
    const fp setup = f.pv(P_SETUP), copy_values = f.pv(P_VALUE_COPY);
    SectionAcc s_set, s_cp;
    s_set.coefs = s_cp.coefs = coefs;
    s_set.begin(); s_cp.begin();
#pragma unroll CS_LINS_UNROLL_A
    for (int i = 0; i < 12; i++) {
        const fp si = f.cur(S_INIT + i), su = f.cur(S_UPD + i), ri = f.cur(R_INIT + i), ru = f.cur(R_UPD + i);
        const fp skn = f.next(S_KEY + i), skc = f.cur(S_KEY + i), rkn = f.next(R_KEY + i), rkc = f.cur(R_KEY + i);
        s_set.term(VALUE_RES + i, fp_sub(si, su));
        s_set.term(VALUE_RES + 12 + i, fp_sub(ri, ru));
        s_set.term(S_KEY_RES + i, fp_sub(skn, si));
        s_set.term(R_KEY_RES + i, fp_sub(rkn, ri));
        s_cp.term(S_KEY_RES + i, fp_sub(skn, skc));
        s_cp.term(R_KEY_RES + i, fp_sub(rkn, rkc));
    }
    const fp s_spent = fp_sub(f.cur(S_INIT + 12), f.cur(S_UPD + 12));
    const fp nd = f.next(DELTA_COPY), ns = f.next(SIGMA_COPY), nn = f.next(NONCE_COPY);
    s_set.term(VALUE_RES + 24, fp_sub(f.cur(R_INIT + 13), f.cur(R_UPD + 13)));
    s_set.term(BALANCE_RES, fp_sub(s_spent, fp_sub(f.cur(R_UPD + 12), f.cur(R_INIT + 12))));
    s_set.term(NONCE_UPD_RES, fp_sub(f.cur(S_UPD + 13), fp_add(f.cur(S_INIT + 13), FP_ONE)));
    s_set.term(DELTA_COPY_RES, fp_sub(nd, s_spent));
    s_set.term(SIGMA_COPY_RES, fp_sub(ns, f.cur(S_UPD + 12)));
    s_set.term(NONCE_COPY_RES, fp_sub(nn, f.cur(S_INIT + 13)));
    s_cp.term(DELTA_COPY_RES, fp_sub(nd, f.cur(DELTA_COPY)));
    s_cp.term(SIGMA_COPY_RES, fp_sub(ns, f.cur(SIGMA_COPY)));
    s_cp.term(NONCE_COPY_RES, fp_sub(nn, f.cur(NONCE_COPY)));
    s_set.flush<G4>(setup, tot);
    s_cp.flush<G4>(copy_values, tot);
}
// merkle::update without its rounds
__device__ __forceinline__ void lin_b_split(const CS_CONST fp *coefs, const Frame &f, fp (&tot)[6]) {
    const fp tx_hash = f.pv(P_MERKLE), hash_input = f.pv(P_HASH_INPUT), finish = f.pv(P_FINISH), hash_flag = f.pv(P_HASH);
    const fp hash_copy = fp_mul(tx_hash, c_not(fp_add(hash_flag, hash_input)));
    const fp hash_init = fp_mul(tx_hash, hash_input);
    SectionAcc sa, sb;
    sa.coefs = sb.coefs = coefs;
#pragma unroll CS_LINS_UNROLL_BLK
    for (int blk = 0; blk < 2; blk++) {
        const int base = blk == 0 ? S_INIT : R_INIT;
        const fp bit = f.next(base + 14), not_bit = c_not(bit);
        sa.begin();
        sa.term(base + 14, c_is_binary(bit));
        sa.flush<G1 | G2>(tx_hash, tot); // slot 14: group 1, slot 43: group 2
        sa.begin(); sb.begin();
#pragma unroll CS_LINS_UNROLL_B
        for (int i = 0; i < 7; i++) {
            const fp ca = f.cur(base + i), na0 = f.next(base + i), cb = f.cur(base + 15 + i), nb0 = f.next(base + 15 + i);
            const fp na7 = f.next(base + 7 + i), nb7 = f.next(base + 22 + i);
            const fp da = fp_sub(ca, na0), db = fp_sub(cb, nb0);
            sa.term(base + i, da);
            sa.term(base + 15 + i, db);
            sb.term(base + i, fp_add(fp_mul(not_bit, da), fp_mul(bit, fp_sub(nb0, na0))));
            sb.term(base + 7 + i, fp_add(fp_mul(bit, fp_sub(ca, na7)), fp_mul(not_bit, fp_sub(nb7, na7))));
            sb.term(base + 15 + i, fp_mul(not_bit, db));
            sb.term(base + 22 + i, fp_mul(bit, fp_sub(cb, nb7)));
        }
        sa.flush<G0 | G1 | G2>(hash_copy, tot);
        sb.flush<G0 | G1 | G2>(hash_init, tot);
    }
    sa.begin(); sb.begin();
#pragma unroll CS_LINS_UNROLL_B
    for (int i = 0; i < 7; i++) {
        const fp nr = f.next(PREV_ROOT + i), cr = f.cur(PREV_ROOT + i);
        sa.term(PREV_ROOT + i, fp_sub(nr, cr));
        sb.term(PREV_ROOT + i, fp_sub(nr, f.next(R_UPD + i)));
        sb.term(INT_ROOT_RES + i, fp_sub(f.cur(S_UPD + i), f.cur(R_INIT + i)));
        sb.term(PREV_MATCH_RES + i, fp_sub(f.next(S_INIT + i), cr));
    }
    sa.flush<G4>(c_not(finish), tot);
    sb.flush<G3 | G4>(finish, tot);
}
// schnorr linear parts, hash copy, range proofs (the boundary terms are added in k_split_finish)
__device__ __forceinline__ void lin_c_split(const CS_CONST fp *coefs, const Frame &f, fp (&tot)[6]) {
    const fp schnorr_mask = f.pv(P_SCHNORR), scalar_mult = f.pv(P_SCALAR_MULT), doubling = f.pv(P_DOUBLING), schnorr_hash = f.pv(P_SCHNORR_HASH);
    const fp range_flag = f.pv(P_RANGE_STEP), range_finish = f.pv(P_RANGE_FINISH);
    const fp copy_hash = fp_mul(c_not(schnorr_hash), schnorr_mask);
    const fp final_add = fp_mul(c_not(scalar_mult), schnorr_mask);
    const fp addition = fp_mul(c_not(doubling), scalar_mult);
    SectionAcc s;
    s.coefs = coefs;
    s.begin(); // flags are part of the values here
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const fp dflag = f.pv(P_DIGEST + i);
        const fp c = f.cur(41 - i), nx = f.next(41 - i);
        s.term(41 - i, fp_add(fp_mul(fp_mul(dflag, doubling), fp_sub(nx, fp_add(fp_dbl(c), f.next(37)))), fp_mul(fp_mul(c_not(dflag), doubling), fp_sub(c, nx))));
        s.term(38 + i, fp_add(fp_mul(addition, fp_sub(f.cur(38 + i), f.next(38 + i))), fp_mul(final_add, fp_sub(f.cur(38 + i), f.cur(42 + i)))));
    }
    // the bit registers of the curve gadgets (slots 18, 37: group 2), left out of the curve families: binary under `doubling`
    // (ecc.rs:96), copied under `addition` (ecc.rs:136)
    {
        const fp b18 = f.cur(18), b37 = f.cur(37);
        s.term(18, fp_add(fp_mul(doubling, c_is_binary(b18)), fp_mul(addition, fp_sub(b18, f.next(18)))));
        s.term(37, fp_add(fp_mul(doubling, c_is_binary(b37)), fp_mul(addition, fp_sub(b37, f.next(37)))));
    }
    s.flush<G2>(FP_ONE, tot);
    s.begin();
#pragma unroll
    for (int i = 0; i < 7; i++) {
        s.term(42 + i, fp_sub(f.cur(42 + i), f.next(42 + i)));
        fp inp = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int m = k * 7 + i;
            const fp cell = m < 12 ? f.next(S_KEY + m) : m < 24 ? f.next(R_KEY + m - 12) : m == 24 ? f.next(DELTA_COPY) : m == 25 ? f.next(NONCE_COPY) : 0;
            inp = fp_add(inp, fp_mul(f.pv(P_HASH_INTERNAL + k), cell));
        }
        s.term(49 + i, fp_sub(f.next(49 + i), inp));
    }
    s.flush<G2>(copy_hash, tot);
    const fp db = f.next(DELTA_BIT), sbit = f.next(SIGMA_BIT);
    s.begin();
    s.term(DELTA_ACC, fp_sub(f.next(DELTA_ACC), fp_add(fp_dbl(f.cur(DELTA_ACC)), db)));
    s.term(DELTA_BIT, c_is_binary(db));
    s.term(SIGMA_ACC, fp_sub(f.next(SIGMA_ACC), fp_add(fp_dbl(f.cur(SIGMA_ACC)), sbit)));
    s.term(SIGMA_BIT, c_is_binary(sbit));
    s.flush<G2 | G3 | G4>(range_flag, tot);
    const fp dr = fp_sub(f.next(DELTA_ACC), f.next(DELTA_COPY));
    s.begin();
    s.term(DELTA_RANGE_RES, dr);
    s.term(SIGMA_RANGE_RES, dr);
    s.flush<G4>(range_finish, tot);
}
// adds to the six polynomials of the first family (alpha, beta of groups 0..4): out = [6][4 even cosets][n].  grid = (n / FNT, 4)
// (extension proofs: one launch per coefficient set `set`; these groups are cheap and bandwidth-bound, nothing is worth sharing)
template <int PART>
#ifndef CS_LIN_B_SPLIT_WAVES
#define CS_LIN_B_SPLIT_WAVES 3 // measured 1.05 (4, 77 spilled registers) / 0.90 (3) / 0.99 ms (2)
#endif
#ifndef CS_LIN_C_SPLIT_WAVES
#define CS_LIN_C_SPLIT_WAVES 2 // 3 (32 spilled registers) and 4 measured: no gain
#endif
__global__ __launch_bounds__(FNT, PART == PART_LIN_C ? CS_LIN_C_SPLIT_WAVES : PART == PART_LIN_B ? CS_LIN_B_SPLIT_WAVES : 4) void k_lin_split(CeParams p, fp *__restrict__ out, unsigned set) {
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)FNT + threadIdx.x;
    const unsigned kc = (p.k0 >> 1) + blockIdx.y; // (k_rounds_split)
    const Frame f = make_frame(p, 2 * blockIdx.y, j);
    out += (size_t)set * SPLIT_TABLES * 4 * n;
    const CS_CONST fp *coefs = as_const(p.coef + (size_t)set * CE_COEF_WORDS);
    fp tot[6] = {0, 0, 0, 0, 0, 0};
    if (PART == PART_LIN_A) lin_a_split(coefs, f, tot);
    if (PART == PART_LIN_B) lin_b_split(coefs, f, tot);
    if (PART == PART_LIN_C) lin_c_split(coefs, f, tot);
    const fp xd1 = split_lift(p, 2 * kc, j);
    // beta of groups 2, 3, 4 in one table: S_2 + x^(n-1) S_3 + x^(2n-2) S_4 (LIN_A only has alpha and group 4)
    tot[3] = PART == PART_LIN_A ? fp_mul(fp_mul(xd1, xd1), tot[5]) : fp_add(tot[3], fp_mul(xd1, fp_add(tot[4], fp_mul(xd1, tot[5]))));
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const bool touched = PART == PART_LIN_A ? (q == 0 || q == 3) : true;
        if (touched) {
            fp *o = out + ((size_t)q * 4 + kc) * n + j;
            *o = fp_add(*o, tot[q]);
        }
    }
}

// ---- the three linear groups in ONE pass over the frame (round 3) -------------------------------------------------------------------
// k_lin_split<LIN_A>, <LIN_B>, <LIN_C> each walk the Merkle registers 0..57 (LIN_A their current rows for the setup constraints,
// LIN_B current and next rows), LIN_A and LIN_C both read the next rows of the key / amount / nonce copies, and each launch does a
// read-modify-write of the same four tables: 9.2 GB of counter traffic for 3.1 GB of distinct cells, at 3.4-5.4 TB/s (profiles/r02_v13).
// Here every cell is loaded once where the groups overlap: the Merkle loop of LIN_B also feeds LIN_A's setup / copy sections (their
// slots are all in degree group 4: two accumulators each) and the hash-input sums of LIN_C; the rest of LIN_B and LIN_C follows.  Same
// terms, same coefficients, same flags -- the sums are exact, so the four polynomials are bit-identical (split == direct at 2^23 points,
// proof bytes: tests/test_gpu_full_size.py).
#ifndef CS_LIN_ALL_WAVES
#define CS_LIN_ALL_WAVES 3
#endif
// The frame of k_lin_all: the same cells as Frame, read with buffer loads -- the column offset c * n * 8 is a scalar operand and the lane's
// row offset one 32-bit register for all columns, so a load costs no vector instruction (the 64-bit pointer form spends a v_lshl_add_u64 on
// every load: 1 100 of this kernel's 12 900 vector instructions).  One coset's table must stay below 4 GB (log_n <= 22: launch_lin_all).
struct FrameB {
    __amdgpu_buffer_rsrc_t rs;
    uint32_t col_bytes, cur_off, next_off;
    const fp *per_p;
    __device__ __forceinline__ fp ld(uint32_t off, int c) const {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, (uint32_t)c * col_bytes, 0);
        return ((uint64_t)v.y << 32) | v.x;
    }
    __device__ __forceinline__ fp cur(int c) const { return ld(cur_off, c); }
    __device__ __forceinline__ fp next(int c) const { return ld(next_off, c); }
    __device__ __forceinline__ fp pv(int c) const { return per_p[(size_t)c * 1024]; }
};
__device__ __forceinline__ FrameB make_frame_b(const CeParams &p, unsigned kk, size_t j) {
    const size_t n = (size_t)1 << p.log_n;
    FrameB f;
    f.rs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.lde + (size_t)kk * 94 * n), 0, (uint32_t)(94 * n * 8), 0x00020000); // raw buffer, bounds = the table
    f.col_bytes = (uint32_t)(n * 8);
    f.cur_off = (uint32_t)(j * 8);
    f.next_off = (uint32_t)(((j + 1) & (n - 1)) * 8);
    f.per_p = p.ptab + (size_t)(p.k0 + kk) * 48 * 1024 + (j & 1023);
    return f;
}
template <int... K, class Fn>
__device__ __forceinline__ void static_for(std::integer_sequence<int, K...>, Fn fn) { (fn(std::integral_constant<int, K>{}), ...); }

// One pass over the frame for the three linear groups, as 21 straight-line stages: the seven elements of the sender's leaf pair, the seven of the
// receiver's, the amount copies, the root carry (two stages), the Schnorr limbs, the hash copy (two stages), the range proofs.
//   * every slot -- and so its degree group and its coefficient's address -- is a compile-time value: a term is two multiply-accumulates with a
//     scalar operand.  (Round 3 kept the element loops rolled: the group was then a run-time value, and each term cost a scalar load waited for
//     at once and a chain of scalar branches: 13 689 vector + 5 436 scalar instructions per wave, 425 branches.)
//   * stage k + 1's cells are requested before stage k's arithmetic starts.  Every wave of a CU runs this same program nearly in step, so without
//     the overlap inside a wave the CU alternates between all waves waiting on memory and all waves computing: the old kernel's 2.14 ms were its
//     1.0 ms of memory time PLUS its 1.1 ms of issue time (profiles/r04_lin_all.txt).
//   * scheduling barriers between the stages, and the sums pinned at the end of each (SectionAccT::pin), keep the order written here: left alone
//     the scheduler lifts the kernel's ~300 loads above the arithmetic and spills, and the passes before it sink a stage's arithmetic to the flush.
#define CS_LIN_SECTION() __builtin_amdgcn_sched_barrier(0)
template <class F>
__device__ __forceinline__ void lin_all_split(const CS_CONST fp *coefs, const F &f, fp (&tot)[6]) {
    constexpr int NSTAGE = 21, NL = 32;
    SectionAccT<G4> s_set, s_cp, ra;
    SectionAccT<G0 | G1 | G2> sa, sb;
    SectionAccT<G3 | G4> rb;
    SectionAccT<G2> s;
    SectionAccT<G2 | G3 | G4> sr;
    s_set.coefs = s_cp.coefs = ra.coefs = sa.coefs = sb.coefs = rb.coefs = s.coefs = sr.coefs = coefs;
    s_set.begin(); s_cp.begin();
    fp bit = 0, not_bit = 0, s_spent = 0, su12 = 0, si13 = 0, schnorr_mask = 0;

    // what stage k reads, in the order its arithmetic names them
    auto load = [&](auto kc, fp (&L)[NL]) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < 14) {
            constexpr int blk = k / 7, i = k % 7, base = blk == 0 ? S_INIT : R_INIT, key = blk == 0 ? S_KEY : R_KEY;
            L[0] = f.cur(base + i); L[1] = f.next(base + i); L[2] = f.cur(base + 15 + i); L[3] = f.next(base + 15 + i);
            L[4] = f.cur(base + 7 + i); L[5] = f.next(base + 7 + i); L[6] = f.cur(base + 22 + i); L[7] = f.next(base + 22 + i);
            L[8] = f.next(key + i); L[9] = f.cur(key + i);
            if constexpr (i < 5) { L[10] = f.next(key + 7 + i); L[11] = f.cur(key + 7 + i); }
            if constexpr (i == 0) { L[12] = f.next(base + 14); L[13] = f.pv(P_MERKLE); }
            if constexpr (i == 6) { L[12] = f.pv(P_HASH_INPUT); L[13] = f.pv(P_HASH); L[14] = f.pv(P_MERKLE); }
        } else if constexpr (k == 14) {
            L[0] = f.next(DELTA_COPY); L[1] = f.next(SIGMA_COPY); L[2] = f.next(NONCE_COPY);
            L[3] = f.cur(DELTA_COPY); L[4] = f.cur(SIGMA_COPY); L[5] = f.cur(NONCE_COPY);
            L[6] = f.pv(P_SETUP); L[7] = f.pv(P_VALUE_COPY);
        } else if constexpr (k == 15 || k == 16) {
            constexpr int i0 = k == 15 ? 0 : 4, cnt = k == 15 ? 4 : 3;
#pragma unroll
            for (int q = 0; q < cnt; q++) {
                const int i = i0 + q;
                L[6 * q] = f.next(PREV_ROOT + i); L[6 * q + 1] = f.cur(PREV_ROOT + i); L[6 * q + 2] = f.next(R_UPD + i);
                L[6 * q + 3] = f.cur(S_UPD + i); L[6 * q + 4] = f.cur(R_INIT + i); L[6 * q + 5] = f.next(S_INIT + i);
            }
            if constexpr (k == 16) L[18] = f.pv(P_FINISH);
        } else if constexpr (k == 17) {
#pragma unroll
            for (int i = 0; i < 4; i++) { L[i] = f.cur(38 + i); L[4 + i] = f.next(38 + i); L[8 + i] = f.cur(42 + i); L[19 + i] = f.pv(P_DIGEST + i); }
            L[12] = f.next(37); L[13] = f.cur(37); L[14] = f.cur(18); L[15] = f.next(18);
            L[16] = f.pv(P_SCHNORR); L[17] = f.pv(P_SCALAR_MULT); L[18] = f.pv(P_DOUBLING);
        } else if constexpr (k == 18 || k == 19) {
            // hash copy.  Its hash-internal inputs: inp[i] = sum_k' pv(P_HASH_INTERNAL + k') cell(7 k' + i) over the 26 cells next(S_KEY ..+12),
            // next(R_KEY ..+12), next(DELTA_COPY), next(NONCE_COPY)  (src/schnorr/air.rs:499-505)
            constexpr int i0 = k == 18 ? 0 : 4, cnt = k == 18 ? 4 : 3;
#pragma unroll
            for (int q = 0; q < cnt; q++) {
                const int i = i0 + q;
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const int m = 7 * kk + i;
                    if (m < 26) L[7 * q + kk] = f.next(m < 12 ? S_KEY + m : m < 24 ? R_KEY + m - 12 : m == 24 ? DELTA_COPY : NONCE_COPY);
                }
                L[7 * q + 4] = f.cur(42 + i); L[7 * q + 5] = f.next(42 + i); L[7 * q + 6] = f.next(49 + i);
            }
#pragma unroll
            for (int kk = 0; kk < 4; kk++) L[28 + kk] = f.pv(P_HASH_INTERNAL + kk);
            if constexpr (k == 19) L[21] = f.pv(P_SCHNORR_HASH);
        } else {
            L[0] = f.next(DELTA_BIT); L[1] = f.next(SIGMA_BIT); L[2] = f.next(DELTA_ACC); L[3] = f.cur(DELTA_ACC);
            L[4] = f.next(SIGMA_ACC); L[5] = f.cur(SIGMA_ACC); L[6] = f.next(DELTA_COPY);
            L[7] = f.pv(P_RANGE_STEP); L[8] = f.pv(P_RANGE_FINISH);
        }
    };

    auto compute = [&](auto kc, const fp (&L)[NL]) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < 14) {
            constexpr int blk = k / 7, i = k % 7, base = blk == 0 ? S_INIT : R_INIT, key_res = blk == 0 ? S_KEY_RES : R_KEY_RES;
            if constexpr (i == 0) {
                bit = L[12]; not_bit = c_not(bit);
                sa.begin();
                sa.term(base + 14, c_is_binary(bit));
                sa.template flush<G1 | G2>(L[13], tot); // slot 14: group 1, slot 43: group 2
                sa.begin(); sb.begin();
            }
            const fp ca = L[0], na0 = L[1], cb = L[2], nb0 = L[3], ca7 = L[4], na7 = L[5], cb7 = L[6], nb7 = L[7], kn = L[8], kc_ = L[9];
            const fp da = fp_sub(ca, na0), db = fp_sub(cb, nb0);
            // merkle::update, hash copy / hash input sections (LIN_B)
            sa.term(base + i, da);
            sa.term(base + 15 + i, db);
            sb.term(base + i, fp_add(fp_mul(not_bit, da), fp_mul(bit, fp_sub(nb0, na0))));
            sb.term(base + 7 + i, fp_add(fp_mul(bit, fp_sub(ca, na7)), fp_mul(not_bit, fp_sub(nb7, na7))));
            sb.term(base + 15 + i, fp_mul(not_bit, db));
            sb.term(base + 22 + i, fp_mul(bit, fp_sub(cb, nb7)));
            // setup / value-copy sections (LIN_A): elements i and 7 + i of the leaf pair (init, updated) and of the key copy
            s_set.term(VALUE_RES + 12 * blk + i, fp_sub(ca, cb));
            s_set.term(key_res + i, fp_sub(kn, ca));
            s_cp.term(key_res + i, fp_sub(kn, kc_));
            if constexpr (i < 5) {
                const fp kn7 = L[10], kc7 = L[11];
                s_set.term(VALUE_RES + 12 * blk + 7 + i, fp_sub(ca7, cb7));
                s_set.term(key_res + 7 + i, fp_sub(kn7, ca7));
                s_cp.term(key_res + 7 + i, fp_sub(kn7, kc7));
            } else if constexpr (i == 5) { // element 12: balances
                if constexpr (blk == 0) { s_spent = fp_sub(ca7, cb7); su12 = cb7; }
                else s_set.term(BALANCE_RES, fp_sub(s_spent, fp_sub(cb7, ca7)));
            } else {                       // element 13: nonces
                if constexpr (blk == 0) { si13 = ca7; s_set.term(NONCE_UPD_RES, fp_sub(cb7, fp_add(ca7, FP_ONE))); }
                else s_set.term(VALUE_RES + 24, fp_sub(ca7, cb7));
            }
            if constexpr (i == 6) {
                const fp hash_input = L[12], hash_flag = L[13], tx_hash = L[14];
                sa.template flush<G0 | G1 | G2>(fp_mul(tx_hash, c_not(fp_add(hash_flag, hash_input))), tot); // hash copy
                sb.template flush<G0 | G1 | G2>(fp_mul(tx_hash, hash_input), tot);                            // hash input
            } else {
                sa.pin(); sb.pin();
            }
            s_set.pin(); s_cp.pin();
        } else if constexpr (k == 14) { // amount / balance / nonce copies (LIN_A)
            const fp nd = L[0], ns = L[1], nn = L[2];
            s_set.term(DELTA_COPY_RES, fp_sub(nd, s_spent));
            s_set.term(SIGMA_COPY_RES, fp_sub(ns, su12));
            s_set.term(NONCE_COPY_RES, fp_sub(nn, si13));
            s_cp.term(DELTA_COPY_RES, fp_sub(nd, L[3]));
            s_cp.term(SIGMA_COPY_RES, fp_sub(ns, L[4]));
            s_cp.term(NONCE_COPY_RES, fp_sub(nn, L[5]));
            s_set.template flush<G4>(L[6], tot);
            s_cp.template flush<G4>(L[7], tot);
        } else if constexpr (k == 15 || k == 16) { // root carry / finish of merkle::update (LIN_B)
            constexpr int i0 = k == 15 ? 0 : 4, cnt = k == 15 ? 4 : 3;
            if constexpr (k == 15) { ra.begin(); rb.begin(); }
#pragma unroll
            for (int q = 0; q < cnt; q++) {
                const int i = i0 + q;
                const fp nr = L[6 * q], cr = L[6 * q + 1];
                ra.term(PREV_ROOT + i, fp_sub(nr, cr));
                rb.term(PREV_ROOT + i, fp_sub(nr, L[6 * q + 2]));
                rb.term(INT_ROOT_RES + i, fp_sub(L[6 * q + 3], L[6 * q + 4]));
                rb.term(PREV_MATCH_RES + i, fp_sub(L[6 * q + 5], cr));
            }
            if constexpr (k == 16) {
                const fp finish = L[18];
                ra.template flush<G4>(c_not(finish), tot);
                rb.template flush<G3 | G4>(finish, tot);
            } else {
                ra.pin(); rb.pin();
            }
        } else if constexpr (k == 17) { // schnorr: limbs of h, scalar bits (LIN_C).  The flags are part of the values here
            schnorr_mask = L[16];
            const fp scalar_mult = L[17], doubling = L[18];
            const fp final_add = fp_mul(c_not(scalar_mult), schnorr_mask);
            const fp addition = fp_mul(c_not(doubling), scalar_mult);
            const fp n37 = L[12];
            s.begin();
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const fp dflag = L[19 + i];
                const fp c = L[3 - i], nx = L[7 - i]; // cur / next of register 41 - i
                s.term(41 - i, fp_add(fp_mul(fp_mul(dflag, doubling), fp_sub(nx, fp_add(fp_dbl(c), n37))), fp_mul(fp_mul(c_not(dflag), doubling), fp_sub(c, nx))));
                s.term(38 + i, fp_add(fp_mul(addition, fp_sub(L[i], L[4 + i])), fp_mul(final_add, fp_sub(L[i], L[8 + i]))));
            }
            const fp b37 = L[13], b18 = L[14];
            s.term(18, fp_add(fp_mul(doubling, c_is_binary(b18)), fp_mul(addition, fp_sub(b18, L[15]))));
            s.term(37, fp_add(fp_mul(doubling, c_is_binary(b37)), fp_mul(addition, fp_sub(b37, n37))));
            s.template flush<G2>(FP_ONE, tot);
        } else if constexpr (k == 18 || k == 19) { // hash copy (LIN_C)
            constexpr int i0 = k == 18 ? 0 : 4, cnt = k == 18 ? 4 : 3;
            if constexpr (k == 18) s.begin();
#pragma unroll
            for (int q = 0; q < cnt; q++) {
                const int i = i0 + q;
                fp inp = 0;
#pragma unroll
                for (int kk = 0; kk < 4; kk++)
                    if (7 * kk + i < 26) inp = fp_add(inp, fp_mul(L[28 + kk], L[7 * q + kk]));
                s.term(42 + i, fp_sub(L[7 * q + 4], L[7 * q + 5]));
                s.term(49 + i, fp_sub(L[7 * q + 6], inp));
            }
            if constexpr (k == 19) s.template flush<G2>(fp_mul(c_not(L[21]), schnorr_mask), tot);
            else s.pin();
        } else { // range proofs (LIN_C)
            const fp dbit = L[0], sbit = L[1];
            sr.begin();
            sr.term(DELTA_ACC, fp_sub(L[2], fp_add(fp_dbl(L[3]), dbit)));
            sr.term(DELTA_BIT, c_is_binary(dbit));
            sr.term(SIGMA_ACC, fp_sub(L[4], fp_add(fp_dbl(L[5]), sbit)));
            sr.term(SIGMA_BIT, c_is_binary(sbit));
            sr.template flush<G2 | G3 | G4>(L[7], tot);
            const fp dr = fp_sub(L[2], L[6]);
            sr.begin();
            sr.term(DELTA_RANGE_RES, dr);
            sr.term(SIGMA_RANGE_RES, dr);
            sr.template flush<G4>(L[8], tot);
        }
    };

    fp LA[NL], LB[NL];
    load(std::integral_constant<int, 0>{}, LA);
    static_for(std::make_integer_sequence<int, NSTAGE>{}, [&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k + 1 < NSTAGE) load(std::integral_constant<int, k + 1>{}, (k & 1) ? LA : LB);
        CS_LIN_SECTION();
        compute(kc, (k & 1) ? LB : LA);
        CS_LIN_SECTION();
    });
}
#undef CS_LIN_SECTION
// adds to the four polynomials of the first family: out = [4][4 even cosets][n].  grid = (n / FNT, 4); one launch per coefficient set.
// BUF: the frame is read with buffer loads (FrameB; a coset's table below 4 GB), else through 64-bit pointers.
template <bool BUF>
__global__ __launch_bounds__(FNT, CS_LIN_ALL_WAVES) void k_lin_all(CeParams p, fp *__restrict__ out, unsigned set) {
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)FNT + threadIdx.x;
    const unsigned kc = (p.k0 >> 1) + blockIdx.y; // (k_rounds_split)
    out += (size_t)set * SPLIT_TABLES * 4 * n;
    const CS_CONST fp *coefs = as_const(p.coef + (size_t)set * CE_COEF_WORDS);
    fp tot[6] = {0, 0, 0, 0, 0, 0};
    if constexpr (BUF) lin_all_split(coefs, make_frame_b(p, 2 * blockIdx.y, j), tot);
    else lin_all_split(coefs, make_frame(p, 2 * blockIdx.y, j), tot);
    const fp xd1 = split_lift(p, 2 * kc, j);
    // beta of groups 2, 3, 4 in one table: S_2 + x^(n-1) S_3 + x^(2n-2) S_4
    tot[3] = fp_add(tot[3], fp_mul(xd1, fp_add(tot[4], fp_mul(xd1, tot[5]))));
#pragma unroll
    for (int q = 0; q < 4; q++) {
        fp *o = out + ((size_t)q * 4 + kc) * n + j;
        *o = fp_add(*o, tot[q]);
    }
}

// grid = (n / FNT, nk)
#ifndef CS_ROUNDS_WAVES
#define CS_ROUNDS_WAVES 3
#endif
#ifndef CS_ROUNDS_UNROLL
#define CS_ROUNDS_UNROLL 1
#endif
#ifndef CS_EC_WAVES
#define CS_EC_WAVES 2
#endif
#ifndef CS_LIN_WAVES
#define CS_LIN_WAVES 4
#endif
constexpr int part_waves(int part, int m) {
    return part == PART_ROUNDS ? (m == 1 ? CS_ROUNDS_WAVES : 2) : (part >= PART_DBL0 && part <= PART_FINAL) ? CS_EC_WAVES : part == PART_LIN_C ? 2 : (m == 1 ? CS_LIN_WAVES : 2);
}
template <int PART, int M>
__global__ __launch_bounds__(FNT, part_waves(PART, M)) void k_eval_fused(CeParams p) {
    __shared__ fp xp_lds[5 * FNT];
    extern __shared__ __attribute__((aligned(16))) uint8_t rounds_lds[]; // PART_ROUNDS only: ROUNDS_LDS bytes
#ifdef CS_ROUNDS_MFMA
    if (PART == PART_ROUNDS) {
        const uint4 *src = (const uint4 *)(p.rtab + RT_MT);
        for (unsigned i = threadIdx.x; i < MT_BYTES / 16; i += FNT) ((uint4 *)rounds_lds)[i] = src[i];
        __syncthreads();
    }
#endif
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)FNT + threadIdx.x;
    const unsigned kk = blockIdx.y;
    const Frame f = make_frame(p, kk, j);
    // x = shift_k * w_n^j and its powers (per-coset constants precomputed on the host)
    const fp *cc = p.coset + (size_t)(p.k0 + kk) * CE_COSET_CONSTS;
    const fp x = fp_mul(cc[0], p.w[j]);
#pragma unroll
    for (int g = 0; g < 5; g++) xp_lds[g * FNT + threadIdx.x] = fp_mul(cc[2 + g], p.w[(j * p.adj_mod_n[g]) & (n - 1)]);
    Fused<M> acc;
    acc.coefs = as_const(p.coef);
    acc.xp = xp_lds;
    acc.cnt = 0;
#pragma unroll
    for (int c = 0; c < M; c++) acc.total[c] = 0;

    if (PART == PART_ROUNDS) {
        __shared__ fp ark2_lds[8 * 14];
        __shared__ __attribute__((aligned(16))) fp img_lds[(FNT / 64) * RW_IMG];
        __shared__ fp atab_lds[M * RT_SECTIONS * 8];
        for (unsigned e = threadIdx.x; e < M * RT_SECTIONS * 8; e += FNT) { // (set, section, r) <- A[set][section][k][r]
            const unsigned r = e & 7, sec = (e >> 3) % RT_SECTIONS, c = e / (8 * RT_SECTIONS);
            atab_lds[e] = p.rtab[(size_t)c * CE_RTAB_WORDS + RT_A + sec * 64 + (p.k0 + kk) * 8 + r];
        }
        if (threadIdx.x < 8 * 14) { // row r = j mod 8 of the workgroup's first rows (FNT is a multiple of 8), constant c
            const unsigned r = threadIdx.x / 14, c = threadIdx.x % 14;
            ark2_lds[threadIdx.x] = p.ptab[((size_t)(p.k0 + kk) * 48 + P_ARK + 14 + c) * 1024 + ((blockIdx.x * (size_t)FNT + r) & 1023)];
        }
        __syncthreads();
        fused_rounds(acc, f, p.rtab, p.k0 + kk, (unsigned)(j & 7), rounds_lds, rounds_lds + MT_BYTES + (size_t)(threadIdx.x >> 6) * 64 * mdsmfma::ROW_BYTES,
                     ark2_lds, atab_lds, img_lds + (threadIdx.x >> 6) * RW_IMG, p.lde + (size_t)kk * 94 * n);
    }
    if (PART >= PART_DBL0 && PART <= PART_FINAL) {
        const fp scalar_mult = f.pv(P_SCALAR_MULT), doubling = f.pv(P_DOUBLING);
        if (PART == PART_DBL0) fused_doubling(acc, f, 0, doubling);
        if (PART == PART_DBL1) fused_doubling(acc, f, 19, doubling);
        if (PART == PART_ADD0) fused_addition(acc, f, 0, const6(c_generator), const6(c_generator + 6), fp_mul(c_not(doubling), scalar_mult));
        if (PART == PART_ADD1) // pkey = next[S_KEY..], src/air.rs:575
            fused_addition(acc, f, 19, load6(f, S_KEY, true), load6(f, S_KEY + 6, true), fp_mul(c_not(doubling), scalar_mult));
        if (PART == PART_FINAL) fused_final_addition(acc, f, fp_mul(c_not(scalar_mult), f.pv(P_SCHNORR)));
    }
    if (PART == PART_LIN_A) fused_linear_a(acc, f);
    if (PART == PART_LIN_B) fused_linear_b(acc, f);
    if (PART == PART_LIN_C) fused_linear_c(acc, f);

    // transition divisor (x^n - 1) / (x - w^(n-1)); x^n is constant on a coset
    const fp divisor = fp_mul(fp_sub(x, p.w_last), cc[1]);
#pragma unroll
    for (int c = 0; c < M; c++) {
        fp t = fp_mul(acc.total[c], divisor);
        fp *o = (c == 0 ? p.out : p.out_ext[c == 0 ? 0 : c - 1]) + (size_t)kk * n + j;
        if (PART == PART_LIN_C) {
            // boundary constraints on registers 58, 59 at the first and last step (src/air.rs:175-184)
            const fp xb = fp_mul(cc[7], p.w[(j * p.badj_mod_n) & (n - 1)]);
            const fp *ba = p.coef + c * CE_COEF_WORDS + 230, *bb = ba + 4;
            const fp r58 = f.cur(58), r59 = f.cur(59);
            const fp first = fp_add(fp_mul(fp_sub(r58, p.pubd ? p.pubd[0] : p.pub[0]), fp_add(ba[0], fp_mul(bb[0], xb))), fp_mul(fp_sub(r59, p.pubd ? p.pubd[1] : p.pub[1]), fp_add(ba[1], fp_mul(bb[1], xb))));
            const fp last = fp_add(fp_mul(fp_sub(r58, p.pubd ? p.pubd[7] : p.pub[2]), fp_add(ba[2], fp_mul(bb[2], xb))), fp_mul(fp_sub(r59, p.pubd ? p.pubd[8] : p.pub[3]), fp_add(ba[3], fp_mul(bb[3], xb))));
            // 1/(x - 1) and 1/(x - w^(n-1)) depend on the domain only: cached tables
            const fp *bi = p.binv + (size_t)(p.k0 + kk) * 2 * n + j;
            t = fp_add(t, fp_mul(first, bi[0]));
            t = fp_add(t, fp_mul(last, bi[n]));
        }
        *o = PART == PART_ROUNDS ? t : fp_add(*o, t); // ROUNDS is launched first, the others accumulate in stream order
    }
}

// the transition constraints of MerkleAir at one point, through any accumulator with add(slot, flag, value)
// ROUNDS = false: without the four Rescue round gadgets (k_merkle_rounds evaluates them in their folded form)
template <bool ROUNDS = true, class Acc>
__device__ __forceinline__ void merkle_transitions(Acc &acc, const Frame &f) {
    // periodic layout: setup, hash(tx), hash_input, finish, hash_mask, ark[28]; the gadget templates read the round
    // constants at P_ARK + i relative to per_p, so give them a view shifted by (5 - P_ARK) columns
    Frame fr = f;
    fr.per_p = f.per_p - (size_t)(P_ARK - 5) * 512;
    const fp setup = f.pv(0), tx_hash = f.pv(1), hash_input = f.pv(2), finish = f.pv(3), hash_flag = f.pv(4);
#pragma unroll 1
    for (int i = 0; i < 12; i++) {
        acc.add(VALUE_RES + i, setup, fp_sub(f.cur(S_INIT + i), f.cur(S_UPD + i)));
        acc.add(VALUE_RES + 12 + i, setup, fp_sub(f.cur(R_INIT + i), f.cur(R_UPD + i)));
    }
    acc.add(VALUE_RES + 24, setup, fp_sub(f.cur(R_INIT + 13), f.cur(R_UPD + 13)));
    acc.add(BALANCE_RES, setup, fp_sub(fp_sub(f.cur(S_INIT + 12), f.cur(S_UPD + 12)), fp_sub(f.cur(R_UPD + 12), f.cur(R_INIT + 12))));
    acc.add(NONCE_UPD_RES, setup, fp_sub(f.cur(S_UPD + 13), fp_add(f.cur(S_INIT + 13), FP_ONE)));
    if (ROUNDS) {
        enforce_round(acc, fr, S_INIT, S_INIT, hash_flag, 0, 0, false);
        enforce_round(acc, fr, S_UPD, S_UPD, hash_flag, 0, 0, false);
        enforce_round(acc, fr, R_INIT, R_INIT, hash_flag, 0, 0, false);
        enforce_round(acc, fr, R_UPD, R_UPD, hash_flag, 0, 0, false);
    }
    merkle_auth_rest(acc, f, S_INIT, tx_hash, hash_input, hash_flag);
    merkle_auth_rest(acc, f, R_INIT, tx_hash, hash_input, hash_flag);
    const fp not_finish = c_not(finish);
#pragma unroll 1
    for (int i = 0; i < 7; i++) {
        const fp nr = f.next(PREV_ROOT + i), cr = f.cur(PREV_ROOT + i);
        acc.add(PREV_ROOT + i, not_finish, fp_sub(nr, cr));
        acc.add(PREV_ROOT + i, finish, fp_sub(nr, f.next(R_UPD + i)));
        acc.add(INT_ROOT_RES + i, finish, fp_sub(f.cur(S_UPD + i), f.cur(R_INIT + i)));
        acc.add(PREV_MATCH_RES + i, finish, fp_sub(f.next(S_INIT + i), cr));
    }
}
// =====================================================================================================
// Standalone sub-AIRs: parity-oriented kernels, every constraint materialised.
// MerkleAir::evaluate_transition  /root/reference/src/merkle/update/air.rs:64-141, :215-289
__global__ __launch_bounds__(NT) void k_eval_transitions_merkle(const fp *lde, const fp *ptab, fp *out, unsigned log_n, unsigned k0) {
    const size_t n = (size_t)1 << log_n;
    const size_t j = blockIdx.x * (size_t)NT + threadIdx.x;
    const unsigned kk = blockIdx.y;
    const fp *base = lde + (size_t)kk * 65 * n;
    Frame f;
    f.n = n;
    f.cur_p = base + j;
    f.next_p = base + ((j + 1) & (n - 1));
    f.per_p = ptab + (size_t)(k0 + kk) * 33 * 512 + (j & 511);
    f.pcycle = 512;
    AccAll acc{out + (size_t)kk * 106 * n + j, n};
    merkle_transitions(acc, f);
}
// SchnorrAir::evaluate_transition  src/schnorr/air.rs:68-109 -> evaluate_constraints :394-531.
// lde: [nk][56][n]; aux: [nk][19][n] (LDE of pkey x12 and message-chunk x7 columns); ptab: [b][36][512] (8 masks, 28 ark)
// PART: 0 = double-and-add step of s*G (slots 0..18), 1 = of h*P (19..37), 2 = limb accumulators, Rescue round, hash copies, 3 = final
// addition.  One kernel for everything needs 450 VGPRs (one wave per SIMD); the parts accumulate into the zero-filled output one after
// the other on the stream.
template <int PART>
__global__ __launch_bounds__(NT) void k_eval_transitions_schnorr(const fp *lde, const fp *aux, const fp *ptab, fp *out, unsigned log_n, unsigned k0) {
    const size_t n = (size_t)1 << log_n;
    const size_t j = blockIdx.x * (size_t)NT + threadIdx.x;
    const unsigned kk = blockIdx.y;
    const fp *base = lde + (size_t)kk * 56 * n;
    Frame f;
    f.n = n;
    f.cur_p = base + j;
    f.next_p = base + ((j + 1) & (n - 1));
    f.per_p = ptab + (size_t)(k0 + kk) * 36 * 512 + (j & 511);
    f.pcycle = 512;
    Frame fr = f; // view whose column P_ARK is the first round-constant column (index 8 here)
    fr.per_p = f.per_p - (size_t)(P_ARK - 8) * 512;
    const fp *ax = aux + (size_t)kk * 19 * n + j;
    AccAll acc{out + (size_t)kk * 56 * n + j, n};
    const fp global_mask = f.pv(0), scalar_mult = f.pv(1), doubling = f.pv(2), hash_flag = f.pv(7);
    const fp copy_hash = fp_mul(c_not(hash_flag), global_mask);
    const fp final_add = fp_mul(c_not(scalar_mult), global_mask);
    const fp addition = fp_mul(c_not(doubling), scalar_mult);
    if constexpr (PART == 0) {
        const Fp6 gx = const6(c_generator), gy = const6(c_generator + 6);
        enforce_scalar_mult_step(acc, f, 0, gx, gy, doubling, addition);
    }
    if constexpr (PART == 1) {
        const Fp6 px = fp6_load_strided(ax, n), py = fp6_load_strided(ax + 6 * n, n); // periodic pkey columns
        enforce_scalar_mult_step(acc, f, 19, px, py, doubling, addition);
    }
    if constexpr (PART == 2) {
#pragma unroll 1
        for (int i = 0; i < 4; i++) {
            const fp dflag = f.pv(3 + i);
            const fp c = f.cur(41 - i), nx = f.next(41 - i);
            acc.add(41 - i, fp_mul(dflag, doubling), fp_sub(nx, fp_add(fp_dbl(c), f.next(37))));
            acc.add(41 - i, fp_mul(c_not(dflag), doubling), fp_sub(c, nx));
            acc.add(38 + i, addition, fp_sub(f.cur(38 + i), f.next(38 + i)));
            acc.add(38 + i, final_add, fp_sub(f.cur(38 + i), f.cur(42 + i)));
        }
        enforce_round(acc, fr, 42, 42, hash_flag, 0, 0, false);
#pragma unroll 1
        for (int i = 0; i < 7; i++) {
            acc.add(42 + i, copy_hash, fp_sub(f.cur(42 + i), f.next(42 + i)));
            acc.add(49 + i, copy_hash, fp_sub(f.next(49 + i), ax[(size_t)(12 + i) * n]));
        }
    }
    if constexpr (PART == 3) {
        const Point s = {load6(f, 0, false), load6(f, 6, false), load6(f, 12, false)};
        const Point hp = {load6(f, 19, false), load6(f, 25, false), load6(f, 31, false)};
        const Point r = ec_add<true>(s, hp);
        const Fp6 xz = mul6_call(load6(f, 0, true), r.z);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            acc.add(i, final_add, fp_sub(xz.c[i], r.x.c[i]));
            acc.add(6 + i, final_add, fp_sub(f.next(6 + i), r.y.c[i]));
            acc.add(12 + i, final_add, fp_sub(f.next(12 + i), r.z.c[i]));
        }
    }
}
// ---- SchnorrAir, fused: the merged transition sum  sum_i (alpha_i + beta_i x^adj_i) C_i(x)  without the 56 materialised values -------
// The same gadget templates as the TransactionAir evaluator (lazy F_p6 arithmetic, one 128-bit accumulation per section) behind an
// accumulator that looks the coefficient and the degree group of a slot up in the tables of the generic merge (AirCombineParams).
// Six launches (doubling / addition of s*G, of h*P, final addition, the rest) that accumulate into one value per point; k_air_combine
// then takes that value instead of walking the materialised evaluations (p.tsum).  Exact arithmetic: the merged evaluations -- and
// the proof bytes -- equal those of the materialising path (tests/test_gpu_small_airs.py).
struct AirSum {
    const CS_CONST fp *alpha, *beta;
    const CS_CONST uint32_t *grp;
    const fp *xp; // LDS [AIR_MAX_GROUPS][FNT]: x^adj_g of this lane's point
    Acc128 s;
    int cnt;
    fp total;
    __device__ __forceinline__ fp coef(int i) const { return fp_add(alpha[i], fp_mul(beta[i], xp[grp[i] * FNT + threadIdx.x])); }
    __device__ __forceinline__ void begin() { s = acc_zero(); cnt = 0; }
    __device__ __forceinline__ void term(int i, fp v) {
        acc_mad(s, coef(i), v);
        if (++cnt == 7) { acc_fold(s); cnt = 0; }
    }
    __device__ __forceinline__ void end(fp flag) {
        acc_fold(s);
        total = fp_add(total, fp_mul(flag, acc_reduce(s)));
    }
    __device__ __forceinline__ void add(int i, fp flag, fp val) { total = fp_add(total, fp_mul(coef(i), fp_mul(flag, val))); } // AccAll's interface
};
enum { SF_DBL0 = 0, SF_ADD0, SF_DBL1, SF_ADD1, SF_FINAL, SF_REST, SF_PARTS };
// AFTER_SPLIT: the four doubling / addition parts came from the degree-split form (k_schnorr_ec_split + k_schnorr_split_finish wrote the
// output first): every part here adds to it, and SF_REST also carries the terms of the two bit registers (slots 18, 37) that the split
// curve families leave out
// FOLDED_ROUND (SF_REST): the Rescue round of the message hash is left to k_merkle_rounds<4, 1, 56, ...> (its folded form)
template <int PART, bool AFTER_SPLIT = false, bool FOLDED_ROUND = false>
__global__ __launch_bounds__(FNT, PART == SF_REST ? 2 : CS_EC_WAVES) void k_schnorr_fused(AirCombineParams p, const fp *__restrict__ aux,
                                                                                       const fp *__restrict__ ptab) {
    __shared__ fp xp_lds[AIR_MAX_GROUPS * FNT];
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)FNT + threadIdx.x;
    const unsigned kk = blockIdx.y, k = p.k0 + kk;
    const fp *base = p.lde + (size_t)kk * 56 * n;
    Frame f;
    f.n = n;
    f.cur_p = base + j;
    f.next_p = base + ((j + 1) & (n - 1));
    f.per_p = ptab + (size_t)k * 36 * 512 + (j & 511);
    f.pcycle = 512;
    for (unsigned g = 0; g < p.n_tgrp; g++) xp_lds[g * FNT + threadIdx.x] = fp_mul(p.tgrp_shift[k][g], p.w[(j * p.tgrp_adj[g]) & (n - 1)]);
    AirSum acc{as_const(p.t_alpha), as_const(p.t_beta), as_const(p.t_grp), xp_lds, acc_zero(), 0, 0};
    const fp *ax = aux + (size_t)kk * 19 * n + j;
    const fp global_mask = f.pv(0), scalar_mult = f.pv(1), doubling = f.pv(2), hash_flag = f.pv(7);
    const fp addition = fp_mul(c_not(doubling), scalar_mult);
    if constexpr (PART == SF_DBL0) fused_doubling(acc, f, 0, doubling);
    if constexpr (PART == SF_DBL1) fused_doubling(acc, f, 19, doubling);
    if constexpr (PART == SF_ADD0) fused_addition(acc, f, 0, const6(c_generator), const6(c_generator + 6), addition);
    if constexpr (PART == SF_ADD1) fused_addition(acc, f, 19, fp6_load_strided(ax, n), fp6_load_strided(ax + 6 * n, n), addition); // periodic pkey columns
    if constexpr (PART == SF_FINAL) fused_final_addition(acc, f, fp_mul(c_not(scalar_mult), global_mask));
    if constexpr (PART == SF_REST) {
        const fp copy_hash = fp_mul(c_not(hash_flag), global_mask), final_add = fp_mul(c_not(scalar_mult), global_mask);
        Frame fr = f; // view whose column P_ARK is the first round-constant column (index 8 here)
        fr.per_p = f.per_p - (size_t)(P_ARK - 8) * 512;
#pragma unroll 1
        for (int i = 0; i < 4; i++) {
            const fp dflag = f.pv(3 + i);
            const fp c = f.cur(41 - i), nx = f.next(41 - i);
            acc.add(41 - i, fp_mul(dflag, doubling), fp_sub(nx, fp_add(fp_dbl(c), f.next(37))));
            acc.add(41 - i, fp_mul(c_not(dflag), doubling), fp_sub(c, nx));
            acc.add(38 + i, addition, fp_sub(f.cur(38 + i), f.next(38 + i)));
            acc.add(38 + i, final_add, fp_sub(f.cur(38 + i), f.cur(42 + i)));
        }
        if (!FOLDED_ROUND) enforce_round(acc, fr, 42, 42, hash_flag, 0, 0, false);
#pragma unroll 1
        for (int i = 0; i < 7; i++) {
            acc.add(42 + i, copy_hash, fp_sub(f.cur(42 + i), f.next(42 + i)));
            acc.add(49 + i, copy_hash, fp_sub(f.next(49 + i), ax[(size_t)(12 + i) * n]));
        }
        if (AFTER_SPLIT) { // the bit registers of the two ladders: binary under `doubling` (ecc.rs:96), copied under `addition` (ecc.rs:136)
            const fp b18 = f.cur(18), b37 = f.cur(37);
            acc.add(18, doubling, c_is_binary(b18));
            acc.add(18, addition, fp_sub(b18, f.next(18)));
            acc.add(37, doubling, c_is_binary(b37));
            acc.add(37, addition, fp_sub(b37, f.next(37)));
        }
    }
    fp *o = p.out + (size_t)kk * n + j;
    *o = (PART == SF_DBL0 && !AFTER_SPLIT) ? acc.total : fp_add(*o, acc.total); // the launches follow each other on the stream
}

// ---- SchnorrAir: the doubling / addition gadgets in the degree-split form of the TransactionAir evaluator ------------------------------
// Same algebra as k_ec_split (the sums S of coefficient x term WITHOUT their periodic flag have degree < 4n: doubling quartic, addition of
// the constant generator cubic, addition of the public key L - bit Q with Q quartic), on SchnorrAir's 56-register frame: the public key
// is the periodic / public-input column pair of `aux` at the CURRENT row (src/schnorr/air.rs:228-290), the coefficients and degree
// groups those of SchnorrAir (slots 0..5 and 19..36: first group, 6..17: second -- the same partition of the curve slots as in
// TransactionAir, so SplitAcc applies with the coefficients laid out alpha[i] | beta[115 + i]).
// Eight polynomials per proof on the even cosets, out = [8][4][n]: doubling alpha, beta_0, beta_1 | addition alpha, beta_0, beta_1 |
// addition x bit alpha, beta_0.  PART: 1 = doubling of s*G (writes the doubling family), 3 = doubling of h*P (adds to it), 2 = addition of
// G (writes the addition family), 4 = addition of P (adds its linear half to the addition family, writes the third family).
constexpr int SCH_SPLIT_EC_TABLES = 8, SCH_SPLIT_TABLES = 11; // the doubling / addition families | + the final addition's three sums
template <int PART>
__global__ __launch_bounds__(FNT, CS_EC_WAVES) void k_schnorr_ec_split(const fp *__restrict__ lde, const fp *__restrict__ aux, const fp *__restrict__ coefs_tx_layout,
                                                                      fp *__restrict__ out, unsigned log_n) {
    const size_t n = (size_t)1 << log_n;
    const size_t j = blockIdx.x * (size_t)FNT + threadIdx.x;
    const unsigned kc = blockIdx.y, kk = 2 * kc;
    const fp *base = lde + (size_t)kk * 56 * n;
    Frame f;
    f.n = n;
    f.cur_p = base + j;
    f.next_p = base + ((j + 1) & (n - 1));
    f.per_p = nullptr; // no periodic value is read: the flags are applied by k_schnorr_split_finish
    f.pcycle = 512;
    SplitAcc<1> acc;
    acc.coefs = as_const(coefs_tx_layout);
    fp *fam = out + (size_t)(PART == PART_DBL0 || PART == PART_DBL1 ? 0 : 3) * 4 * n;
    if (PART == PART_DBL0) fused_doubling(acc, f, 0, (fp)0);
    if (PART == PART_DBL1) fused_doubling(acc, f, 19, (fp)0);
    if (PART == PART_ADD0) fused_addition(acc, f, 0, const6(c_generator), const6(c_generator + 6), (fp)0);
    if (PART == PART_ADD1) {
        acc.begin(); // the linear half: next - cur of registers 19..36 (all in the first group)
#pragma unroll
        for (int i = 0; i < 18; i++) acc.term(19 + i, fp_sub(f.next(19 + i), f.cur(19 + i)));
#pragma unroll
        for (int q = 0; q < 2; q++) {
            fp *o = fam + ((size_t)q * 4 + kc) * n + j;
            *o = fp_add(*o, acc.result(0, q));
        }
        const fp *ax = aux + (size_t)kk * 19 * n + j;
        const Point pt = {load6(f, 19, false), load6(f, 25, false), load6(f, 31, false)};
        const Point a = ec_add_mixed<CS_EC_CALL>(pt, fp6_load_strided(ax, n), fp6_load_strided(ax + 6 * n, n));
        acc.begin();
#pragma unroll
        for (int i = 0; i < 6; i++) {
            acc.term(19 + i, fp_sub(a.x.c[i], pt.x.c[i]));
            acc.term(25 + i, fp_sub(a.y.c[i], pt.y.c[i]));
            acc.term(31 + i, fp_sub(a.z.c[i], pt.z.c[i]));
        }
        fam = out + (size_t)6 * 4 * n; // third family: alpha, beta_0
    }
    constexpr int NQ = PART == PART_ADD1 ? 2 : 3;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        fp *o = fam + ((size_t)q * 4 + kc) * n + j;
        const fp v = acc.result(0, q);
        *o = PART == PART_DBL1 ? fp_add(*o, v) : v;
    }
}
// The final addition on FIVE cosets (the scheme of k_final_split / k_final_hi above, with the beta sums of the two degree groups kept
// apart): its three sums F (alpha, beta_0, beta_1; degree <= 5 (n - 1) without the flag) are evaluated on the even cosets, where they
// equal T = F mod (y^4n - 1), tables 8..10 of `even`, and directly on LDE coset 1, which pins H = (T - F) / 2 there; H has degree < n
// and is interpolated from that one coset and extended to cosets 3, 5, 7 by the caller.  coset < 0: the even cosets (grid.y = 4, out =
// tables [3][4][n]); coset = 1: out = [3][n].
__global__ __launch_bounds__(FNT, CS_EC_WAVES) void k_schnorr_final_split(const fp *__restrict__ lde, const fp *__restrict__ coefs_tx_layout, fp *__restrict__ out,
                                                                         unsigned log_n, int coset) {
    const size_t n = (size_t)1 << log_n;
    const size_t j = blockIdx.x * (size_t)FNT + threadIdx.x;
    const unsigned kc = blockIdx.y, kk = coset < 0 ? 2 * kc : (unsigned)coset;
    const fp *base = lde + (size_t)kk * 56 * n;
    Frame f;
    f.n = n;
    f.cur_p = base + j;
    f.next_p = base + ((j + 1) & (n - 1));
    f.per_p = nullptr; // no periodic value is read: the flag is applied by k_schnorr_split_finish
    f.pcycle = 512;
    SplitAcc<1> acc;
    acc.coefs = as_const(coefs_tx_layout);
    fused_final_addition(acc, f, (fp)0);
#pragma unroll
    for (int q = 0; q < 3; q++) {
        if (coset < 0) out[((size_t)q * 4 + kc) * n + j] = acc.result(0, q);
        else out[(size_t)q * n + j] = acc.result(0, q);
    }
}
// hi[q][j] = (T_q - F_q) / 2 on LDE coset 1: odd = [4 odd cosets][T][n] (coset 1 first), direct = [3][n]
__global__ __launch_bounds__(256) void k_schnorr_final_hi(const fp *__restrict__ odd, const fp *__restrict__ direct, fp *__restrict__ hi, fp half, unsigned log_n) {
    const size_t n = (size_t)1 << log_n;
    const size_t j = blockIdx.x * (size_t)256 + threadIdx.x;
    const unsigned q = blockIdx.y;
    hi[(size_t)q * n + j] = fp_mul(fp_sub(odd[((size_t)SCH_SPLIT_EC_TABLES + q) * n + j], direct[(size_t)q * n + j]), half);
}
// recombination over all cosets: out[k][j] = doubling(x) (D_a + x^adj_0 D_b0 + x^adj_1 D_b1) + addition(x) [(A_a + x^adj_0 A_b0 + x^adj_1 A_b1)
// - bit37 (Q_a + x^adj_0 Q_b0)] + final(x) (F_a + x^adj_0 F_b0 + x^adj_1 F_b1), F = T on the even cosets and T - 2 H on the odd ones
// (hi = [4 odd cosets][3][n]; null: the final addition is left to k_schnorr_fused<SF_FINAL>); the remaining parts (hash / limb
// constraints, bit registers) ADD to it afterwards.
// even = [T][4][n] (even cosets), odd = [4 odd cosets][T][n] (their extension), T = 8 or 11; g0, g1 = degree groups of slots 0 and 6.
template <int T>
__global__ __launch_bounds__(256) void k_schnorr_split_finish(AirCombineParams p, const fp *__restrict__ even, const fp *__restrict__ odd,
                                                              const fp *__restrict__ ptab, unsigned g0, unsigned g1, const fp *__restrict__ hi) {
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)256 + threadIdx.x;
    const unsigned k = blockIdx.y; // all eight cosets present, k0 = 0
    const fp x0 = fp_mul(p.tgrp_shift[k][g0], p.w[(j * p.tgrp_adj[g0]) & (n - 1)]), x1 = fp_mul(p.tgrp_shift[k][g1], p.w[(j * p.tgrp_adj[g1]) & (n - 1)]);
    const fp *per = ptab + (size_t)k * 36 * 512 + (j & 511);
    const fp scalar_mult = per[(size_t)1 * 512], doubling = per[(size_t)2 * 512];
    const fp addition = fp_mul(c_not(doubling), scalar_mult);
    const fp bit37 = p.lde[((size_t)k * 56 + 37) * n + j];
    auto value = [&](int t) { return (k & 1) ? odd[((size_t)(k >> 1) * T + t) * n + j] : even[((size_t)t * 4 + (k >> 1)) * n + j]; };
    const fp dbl = fp_add(value(0), fp_add(fp_mul(value(1), x0), fp_mul(value(2), x1)));
    const fp add = fp_add(value(3), fp_add(fp_mul(value(4), x0), fp_mul(value(5), x1)));
    const fp addbit = fp_add(value(6), fp_mul(value(7), x0));
    fp total = fp_add(fp_mul(doubling, dbl), fp_mul(addition, fp_sub(add, fp_mul(bit37, addbit))));
    if (T > SCH_SPLIT_EC_TABLES) {
        fp fa = value(8), fb0 = value(9), fb1 = value(10);
        if (k & 1) {
            const fp *h = hi + (size_t)(k >> 1) * 3 * n + j;
            fa = fp_sub(fa, fp_dbl(h[0]));
            fb0 = fp_sub(fb0, fp_dbl(h[n]));
            fb1 = fp_sub(fb1, fp_dbl(h[2 * n]));
        }
        const fp final_add = fp_mul(c_not(scalar_mult), per[0]); // (1 - scalar_mult) * global mask
        total = fp_add(total, fp_mul(final_add, fp_add(fa, fp_add(fp_mul(fb0, x0), fp_mul(fb1, x1)))));
    }
    p.out[(size_t)k * n + j] = total;
}

// MerkleAir without its round gadgets, for the table-driven accumulator: the same terms as merkle_transitions<false>, grouped by their
// flag into sections (one multiply-accumulate per term and one reduction per section instead of three field products per term; the two
// authentication-path blocks through the TransactionAir evaluator's fused_merkle_auth_rest).  Exact arithmetic: the same sum.
struct AirSumView { // the interface of Fused<1> over an AirSum
    AirSum &a;
    __device__ __forceinline__ fp coef(int, int i) const { return a.coef(i); }
    __device__ __forceinline__ void begin() { a.begin(); }
    __device__ __forceinline__ void term(int i, fp v) { a.term(i, v); }
    __device__ __forceinline__ void end(fp flag) { a.end(flag); }
    __device__ __forceinline__ void add(int, fp v) { a.total = fp_add(a.total, v); }
};
__device__ __forceinline__ void merkle_linear_sections(AirSum &acc, const Frame &f) {
    const fp setup = f.pv(0), tx_hash = f.pv(1), hash_input = f.pv(2), finish = f.pv(3), hash_flag = f.pv(4);
    acc.begin();
#pragma unroll 1
    for (int i = 0; i < 12; i++) {
        acc.term(VALUE_RES + i, fp_sub(f.cur(S_INIT + i), f.cur(S_UPD + i)));
        acc.term(VALUE_RES + 12 + i, fp_sub(f.cur(R_INIT + i), f.cur(R_UPD + i)));
    }
    acc.term(VALUE_RES + 24, fp_sub(f.cur(R_INIT + 13), f.cur(R_UPD + 13)));
    acc.term(BALANCE_RES, fp_sub(fp_sub(f.cur(S_INIT + 12), f.cur(S_UPD + 12)), fp_sub(f.cur(R_UPD + 12), f.cur(R_INIT + 12))));
    acc.term(NONCE_UPD_RES, fp_sub(f.cur(S_UPD + 13), fp_add(f.cur(S_INIT + 13), FP_ONE)));
    acc.end(setup);
    {
        const fp hash_copy = fp_mul(tx_hash, c_not(fp_add(hash_flag, hash_input))), hash_init = fp_mul(tx_hash, hash_input);
        AirSumView v{acc};
        fused_merkle_auth_rest_sets<1>(v, f, S_INIT, tx_hash, hash_copy, hash_init);
        fused_merkle_auth_rest_sets<1>(v, f, R_INIT, tx_hash, hash_copy, hash_init);
    }
    acc.begin();
#pragma unroll 1
    for (int i = 0; i < 7; i++) acc.term(PREV_ROOT + i, fp_sub(f.next(PREV_ROOT + i), f.cur(PREV_ROOT + i)));
    acc.end(c_not(finish));
    acc.begin();
#pragma unroll 1
    for (int i = 0; i < 7; i++) {
        acc.term(PREV_ROOT + i, fp_sub(f.next(PREV_ROOT + i), f.next(R_UPD + i)));
        acc.term(INT_ROOT_RES + i, fp_sub(f.cur(S_UPD + i), f.cur(R_INIT + i)));
        acc.term(PREV_MATCH_RES + i, fp_sub(f.next(S_INIT + i), f.cur(PREV_ROOT + i)));
    }
    acc.end(finish);
}
// MerkleAir, fused: the same body as k_eval_transitions_merkle behind the table-driven accumulator; one value per point of the
// cosets of the constraint-evaluation domain (the others are left alone: k_air_combine writes their zeros).
// AFTER_ROUNDS: k_merkle_rounds wrote the four round gadgets' sum first; this kernel adds every other constraint to it.
template <bool AFTER_ROUNDS>
__global__ __launch_bounds__(FNT, 2) void k_merkle_fused(AirCombineParams p, const fp *__restrict__ ptab) {
    __shared__ fp xp_lds[AIR_MAX_GROUPS * FNT];
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)FNT + threadIdx.x;
    const unsigned kk = blockIdx.y, k = p.k0 + kk;
    if (k % p.stride) return; // uniform over the workgroup
    const fp *base = p.lde + (size_t)kk * 65 * n;
    Frame f;
    f.n = n;
    f.cur_p = base + j;
    f.next_p = base + ((j + 1) & (n - 1));
    f.per_p = ptab + (size_t)k * 33 * 512 + (j & 511);
    f.pcycle = 512;
    for (unsigned g = 0; g < p.n_tgrp; g++) xp_lds[g * FNT + threadIdx.x] = fp_mul(p.tgrp_shift[k][g], p.w[(j * p.tgrp_adj[g]) & (n - 1)]);
    AirSum acc{as_const(p.t_alpha), as_const(p.t_beta), as_const(p.t_grp), xp_lds, acc_zero(), 0, 0};
    if (AFTER_ROUNDS) merkle_linear_sections(acc, f);
    else merkle_transitions<true>(acc, f);
    fp *o = p.out + (size_t)kk * n + j;
    *o = AFTER_ROUNDS ? fp_add(*o, acc.total) : acc.total;
}

// MerkleAir's four Rescue round gadgets (registers S_INIT, S_UPD, R_INIT, R_UPD under the hash mask; result slots = the registers;
// src/merkle/update/air.rs:215-289) in the folded form of the TransactionAir evaluator (k_rounds_setup / k_rounds_split above): the
// forward half is linear in the cubes, so sum_i c_i (MDS cube + ark1)_i = (MDS^T c) . cube + c . ark1 -- two 14-term dot products per
// window (alpha, beta: the 56 round slots share ONE declared degree, i.e. one power x^adj) instead of a 14 x 14 product; the window's
// cells come by LDS-DMA, every LDE cell fetched once.  Exact arithmetic: the merged value is unchanged.
// rtab (u64 words; sections = (window, {alpha, beta})): A[8][8 cosets][8] | limbs of U[8][14] | limbs of INV_MDS[196] | G[8][14]
// (MR_* offsets: rounds_layout.h)
// W0: first Rescue window of the AIR in c_windows (MerkleAir 0..3, SchnorrAir's message hash 4); PCOLS / ARKCOL: columns of the AIR's
// periodic table and the first of its 28 round-constant columns
template <int W0, int PCOLS, int ARKCOL>
__global__ void k_merkle_rounds_setup(const fp *__restrict__ t_alpha, const fp *__restrict__ t_beta, const fp *__restrict__ ptab, fp *__restrict__ rtab,
                                      unsigned n_cosets) {
    const int sec = blockIdx.x, reg = c_windows[W0 + (sec >> 1)].reg, t = threadIdx.x;
    __shared__ fp gam[14];
    if (t < 14) gam[t] = ((sec & 1) ? t_beta : t_alpha)[reg + t];
    __syncthreads();
    if (t < 14) {
        rtab[MR_G + sec * 14 + t] = gam[t];
        fp u = 0;
        for (int i = 0; i < 14; i++) u = fp_add(u, fp_mul(gam[i], c_mds[i * 14 + t]));
        split_limbs(u, (uint32_t *)(rtab + MR_UL) + (sec * 14 + t) * 4);
    }
    if (sec == 0)
        for (int e = t; e < 196; e += blockDim.x) split_limbs(c_inv_mds[e], (uint32_t *)(rtab + MR_ML) + e * 4);
    if (t < 64) { // sum_i c_i ark1_i on coset k at rows = r mod 8 (the round constants' extension has period 8 in the row index)
        const unsigned k = t >> 3, r = t & 7;
        fp a = 0;
        if (k < n_cosets)
            for (int i = 0; i < 14; i++) a = fp_add(a, fp_mul(gam[i], ptab[((size_t)k * PCOLS + ARKCOL + i) * 512 + r]));
        rtab[MR_A + sec * 64 + t] = a;
    }
}
// grid = (n / FNT, nk); xg = the degree group of the round slots (p.t_grp[S_INIT]).  The same kernel serves SchnorrAir's message hash
// (one window, registers 42..55 of a 56-register frame, hash mask in periodic column 7): NWIN windows from c_windows[W0], WIDTH
// registers per coset, FLAGCOL the mask; ADD: the sum is added to the output instead of written.
template <int W0, int NWIN, int WIDTH, int PCOLS, int FLAGCOL, int ARKCOL, bool ADD>
__global__ __launch_bounds__(FNT, 3) void k_merkle_rounds(AirCombineParams p, const fp *__restrict__ ptab, const fp *__restrict__ rtab, unsigned xg) {
    __shared__ fp ark2_lds[8 * 14];
    __shared__ __attribute__((aligned(16))) fp img_lds[(FNT / 64) * RW_IMG];
    __shared__ fp atab_lds[MR_SECTIONS * 8];
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)FNT + threadIdx.x;
    const unsigned kk = blockIdx.y, k = p.k0 + kk;
    if (k % p.stride) return; // uniform over the workgroup
    const fp *base = p.lde + (size_t)kk * WIDTH * n;
    if (threadIdx.x < 8 * 14) {
        const unsigned r = threadIdx.x / 14, c = threadIdx.x % 14;
        ark2_lds[threadIdx.x] = ptab[((size_t)k * PCOLS + ARKCOL + 14 + c) * 512 + ((blockIdx.x * (size_t)FNT + r) & 511)];
    }
    if (threadIdx.x < MR_SECTIONS * 8) atab_lds[threadIdx.x] = rtab[MR_A + (threadIdx.x >> 3) * 64 + k * 8 + (threadIdx.x & 7)];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned jr = (unsigned)(j & 7);
    fp *img = img_lds + (threadIdx.x >> 6) * RW_IMG;
    const fp *ark2 = ark2_lds + jr * 14, *atab = atab_lds + jr;
    const fp flag = ptab[((size_t)k * PCOLS + FLAGCOL) * 512 + (j & 511)];
    const fp xp = fp_mul(p.tgrp_shift[k][xg], p.w[(j * p.tgrp_adj[xg]) & (n - 1)]);
    const CS_CONST uint32_t *ul = as_const((const uint32_t *)(rtab + MR_UL)), *ml = as_const((const uint32_t *)(rtab + MR_ML));
    const CS_CONST fp *gt = as_const(rtab + MR_G);
    const fp *rows = base + j + lane; // rows j0 + 2 lane, + 1 of the wave's window (j0 = j - lane); rows n, n + 1 wrap to 0, 1
    if (lane == 32 && ((j - lane + 64) & (n - 1)) == 0) rows -= n;
    rounds_fetch_window(rows, n, c_windows[W0].reg, lane, img);
    fp ta = 0, tb = 0;
#pragma unroll 1
    for (int wdx = 0; wdx < NWIN; wdx++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fp d[14];
#pragma unroll
        for (int jj = 0; jj < 14; jj++) d[jj] = fp_sub(img[jj * RW_ROWS + lane + 1], ark2[jj]);
        Acc128 sa = acc_zero(), sb = acc_zero();
        const CS_CONST fp *ga = gt + (wdx * 2) * 14;
#pragma unroll 1
        for (int i = 0; i < 14; i++) {
            const fp s2 = fp_cube(dot14l(ml + i * 56, d));
            acc_mad(sa, ga[i], s2);
            acc_mad(sb, ga[14 + i], s2);
            if (i == 6) { acc_fold(sa); acc_fold(sb); }
        }
        fp cube[14];
#pragma unroll
        for (int jj = 0; jj < 14; jj++) cube[jj] = fp_cube(img[jj * RW_ROWS + lane]);
        if (wdx + 1 < NWIN) { // the image is free again: the next window arrives behind the forward half
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            rounds_fetch_window(rows, n, c_windows[W0 + wdx + 1].reg, lane, img);
        }
        acc_fold(sa); acc_fold(sb);
        const fp fa = fp_add(dot14l(ul + (wdx * 2) * 56, cube), atab[(wdx * 2) * 8]);
        const fp fb = fp_add(dot14l(ul + (wdx * 2 + 1) * 56, cube), atab[(wdx * 2 + 1) * 8]);
        ta = fp_add(ta, fp_sub(acc_reduce(sa), fa));
        tb = fp_add(tb, fp_sub(acc_reduce(sb), fb));
    }
    const fp v = fp_mul(flag, fp_add(ta, fp_mul(xp, tb)));
    fp *o = p.out + (size_t)kk * n + j;
    *o = ADD ? fp_add(*o, v) : v;
}

// RangeProofAir::evaluate_transition  src/range/air.rs:60-98 (enforce_double_and_add_step with flag ONE)
__global__ void k_eval_transitions_range(const fp *lde, fp *out, unsigned log_n) {
    const size_t n = (size_t)1 << log_n;
    const size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (j >= n) return;
    const unsigned kk = blockIdx.y;
    const fp *base = lde + (size_t)kk * 2 * n;
    const size_t jn = (j + 1) & (n - 1);
    const fp nb = base[jn];
    out[((size_t)kk * 2 + 1) * n + j] = fp_sub(base[n + jn], fp_add(fp_dbl(base[n + j]), nb)); // result[1]: accumulator
    out[((size_t)kk * 2 + 0) * n + j] = c_is_binary(nb);                                         // result[0]: bit
}

struct AssertShifts { fp s[8]; };
// Generic merge of materialised transition evaluations with single-step boundary constraints:
//   out = sum_i (alpha_i + beta_i x^adj_i) C_i(x) / Z(x) + sum_a (T_reg(x) - v_a)(alpha_a + beta_a x^badj) / (x - w^step_a)
// Cosets outside the constraint-evaluation domain (k % stride != 0) are written as 0.
__global__ void k_air_combine(AirCombineParams p) {
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (j >= n) return;
    const unsigned kk = blockIdx.y, k = p.k0 + kk;
    fp *o = p.out + (size_t)kk * n + j;
    if (k % p.stride) { *o = 0; return; }
    const fp x = fp_mul(p.shifts[k], p.w[j]);
    // The degree adjustments take a handful of distinct values and the assertions share a few divisors: one power per group and
    // point -- x^e = shift_k^e w_n^((j e) mod n), a table product instead of a square-and-multiply chain -- and ONE inversion per
    // point for all assertion divisors (Montgomery's trick); 1 / (x^n - 1) is constant over a coset.
    const size_t mask = n - 1;
    fp xp[AIR_MAX_GROUPS], zi[AIR_MAX_GROUPS], xb[AIR_MAX_GROUPS];
    if (!p.tsum)
        for (unsigned g = 0; g < p.n_tgrp; g++) xp[g] = fp_mul(p.tgrp_shift[k][g], p.w[(j * p.tgrp_adj[g]) & mask]);
    if (p.agrp_inv[0]) { // the divisors depend on the domain only: cached inverses
        for (unsigned g = 0; g < p.n_agrp; g++) {
            const size_t period = n / p.agrp_m[g];
            zi[g] = p.agrp_inv[g][(size_t)k * period + (j & (period - 1))];
            xb[g] = fp_mul(p.agrp_bshift[k][g], p.w[(j * p.agrp_badj[g]) & mask]);
        }
    } else {
        fp den[AIR_MAX_GROUPS], pre[AIR_MAX_GROUPS], run = FP_ONE;
        for (unsigned g = 0; g < p.n_agrp; g++) {
            den[g] = fp_sub(fp_mul(p.agrp_mshift[k][g], p.w[(j * p.agrp_m[g]) & mask]), p.agrp_zc[g]);
            xb[g] = fp_mul(p.agrp_bshift[k][g], p.w[(j * p.agrp_badj[g]) & mask]);
            pre[g] = run;              // product of the denominators before g
            run = fp_mul(run, den[g]);
        }
        fp inv = fp_inv(run);
        for (unsigned g = p.n_agrp; g-- > 0;) {
            zi[g] = fp_mul(inv, pre[g]);
            inv = fp_mul(inv, den[g]);
        }
    }
    fp acc = 0;
    if (p.tsum) acc = p.tsum[(size_t)kk * n + j]; // merged by a fused evaluator (may be the output table itself: read before the write below)
    else
        for (unsigned i = 0; i < p.n_constraints; i++)
            acc = fp_add(acc, fp_mul(p.evals[((size_t)kk * p.n_constraints + i) * n + j], fp_add(p.t_alpha[i], fp_mul(p.t_beta[i], xp[p.t_grp[i]]))));
    acc = fp_mul(acc, fp_mul(fp_sub(x, p.w_last), p.zinv_coset[k]));
    // boundary constraints (single, periodic and sequence assertions): divisor x^m - w^(first*m).  Group by group (the assertion
    // tables are uniform over the wave, so the skip is a scalar branch): sum_a (alpha_a + beta_a x^badj)(T_a - v_a) =
    // sum_a alpha_a d_a + x^badj sum_a beta_a d_a, two 128-bit multiply-accumulates per assertion and one reduction pair per group.
    for (unsigned g = 0; g < p.n_agrp; g++) {
        Acc128 sa = acc_zero(), sb = acc_zero();
        int cnt = 0;
        for (unsigned a = 0; a < p.n_assertions; a++) {
            if (p.a_grp[a] != g) continue;
            const fp tv = p.lde[((size_t)kk * p.width + p.a_reg[a]) * n + j];
            const fp cv = p.a_seq[a] >= 0 ? p.avals[((size_t)kk * p.n_avals + p.a_seq[a]) * n + j] : p.a_value[a];
            const fp d = fp_sub(tv, cv);
            acc_mad(sa, p.b_alpha[a], d);
            acc_mad(sb, p.b_beta[a], d);
            if (++cnt == 7) { acc_fold(sa); acc_fold(sb); cnt = 0; }
        }
        acc_fold(sa); acc_fold(sb);
        acc = fp_add(acc, fp_mul(fp_add(acc_reduce(sa), fp_mul(xb[g], acc_reduce(sb))), zi[g]));
    }
    *o = acc;
}

} // namespace

hipError_t launch_eval_transitions(const CeParams &p, unsigned nk, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    hipError_t e = hipMemsetAsync(p.out, 0, (size_t)nk * 115 * n * sizeof(fp), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_eval_transitions, dim3((unsigned)(n / NT), nk), dim3(NT), 0, stream, p);
    return hipGetLastError();
}
hipError_t launch_eval_transitions_merkle(const uint64_t *lde, const uint64_t *ptab, uint64_t *out, unsigned log_n, unsigned k0, unsigned nk,
                                          hipStream_t stream) {
    const size_t n = (size_t)1 << log_n;
    hipError_t e = hipMemsetAsync(out, 0, (size_t)nk * 106 * n * sizeof(fp), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_eval_transitions_merkle, dim3((unsigned)(n / NT), nk), dim3(NT), 0, stream, lde, ptab, out, log_n, k0);
    return hipGetLastError();
}
hipError_t launch_eval_transitions_schnorr(const uint64_t *lde, const uint64_t *aux, const uint64_t *ptab, uint64_t *out, unsigned log_n, unsigned k0,
                                           unsigned nk, hipStream_t stream) {
    const size_t n = (size_t)1 << log_n;
    hipError_t e = hipMemsetAsync(out, 0, (size_t)nk * 56 * n * sizeof(fp), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_eval_transitions_schnorr<0>, dim3((unsigned)(n / NT), nk), dim3(NT), 0, stream, lde, aux, ptab, out, log_n, k0);
    hipLaunchKernelGGL(k_eval_transitions_schnorr<1>, dim3((unsigned)(n / NT), nk), dim3(NT), 0, stream, lde, aux, ptab, out, log_n, k0);
    hipLaunchKernelGGL(k_eval_transitions_schnorr<2>, dim3((unsigned)(n / NT), nk), dim3(NT), 0, stream, lde, aux, ptab, out, log_n, k0);
    hipLaunchKernelGGL(k_eval_transitions_schnorr<3>, dim3((unsigned)(n / NT), nk), dim3(NT), 0, stream, lde, aux, ptab, out, log_n, k0);
    return hipGetLastError();
}
hipError_t launch_schnorr_fused(const AirCombineParams &p, const uint64_t *aux, const uint64_t *ptab, unsigned nk, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    if (n % FNT) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(n / FNT), nk), block(FNT);
    hipLaunchKernelGGL(k_schnorr_fused<SF_DBL0>, grid, block, 0, stream, p, aux, ptab);
    hipLaunchKernelGGL(k_schnorr_fused<SF_ADD0>, grid, block, 0, stream, p, aux, ptab);
    hipLaunchKernelGGL(k_schnorr_fused<SF_DBL1>, grid, block, 0, stream, p, aux, ptab);
    hipLaunchKernelGGL(k_schnorr_fused<SF_ADD1>, grid, block, 0, stream, p, aux, ptab);
    hipLaunchKernelGGL(k_schnorr_fused<SF_FINAL>, grid, block, 0, stream, p, aux, ptab);
    hipLaunchKernelGGL(k_schnorr_fused<SF_REST>, grid, block, 0, stream, p, aux, ptab);
    return hipGetLastError();
}
// SchnorrAir's curve gadgets in the degree-split form (all eight cosets, k0 = 0).  1: launch_schnorr_ec_split -- the eight polynomials on
// the even cosets, d_even = [8][4][n]; the caller interpolates them per coset, carries them to the odd cosets (ntt.h: coset_even_to_odd)
// and extends them, d_odd = [4][8][n]; 2: launch_schnorr_split_finish -- recombination into p.out, then the remaining parts add to it.
hipError_t launch_schnorr_ec_split(const AirCombineParams &p, const uint64_t *aux, const uint64_t *d_coefs_tx_layout, uint64_t *d_even, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    if (n % FNT) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(n / FNT), 4), block(FNT);
    hipLaunchKernelGGL(k_schnorr_ec_split<PART_DBL0>, grid, block, 0, stream, p.lde, aux, d_coefs_tx_layout, d_even, p.log_n);
    hipLaunchKernelGGL(k_schnorr_ec_split<PART_DBL1>, grid, block, 0, stream, p.lde, aux, d_coefs_tx_layout, d_even, p.log_n);
    hipLaunchKernelGGL(k_schnorr_ec_split<PART_ADD0>, grid, block, 0, stream, p.lde, aux, d_coefs_tx_layout, d_even, p.log_n);
    hipLaunchKernelGGL(k_schnorr_ec_split<PART_ADD1>, grid, block, 0, stream, p.lde, aux, d_coefs_tx_layout, d_even, p.log_n);
    return hipGetLastError();
}
hipError_t launch_schnorr_final_split(const AirCombineParams &p, const uint64_t *d_coefs_tx_layout, uint64_t *d_out, int coset, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    if (n % FNT) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_schnorr_final_split, dim3((unsigned)(n / FNT), coset < 0 ? 4 : 1), dim3(FNT), 0, stream, p.lde, d_coefs_tx_layout, d_out, p.log_n, coset);
    return hipGetLastError();
}
hipError_t launch_schnorr_final_hi(const AirCombineParams &p, const uint64_t *d_odd, const uint64_t *d_direct, uint64_t *d_hi, uint64_t half_m, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    hipLaunchKernelGGL(k_schnorr_final_hi, dim3((unsigned)(n / 256), 3), dim3(256), 0, stream, d_odd, d_direct, d_hi, half_m, p.log_n);
    return hipGetLastError();
}
// the folded round gadgets of the sub-AIRs on the matrix cores (rounds_mfma.hip; same values).  CSTARK_ROUNDS_MFMA=0: vector-ALU kernels
static bool merkle_rounds_on_matrix_cores(size_t n) {
    static const bool env = [] { const char *e = getenv("CSTARK_ROUNDS_MFMA"); return !e || atoi(e) != 0; }();
    return env && n % 256 == 0;
}
hipError_t launch_schnorr_split_finish(const AirCombineParams &p, const uint64_t *aux, const uint64_t *ptab, const uint64_t *d_even, const uint64_t *d_odd,
                                       unsigned g0, unsigned g1, hipStream_t stream, uint64_t *d_rtab, unsigned round_group, const uint64_t *d_hi) {
    const size_t n = (size_t)1 << p.log_n;
    if (n % FNT) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(n / FNT), 8), block(FNT);
    if (d_hi) { // tables of SCH_SPLIT_TABLES: the final addition came with them
        hipLaunchKernelGGL(k_schnorr_split_finish<SCH_SPLIT_TABLES>, dim3((unsigned)(n / 256), 8), dim3(256), 0, stream, p, d_even, d_odd, ptab, g0, g1, d_hi);
    } else {    // tables of SCH_SPLIT_EC_TABLES
        hipLaunchKernelGGL(k_schnorr_split_finish<SCH_SPLIT_EC_TABLES>, dim3((unsigned)(n / 256), 8), dim3(256), 0, stream, p, d_even, d_odd, ptab, g0, g1, d_hi);
        hipLaunchKernelGGL((k_schnorr_fused<SF_FINAL, true>), grid, block, 0, stream, p, aux, ptab);
    }
    if (d_rtab) { // the message hash's round gadget in the folded form (as MerkleAir's four), added to the output
        hipLaunchKernelGGL((k_merkle_rounds_setup<4, 36, 8>), dim3(2), dim3(64), 0, stream, p.t_alpha, p.t_beta, ptab, d_rtab, 8u);
        if (merkle_rounds_on_matrix_cores(n)) { const hipError_t e = launch_merkle_rounds_mfma(p, ptab, d_rtab, 8, round_group, 1, stream); if (e != hipSuccess) return e; }
        else hipLaunchKernelGGL((k_merkle_rounds<4, 1, 56, 36, 7, 8, true>), grid, block, 0, stream, p, ptab, (const fp *)d_rtab, round_group);
        hipLaunchKernelGGL((k_schnorr_fused<SF_REST, true, true>), grid, block, 0, stream, p, aux, ptab);
    } else {
        hipLaunchKernelGGL((k_schnorr_fused<SF_REST, true>), grid, block, 0, stream, p, aux, ptab);
    }
    return hipGetLastError();
}
hipError_t launch_merkle_fused(const AirCombineParams &p, const uint64_t *ptab, unsigned nk, hipStream_t stream, uint64_t *d_rtab, unsigned round_group) {
    const size_t n = (size_t)1 << p.log_n;
    if (n % FNT) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(n / FNT), nk);
    if (!d_rtab) {
        hipLaunchKernelGGL(k_merkle_fused<false>, grid, dim3(FNT), 0, stream, p, ptab);
        return hipGetLastError();
    }
    if (p.k0 + nk > 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_merkle_rounds_setup<0, 33, 5>), dim3(MR_SECTIONS), dim3(64), 0, stream, p.t_alpha, p.t_beta, ptab, d_rtab, p.k0 + nk);
    if (merkle_rounds_on_matrix_cores(n)) { const hipError_t e = launch_merkle_rounds_mfma(p, ptab, d_rtab, nk, round_group, 0, stream); if (e != hipSuccess) return e; }
    else hipLaunchKernelGGL((k_merkle_rounds<0, 4, 65, 33, 4, 5, false>), grid, dim3(FNT), 0, stream, p, ptab, (const fp *)d_rtab, round_group);
    hipLaunchKernelGGL(k_merkle_fused<true>, grid, dim3(FNT), 0, stream, p, ptab);
    return hipGetLastError();
}
// RescueAir::evaluate_transition, benches/rescue.rs:200-222 (+ enforce_hash_copy :254-264): the round gadget under the cycle mask, the
// copy of the rate half / reset of the capacity half under its complement.  ptab: [b][29][8] (mask, 28 round constants)
__global__ __launch_bounds__(NT) void k_eval_transitions_rescue(const fp *lde, const fp *ptab, fp *out, unsigned log_n, unsigned k0) {
    const size_t n = (size_t)1 << log_n;
    const size_t j = blockIdx.x * (size_t)NT + threadIdx.x;
    if (j >= n) return;
    const unsigned kk = blockIdx.y;
    const fp *base = lde + (size_t)kk * 14 * n;
    Frame f;
    f.n = n;
    f.cur_p = base + j;
    f.next_p = base + ((j + 1) & (n - 1));
    f.per_p = ptab + (size_t)(k0 + kk) * 29 * 8 + (j & 7);
    f.pcycle = 8;
    AccAll acc{out + (size_t)kk * 14 * n + j, n};
    const fp hash_flag = f.pv(0), copy_flag = c_not(hash_flag);
    Frame fr = f; // the gadget reads the round constants at P_ARK + i relative to per_p: a view shifted by (1 - P_ARK) columns
    fr.per_p = f.per_p - (size_t)(P_ARK - 1) * 8;
    enforce_round(acc, fr, 0, 0, hash_flag, 0, 0, false);
    for (int i = 0; i < 7; i++) {
        acc.add(i, copy_flag, fp_sub(f.cur(i), f.next(i)));
        acc.add(7 + i, copy_flag, f.next(7 + i));
    }
}
hipError_t launch_eval_transitions_rescue(const uint64_t *lde, const uint64_t *ptab, uint64_t *out, unsigned log_n, unsigned k0, unsigned nk,
                                          hipStream_t stream) {
    const size_t n = (size_t)1 << log_n;
    hipError_t e = hipMemsetAsync(out, 0, (size_t)nk * 14 * n * sizeof(fp), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_eval_transitions_rescue, dim3((unsigned)((n + NT - 1) / NT), nk), dim3(NT), 0, stream, lde, ptab, out, log_n, k0);
    return hipGetLastError();
}
hipError_t launch_eval_transitions_range(const uint64_t *lde, uint64_t *out, unsigned log_n, unsigned nk, hipStream_t stream) {
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(k_eval_transitions_range, dim3((unsigned)((n + 63) / 64), nk), dim3(64), 0, stream, lde, out, log_n);
    return hipGetLastError();
}
__global__ void k_assert_inverses(fp *__restrict__ tab, const fp *__restrict__ w, AssertShifts sh, uint64_t m, fp zc, unsigned log_n) {
    const size_t n = (size_t)1 << log_n, period = n / m;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= period) return;
    const unsigned k = blockIdx.y;
    tab[(size_t)k * period + i] = fp_inv(fp_sub(fp_mul(sh.s[k], w[(i * m) & (n - 1)]), zc));
}
hipError_t launch_assert_inverses(uint64_t *d_tab, const uint64_t *d_w, const uint64_t shift_m[8], unsigned b, uint64_t m, uint64_t zc, unsigned log_n,
                                  hipStream_t stream) {
    const size_t period = ((size_t)1 << log_n) / m;
    AssertShifts sh;
    for (unsigned k = 0; k < 8; k++) sh.s[k] = k < b ? shift_m[k] : 0;
    hipLaunchKernelGGL(k_assert_inverses, dim3((unsigned)((period + 255) / 256), b), dim3(256), 0, stream, d_tab, d_w, sh, m, zc, log_n);
    return hipGetLastError();
}
hipError_t launch_air_combine(const AirCombineParams &p, unsigned nk, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    hipLaunchKernelGGL(k_air_combine, dim3((unsigned)((n + 63) / 64), nk), dim3(64), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_rounds_setup(const CeParams &p, hipStream_t stream) {
    const unsigned m = p.m ? p.m : 1;
    if (m > 3) return hipErrorInvalidValue;
#ifdef CS_ROUNDS_MFMA // the matrix-core table of the inverse matrix: 14 more blocks and a zero-fill, only for the opt-in variant
    (void)hipMemsetAsync(p.rtab + RT_MT, 0, MT_BYTES, stream);
    hipLaunchKernelGGL(k_rounds_setup, dim3(RT_SECTIONS + 14, m), dim3(64), 0, stream, p.coef, p.ptab, p.rtab);
#else
    hipLaunchKernelGGL(k_rounds_setup, dim3(RT_SECTIONS, m), dim3(64), 0, stream, p.coef, p.ptab, p.rtab);
#endif
    return hipGetLastError();
}
hipError_t launch_rounds_split(const CeParams &p, uint64_t *d_even, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    const dim3 grid((unsigned)(n / FNT), p.nkc ? p.nkc : 4), block(FNT);
    const unsigned m = p.m ? p.m : 1;
    // the matrix-core kernel (rounds_mfma.hip; same values).  CSTARK_ROUNDS_MFMA=0: the vector-ALU kernels below
    static const bool mfma_env = [] { const char *e = getenv("CSTARK_ROUNDS_MFMA"); return !e || atoi(e) != 0; }();
    if (m <= 3 && mfma_env && n % 512 == 0) return launch_rounds_mfma(p, d_even, stream);
    if (m == 1) hipLaunchKernelGGL(k_rounds_split<1>, grid, block, 0, stream, p, d_even);
    else if (m == 2) hipLaunchKernelGGL(k_rounds_split<2>, grid, block, 0, stream, p, d_even);
    else if (m == 3) hipLaunchKernelGGL(k_rounds_split<3>, grid, block, 0, stream, p, d_even);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t launch_ec_split(const CeParams &p, int part, uint64_t *d_even_family, uint64_t *d_even_linear, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    const dim3 grid((unsigned)(n / FNT), p.nkc ? p.nkc : 4), block(FNT);
    const unsigned m = p.m ? p.m : 1;
#define CS_EC(M)                                                                                                                                      \
    if (part == PART_DBL0) hipLaunchKernelGGL((k_ec_split<PART_DBL0, false, M>), grid, block, 0, stream, p, d_even_family, d_even_linear);           \
    else if (part == PART_DBL1) hipLaunchKernelGGL((k_ec_split<PART_DBL1, true, M>), grid, block, 0, stream, p, d_even_family, d_even_linear);       \
    else if (part == PART_ADD0) hipLaunchKernelGGL((k_ec_split<PART_ADD0, false, M>), grid, block, 0, stream, p, d_even_family, d_even_linear);      \
    else if (part == PART_ADD1) hipLaunchKernelGGL((k_ec_split<PART_ADD1, false, M>), grid, block, 0, stream, p, d_even_family, d_even_linear);      \
    else return hipErrorInvalidValue;
    if (m == 1) { CS_EC(1) } else if (m == 2) { CS_EC(2) } else if (m == 3) { CS_EC(3) } else return hipErrorInvalidValue;
#undef CS_EC
    return hipGetLastError();
}
hipError_t launch_lin_all(const CeParams &p, uint64_t *d_even_family0, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    const dim3 grid((unsigned)(n / FNT), p.nkc ? p.nkc : 4), block(FNT);
    const unsigned m = p.m ? p.m : 1;
    const bool buf = p.log_n <= 22; // 94 n 8 bytes < 2^32: FrameB
    for (unsigned set = 0; set < m; set++) {
        if (buf) hipLaunchKernelGGL(k_lin_all<true>, grid, block, 0, stream, p, d_even_family0, set);
        else hipLaunchKernelGGL(k_lin_all<false>, grid, block, 0, stream, p, d_even_family0, set);
    }
    return hipGetLastError();
}
hipError_t launch_lin_split(const CeParams &p, int part, uint64_t *d_even_family0, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    const dim3 grid((unsigned)(n / FNT), p.nkc ? p.nkc : 4), block(FNT);
    const unsigned m = p.m ? p.m : 1;
    for (unsigned set = 0; set < m; set++) {
        if (part == PART_LIN_A) hipLaunchKernelGGL(k_lin_split<PART_LIN_A>, grid, block, 0, stream, p, d_even_family0, set);
        else if (part == PART_LIN_B) hipLaunchKernelGGL(k_lin_split<PART_LIN_B>, grid, block, 0, stream, p, d_even_family0, set);
        else if (part == PART_LIN_C) hipLaunchKernelGGL(k_lin_split<PART_LIN_C>, grid, block, 0, stream, p, d_even_family0, set);
        else return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_split_finish(const CeParams &p, const uint64_t *d_even, const uint64_t *d_odd, const uint64_t *d_hi, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    const dim3 grid((unsigned)(n / 256), 8), block(256);
    const unsigned m = p.m ? p.m : 1;
    if (m == 1) hipLaunchKernelGGL(k_split_finish<1>, grid, block, 0, stream, p, d_even, d_odd, d_hi);
    else if (m == 2) hipLaunchKernelGGL(k_split_finish<2>, grid, block, 0, stream, p, d_even, d_odd, d_hi);
    else if (m == 3) hipLaunchKernelGGL(k_split_finish<3>, grid, block, 0, stream, p, d_even, d_odd, d_hi);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t launch_split_finish_shard(const CeParams &p, const uint64_t *d_even, const uint64_t *d_odd, const uint64_t *d_hi, const uint64_t *d_bit37_all,
                                     uint64_t *d_out, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    if (p.nkc != 1 && p.nkc != 2) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_split_finish_shard, dim3((unsigned)(n / 256), p.nkc + 4), dim3(256), 0, stream, p, d_even, d_odd, d_hi, d_bit37_all, d_out);
    return hipGetLastError();
}
hipError_t launch_shard_combine(const uint64_t *d_parts, uint64_t *d_out, unsigned log_n, unsigned nkc, hipStream_t stream) {
    const size_t n = (size_t)1 << log_n;
    if (nkc != 1 && nkc != 2) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_shard_combine, dim3((unsigned)(n / 256), 8), dim3(256), 0, stream, d_parts, d_out, log_n, nkc);
    return hipGetLastError();
}
hipError_t launch_final_split(const CeParams &p, int coset, uint64_t *d_out, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    const unsigned even_rows = p.nkc ? p.nkc : 4;
    const dim3 grid((unsigned)(n / FNT), coset < 0 ? even_rows : 1), block(FNT);
    const unsigned m = p.m ? p.m : 1;
    if (m == 1) hipLaunchKernelGGL(k_final_split<1>, grid, block, 0, stream, p, d_out, coset);
    else if (m == 2) hipLaunchKernelGGL(k_final_split<2>, grid, block, 0, stream, p, d_out, coset);
    else if (m == 3) hipLaunchKernelGGL(k_final_split<3>, grid, block, 0, stream, p, d_out, coset);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t launch_final_hi(const CeParams &p, const uint64_t *d_odd, const uint64_t *d_direct, uint64_t *d_hi, uint64_t half_m, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    const unsigned m = p.m ? p.m : 1;
    const dim3 grid((unsigned)(n / 256), 2 * m), block(256);
    if (m == 1) hipLaunchKernelGGL(k_final_hi<1>, grid, block, 0, stream, p, d_odd, d_direct, d_hi, half_m);
    else if (m == 2) hipLaunchKernelGGL(k_final_hi<2>, grid, block, 0, stream, p, d_odd, d_direct, d_hi, half_m);
    else if (m == 3) hipLaunchKernelGGL(k_final_hi<3>, grid, block, 0, stream, p, d_odd, d_direct, d_hi, half_m);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_eval_constraints(const CeParams &p, unsigned nk, hipStream_t stream, hipEvent_t *part_events, unsigned done_mask, bool record_end) {
    const bool rounds_done = done_mask & 1u; // done_mask: bit PART = that part was evaluated by the caller (split evaluation), its event recorded
    const size_t n = (size_t)1 << p.log_n;
    const dim3 grid((unsigned)(n / FNT), nk), block(FNT);
    // part_events (optional, CE_NUM_PARTS + 1 events): recorded around every part so that callers can time each launch
    const unsigned m = p.m ? p.m : 1;
    if (m > 3) return hipErrorInvalidValue;
    // rounds_done: the caller ran launch_rounds_setup and the split evaluation of the Rescue windows (and recorded part_events[0])
    if (!rounds_done) {
        const hipError_t e = launch_rounds_setup(p, stream);
        if (e != hipSuccess) return e;
    }
#define CS_PART(PART)                                                                                                                   \
    if ((done_mask >> PART) & 1u) {                                                                                                     \
    } else if (part_events) (void)hipEventRecord(part_events[PART], stream);                                                            \
    if ((done_mask >> PART) & 1u) {                                                                                                     \
    } else if (m == 1) {                                                                                                                       \
        if (PART == PART_ROUNDS && ROUNDS_DYN_LDS)                                                                                      \
            (void)hipFuncSetAttribute((const void *)k_eval_fused<PART, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ROUNDS_LDS); \
        hipLaunchKernelGGL((k_eval_fused<PART, 1>), grid, block, PART == PART_ROUNDS ? ROUNDS_DYN_LDS : 0, stream, p);                    \
    } else if (m == 2) {                                                                                                                \
        if (PART == PART_ROUNDS && ROUNDS_DYN_LDS)                                                                                      \
            (void)hipFuncSetAttribute((const void *)k_eval_fused<PART, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ROUNDS_LDS); \
        hipLaunchKernelGGL((k_eval_fused<PART, 2>), grid, block, PART == PART_ROUNDS ? ROUNDS_DYN_LDS : 0, stream, p);                    \
    } else {                                                                                                                            \
        if (PART == PART_ROUNDS && ROUNDS_DYN_LDS)                                                                                      \
            (void)hipFuncSetAttribute((const void *)k_eval_fused<PART, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ROUNDS_LDS); \
        hipLaunchKernelGGL((k_eval_fused<PART, 3>), grid, block, PART == PART_ROUNDS ? ROUNDS_DYN_LDS : 0, stream, p);                    \
    }
    CS_PART(PART_ROUNDS) CS_PART(PART_DBL0) CS_PART(PART_ADD0) CS_PART(PART_DBL1) CS_PART(PART_ADD1) CS_PART(PART_FINAL)
    CS_PART(PART_LIN_A) CS_PART(PART_LIN_B) CS_PART(PART_LIN_C)
#undef CS_PART
    if (part_events && record_end) (void)hipEventRecord(part_events[NUM_PARTS], stream);
    static_assert(NUM_PARTS == CE_NUM_PARTS, "part count");
    return hipGetLastError();
}

// table[k][0][j] = 1 / (x - 1), table[k][1][j] = 1 / (x - w^(n-1)), x = shift_k * w^j   (grid = (n / 256, b))
__global__ void k_boundary_inverses(fp *table, const fp *w, const fp *coset, fp w_last, unsigned log_n) {
    const size_t n = (size_t)1 << log_n;
    const size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const unsigned k = blockIdx.y;
    const fp x = fp_mul(coset[(size_t)k * CE_COSET_CONSTS], w[j]);
    const fp d0 = fp_sub(x, FP_ONE), d1 = fp_sub(x, w_last);
    const fp inv01 = fp_inv(fp_mul(d0, d1)); // one inversion for both
    table[((size_t)k * 2 + 0) * n + j] = fp_mul(inv01, d1);
    table[((size_t)k * 2 + 1) * n + j] = fp_mul(inv01, d0);
}
hipError_t build_boundary_inverses(uint64_t *d_table, const uint64_t *d_w, const uint64_t *d_coset, uint64_t w_last, unsigned log_n, unsigned log_b,
                                   hipStream_t stream) {
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(k_boundary_inverses, dim3((unsigned)(n / 256), 1u << log_b), dim3(256), 0, stream, d_table, d_w, d_coset, w_last, log_n);
    return hipGetLastError();
}

} // namespace cs
