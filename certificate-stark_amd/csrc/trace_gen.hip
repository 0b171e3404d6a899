// K1 -- execution-trace generation for the state-transition AIR on gfx950.
//
// Replaces TransactionProver::build_trace (/root/reference/src/prover.rs:37-98): the reference fills
// one 1024-row fragment per transaction with a sequential init/update loop (src/trace.rs:28-142).
// Here the 94 registers are split by the chain that produces them, because each chain is an
// independent, latency-bound recurrence:
//   k_trace_merkle        regs 0..64, rows 0..511 (+ root copy rows 512..1023): four Rescue states,
//                         one lane per state element      (src/merkle/update/trace.rs:19-136)
//   k_trace_schnorr_hash  regs 42..55, rows 512..1023: the 5-permutation message hash, also yields h
//                         (src/schnorr/trace.rs:47-67, src/schnorr/mod.rs:247-288)
//   k_trace_schnorr_ec    regs 0..17 and 19..36, rows 512..1023: the two double-and-add ladders,
//                         one lane per F_p6 product       (src/schnorr/trace.rs:69-121, src/utils/ecc.rs)
//   k_trace_aux           every register that is a closed form of the witness (bit registers, limb and
//                         range accumulators, key/delta/sigma/nonce copies; src/trace.rs:43-53,
//                         src/range/prover.rs:65-84, src/utils/field.rs:16-22)
// Rows are staged in LDS 64 at a time and written column-wise, so every global store is a
// contiguous 512-byte line of one column (the trace is column-major, 94 x N).
#include <stdlib.h>
#include "trace_gen.h"
#include "rescue.cuh"
#include "tower.cuh"

namespace cs {

namespace {

constexpr int TXC = 1024;       // TRANSACTION_CYCLE_LENGTH, src/constants.rs:83
constexpr int MERKLE_LEN = 512; // src/merkle/constants.rs:29
constexpr unsigned LADDER_LDS_PAD = 0;
#ifndef CS_RECUR_TILE_ROWS
#define CS_RECUR_TILE_ROWS 16 // staging rows of the Merkle / message-hash recurrences in the overlapped path
#endif
#ifndef CS_RECUR_PRIO
#define CS_RECUR_PRIO 3 // wave priority of the latency-bound recurrences (0..3)
#endif
constexpr int SCALAR_MUL_LEN = 510; // src/schnorr/constants.rs:30

// this lane's row of the MDS matrix against the state in LDS: 128-bit accumulation of the 14 products, one reduction (572 instead of
// 936 issue cycles for 14 reduced products and 13 modular additions)
__device__ __forceinline__ fp mds_row_dot(const fp (&mrow)[14], const fp *xch) {
    Acc128 a = acc_zero();
#pragma unroll
    for (int j = 0; j < 7; j++) acc_mad(a, mrow[j], xch[j]);
    acc_fold(a);
#pragma unroll
    for (int j = 7; j < 14; j++) acc_mad(a, mrow[j], xch[j]);
    acc_fold(a);
    return acc_reduce(a);
}

// One Rescue round on a 14-element state held one element per lane (rescue.rs:246-263).
// `xch` is the state's 14-word LDS exchange row.  Must be called by every lane of the block
// (it synchronises); lanes with active == false only take part in the barriers.
__device__ __forceinline__ fp rescue_round_lane(fp v, fp *xch, const fp (&mrow)[14], int e, int cyc, bool active) {
    fp x = 0;
    if (active) { x = fp_cube(v); xch[e] = x; }
    __syncthreads();
    if (active) x = fp_inv_sbox(fp_add(mds_row_dot(mrow, xch), c_ark[cyc * 28 + e]));
    __syncthreads();
    if (active) xch[e] = x;
    __syncthreads();
    if (active) v = fp_add(mds_row_dot(mrow, xch), c_ark[cyc * 28 + 14 + e]);
    __syncthreads();
    return v;
}

template <int NCOLS, int LD>
__device__ __forceinline__ void flush_tile(const fp (*tile)[LD], fp *__restrict__ trace, size_t n, size_t row0, int c0, int lane,
                                           int skip_at = -1) {
    // tile column c -> trace column c0 + c (+1 past skip_at); lane = row within the 64-row tile
    for (int c = 0; c < NCOLS; c++) {
        int tc = c0 + c + ((skip_at >= 0 && c >= skip_at) ? 1 : 0);
        trace[(size_t)tc * n + row0 + lane] = tile[lane][c];
    }
}

// TR-row staging tile (TR | 64): lane = (row, column group); one wave
template <int NCOLS, int LD, int TR>
__device__ __forceinline__ void flush_rows(const fp (*tile)[LD], fp *__restrict__ trace, size_t n, size_t row0, int c0, int lane) {
    for (int c = lane / TR; c < NCOLS; c += 64 / TR) trace[(size_t)(c0 + c) * n + row0 + (lane & (TR - 1))] = tile[lane & (TR - 1)][c];
}

// ---------------------------------------------------------------------------------------------------
// STANDALONE: the 65-register, 512-rows-per-transaction trace of MerkleProver (src/merkle/update/prover.rs:28-80)
// TR = rows of the staging tile: 64 alone; 16 when the recurrence runs beside the transforms (prove.hip), whose workgroups need the
// LDS (a 64-row tile is 33 KB: four resident recurrences per CU leave no room for a 75 KB transform workgroup)
template <bool STANDALONE, int TR>
__global__ __launch_bounds__(64) void k_trace_merkle(TxWitnessDev w, fp *__restrict__ trace, size_t n) {
    __builtin_amdgcn_s_setprio(CS_RECUR_PRIO); // latency-bound recurrence: issue ahead of the chip-filling kernels it runs beside
    __shared__ fp tile[TR][65];
    __shared__ fp st[4][14];
    const int t = blockIdx.x, lane = threadIdx.x;
    const int hash_len = 8 * (int)w.depth + 7; // TRANSACTION_HASH_LENGTH
    const bool is_state = lane < 56;
    const int s = is_state ? lane / 14 : 0, e = is_state ? lane % 14 : 0;
    // lanes 0..55: state elements; lane 56: both index-bit registers; lanes 57..63: previous-root registers 58..64
    const int col = is_state ? (s == 0 ? 0 : s == 1 ? 15 : s == 2 ? 29 : 44) + e : (lane == 56 ? 14 : lane + 1);
    fp mrow[14];
#pragma unroll
    for (int j = 0; j < 14; j++) mrow[j] = c_mds[e * 14 + j];

    const fp *sv = w.s_old + 14 * (size_t)t, *rv = w.r_old + 14 * (size_t)t;
    const fp delta = w.deltas[t];
    const uint64_t s_index = w.s_idx[t], r_index = w.r_idx[t];
    const fp *s_branch = w.s_paths + (size_t)7 * (w.depth + 1) * t, *r_branch = w.r_paths + (size_t)7 * (w.depth + 1) * t;

    // row 0: init_merkle_update_state, src/merkle/update/trace.rs:19-48
    fp v = 0, v2 = 0;
    if (is_state) {
        v = (s < 2 ? sv : rv)[e];
        if (s == 1 && e == 12) v = fp_sub(v, delta);
        if (s == 1 && e == 13) v = fp_add(v, FP_ONE);
        if (s == 3 && e == 12) v = fp_add(v, delta);
    } else if (lane >= 57) {
        v = w.initial_roots[7 * (size_t)t + lane - 57];
    }
    tile[0][col] = v;
    if (lane == 56) tile[0][43] = 0;

    const size_t gbase = (size_t)t * (STANDALONE ? MERKLE_LEN : TXC);
    for (int step = 0; step < MERKLE_LEN - 1; step++) {
        if (step < hash_len) {
            const int cyc = step & 7, lvl = step >> 3;
            if (cyc < 7) {
                v = rescue_round_lane(v, st[s], mrow, e, cyc, is_state);
            } else {
                // sibling insertion, src/merkle/update/trace.rs:112-135
                if (is_state) st[s][e] = v;
                __syncthreads();
                if (is_state) {
                    const fp *node = (s < 2 ? s_branch : r_branch) + 7 * (lvl + 1);
                    const int bit = (int)(((s < 2 ? s_index : r_index) >> lvl) & 1);
                    if (!bit) { if (e >= 7) v = node[e - 7]; }
                    else v = e >= 7 ? st[s][e - 7] : node[e];
                } else if (lane == 56) {
                    v = ((s_index >> lvl) & 1) ? FP_ONE : 0;
                    v2 = ((r_index >> lvl) & 1) ? FP_ONE : 0;
                }
                __syncthreads();
            }
            if (step == hash_len - 1) { // root copy, src/merkle/update/trace.rs:87-93
                if (is_state) st[s][e] = v;
                __syncthreads();
                if (lane >= 57) v = st[3][lane - 57];
                __syncthreads();
            }
        }
        const int r = (step + 1) & (TR - 1);
        tile[r][col] = v;
        if (lane == 56) tile[r][43] = v2;
        if (r == TR - 1) {
            __syncthreads();
            flush_rows<65, 65, TR>(tile, trace, n, gbase + (step + 1 - (TR - 1)), 0, lane);
            __syncthreads();
        }
    }
    if (STANDALONE) {
        // index-bit poke at global row 1 (src/merkle/update/prover.rs:72-77); after this workgroup's own flushes
        if (t == 0 && lane == 0) { trace[(size_t)14 * n + 1] = FP_ONE; trace[(size_t)43 * n + 1] = FP_ONE; }
        return;
    }
    // registers 58..64 keep the new root through the Schnorr half (rows 512..1023)
    if (lane >= 57) st[0][lane - 57] = v;
    __syncthreads();
    for (int k = 0; k < 8; k++)
        for (int c = 58; c < 65; c++) trace[(size_t)c * n + gbase + MERKLE_LEN + 64 * k + lane] = st[0][c - 58];
}

// ---------------------------------------------------------------------------------------------------
template <bool STANDALONE, int TR>
__global__ __launch_bounds__(64) void k_trace_schnorr_hash(TxWitnessDev w, fp *__restrict__ trace, size_t n) {
    __builtin_amdgcn_s_setprio(CS_RECUR_PRIO); // latency-bound recurrence: issue ahead of the chip-filling kernels it runs beside
    __shared__ fp tile[TR][15];
    __shared__ fp st[14];
    const int t = blockIdx.x, lane = threadIdx.x;
    const bool active = lane < 14;
    const int e = active ? lane : 0;
    fp mrow[14];
#pragma unroll
    for (int j = 0; j < 14; j++) mrow[j] = c_mds[e * 14 + j];
    const fp *sv = w.s_old + 14 * (size_t)t, *rv = w.r_old + 14 * (size_t)t;

    // row 512: init_sig_verification_state, src/schnorr/trace.rs:18-30
    fp v = (active && e < 6) ? w.sig_rx[6 * (size_t)t + e] : 0;
    if (active) tile[0][e] = v;
    const size_t gbase = STANDALONE ? (size_t)t * MERKLE_LEN : (size_t)t * TXC + MERKLE_LEN;
    for (int step = 0; step < MERKLE_LEN - 1; step++) {
        if (step < 40) { // TOTAL_HASH_LENGTH
            const int cyc = step & 7;
            if (cyc < 7) {
                v = rescue_round_lane(v, st, mrow, e, cyc, active);
            } else if (step < 32) { // message chunk, src/lib.rs:467-481 layout
                if (active && e >= 7) {
                    int m = 7 * (step >> 3) + e - 7;
                    v = m < 12 ? sv[m] : m < 24 ? rv[m - 12] : m == 24 ? w.deltas[t] : m == 25 ? sv[13] : (STANDALONE ? w.msg_tail[2 * (size_t)t + m - 26] : 0);
                }
            } else {
                if (active && e >= 7) v = 0;
            }
            if (step == 38 && active && e < 4) w.h_limbs[4 * (size_t)t + e] = fp_to_u64(v); // build_sig_info h_bytes
        }
        const int r = (step + 1) & (TR - 1);
        if (active) tile[r][e] = v;
        if (r == TR - 1) {
            __syncthreads();
            flush_rows<14, 15, TR>(tile, trace, n, gbase + (step + 1 - (TR - 1)), 42, lane);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Curve ladders.  A point operation is a short program of F_p6 products whose operands are small
// integer combinations of LDS-resident F_p6 "slots"; each product is done by one lane.
struct Term { int8_t slot, coef; };
struct Instr { int8_t out; Term a[3]; Term b[3]; }; // out < 0: no-op; b[0].coef == 0: linear (no product)

enum { SX = 0, SY = 1, SZ = 2, SB3 = 3, SPX = 4, SPY = 5, QX = 27, QY = 28, QZ = 29, NSLOT = 30 };
enum { OP_DOUBLE = 0, OP_ADD_MIXED = 1, OP_ADD_FULL = 2, NPHASE = 6, NROLE = 6 };
#define NOP {-1, {{0, 0}, {0, 0}, {0, 0}}, {{0, 0}, {0, 0}, {0, 0}}}
#define T1(s, c) {{s, c}, {0, 0}, {0, 0}}
#define T2(s, c, s2, c2) {{s, c}, {s2, c2}, {0, 0}}
#define T3(s, c, s2, c2, s3, c3) {{s, c}, {s2, c2}, {s3, c3}}
#define NONE {{0, 0}, {0, 0}, {0, 0}}
#define NOPS6 {NOP, NOP, NOP, NOP, NOP, NOP}
// Product phases take single-slot operands (optionally scaled by a small integer); every other operand is
// materialised by a preceding linear phase, so that no lane recomputes a combination another lane also needs.
// (the tables exist twice: in constant memory for the kernels, and as constexpr twins from which the linear phases take their
// compile-time shape -- how many terms a phase uses and which of them only carry coefficients 0, +1, -1)
#define CS_EC_PROG { \
    /* OP_DOUBLE: X3 = 2XY(Y^2-2XZ-3bZ^2) - 2YZ(X^2+6bXZ-Z^2), ... (complete doubling, ecc.rs:177-242) */ \
    {{{6, T1(0, 1), T1(0, 1)}, {7, T1(1, 1), T1(1, 1)}, {8, T1(2, 1), T1(2, 1)}, {9, T1(0, 1), T1(1, 1)}, {10, T1(0, 1), T1(2, 1)}, {11, T1(1, 1), T1(2, 1)}}, \
     {{12, T1(3, 1), T1(8, 1)}, {13, T1(3, 1), T1(10, 2)}, NOP, NOP, NOP, NOP}, \
     {{19, T3(7, 1, 10, -2, 12, -1), NONE}, {20, T3(7, 1, 10, 2, 12, 1), NONE}, {21, T3(6, 1, 8, -1, 13, 1), NONE}, {22, T2(6, 3, 8, 1), NONE}, NOP, NOP}, \
     {{14, T1(19, 1), T1(20, 1)}, {15, T1(9, 2), T1(19, 1)}, {16, T1(22, 1), T1(21, 1)}, {17, T1(11, 2), T1(21, 1)}, {18, T1(11, 2), T1(7, 1)}, NOP}, \
     {{0, T2(15, 1, 17, -1), NONE}, {1, T2(14, 1, 16, 1), NONE}, {2, T1(18, 4), NONE}, NOP, NOP, NOP}, \
     NOPS6}, \
    /* OP_ADD_MIXED with the affine point in slots 4,5 (complete mixed addition, ecc.rs:330-404) */ \
    {{{6, T1(0, 1), T1(4, 1)}, {7, T1(1, 1), T1(5, 1)}, {8, T1(26, 1), T1(25, 1)}, {9, T1(4, 1), T1(2, 1)}, {10, T1(5, 1), T1(2, 1)}, {11, T1(2, 1), T1(3, 1)}}, \
     {{19, T2(9, 1, 0, 1), NONE}, NOP, NOP, NOP, NOP, NOP}, \
     {{12, T1(19, 1), T1(3, 1)}, NOP, NOP, NOP, NOP, NOP}, \
     {{20, T3(7, 1, 11, -1, 19, -1), NONE}, {21, T3(7, 1, 11, 1, 19, 1), NONE}, {22, T2(6, 3, 2, 1), NONE}, {23, T3(12, 1, 6, 1, 2, -1), NONE}, \
      {24, T2(10, 1, 1, 1), NONE}, {25, T3(8, 1, 6, -1, 7, -1), NONE}}, \
     {{13, T1(20, 1), T1(21, 1)}, {14, T1(22, 1), T1(23, 1)}, {15, T1(24, 1), T1(23, 1)}, {16, T1(25, 1), T1(20, 1)}, {17, T1(25, 1), T1(22, 1)}, {18, T1(24, 1), T1(21, 1)}}, \
     {{0, T2(16, 1, 15, -1), NONE}, {1, T2(13, 1, 14, 1), NONE}, {2, T2(18, 1, 17, 1), NONE}, NOP, NOP, NOP}}, \
    /* OP_ADD_FULL with the projective point in slots 27..29 (complete addition, ecc.rs:244-328) */ \
    {{{6, T1(0, 1), T1(27, 1)}, {7, T1(1, 1), T1(28, 1)}, {8, T1(2, 1), T1(29, 1)}, {9, T1(20, 1), T1(21, 1)}, {10, T1(22, 1), T1(23, 1)}, {11, T1(24, 1), T1(25, 1)}}, \
     {{19, T3(10, 1, 6, -1, 8, -1), NONE}, NOP, NOP, NOP, NOP, NOP}, \
     {{12, T1(3, 1), T1(8, 1)}, {13, T1(3, 1), T1(19, 1)}, NOP, NOP, NOP, NOP}, \
     {{20, T3(7, 1, 12, -1, 19, -1), NONE}, {21, T3(7, 1, 12, 1, 19, 1), NONE}, {22, T2(6, 3, 8, 1), NONE}, {23, T3(13, 1, 6, 1, 8, -1), NONE}, \
      {24, T3(9, 1, 6, -1, 7, -1), NONE}, {25, T3(11, 1, 7, -1, 8, -1), NONE}}, \
     {{14, T1(20, 1), T1(21, 1)}, {15, T1(22, 1), T1(23, 1)}, {16, T1(25, 1), T1(23, 1)}, {17, T1(24, 1), T1(20, 1)}, {18, T1(24, 1), T1(22, 1)}, {26, T1(25, 1), T1(21, 1)}}, \
     {{0, T2(17, 1, 16, -1), NONE}, {1, T2(14, 1, 15, 1), NONE}, {2, T2(26, 1, 18, 1), NONE}, NOP, NOP, NOP}}, \
}
// operand sums that feed the first product phase of the additions (a "phase -1" run before the program):
//   mixed: 26 = x2 + y2 (constant per ladder, set once), 25 = X + Y;  full: 20..25 = X1+Y1, X2+Y2, X1+Z1, X2+Z2, Y1+Z1, Y2+Z2
#define CS_EC_PRE { \
    NOPS6, \
    {{25, T2(0, 1, 1, 1), NONE}, NOP, NOP, NOP, NOP, NOP}, \
    {{20, T2(0, 1, 1, 1), NONE}, {21, T2(27, 1, 28, 1), NONE}, {22, T2(0, 1, 2, 1), NONE}, {23, T2(27, 1, 29, 1), NONE}, {24, T2(1, 1, 2, 1), NONE}, {25, T2(28, 1, 29, 1), NONE}}, \
}
__constant__ Instr c_prog[3][NPHASE][NROLE] = CS_EC_PROG;
__constant__ Instr c_pre[3][NROLE] = CS_EC_PRE;
constexpr Instr k_prog[3][NPHASE][NROLE] = CS_EC_PROG;
constexpr Instr k_pre[3][NROLE] = CS_EC_PRE;
#undef CS_EC_PROG
#undef CS_EC_PRE
// shape of a linear phase: (number of leading terms any instruction uses) | (bit 4 + k set: term k only has coefficients 0, 1, -1)
constexpr int lin_shape(const Instr (&ph)[NROLE]) {
    int nt = 0, unit = 7;
    for (int m = 0; m < NROLE; m++) {
        if (ph[m].out < 0) continue;
        for (int k = 0; k < 3; k++) {
            const int c = ph[m].a[k].coef;
            if (c != 0 && k + 1 > nt) nt = k + 1;
            if (c < -1 || c > 1) unit &= ~(1 << k);
        }
    }
    return nt | (unit << 4);
}
#undef NOP
#undef T1
#undef T2
#undef T3
#undef NONE
#undef NOPS6

// phase kinds of the three programs, known at compile time: 1 product phase, 0 linear phase, -1 end
constexpr int8_t PHASE_KIND[3][NPHASE] = {{1, 1, 0, 1, 0, -1}, {1, 0, 1, 0, 1, 0}, {1, 0, 1, 0, 1, 0}};
// the programs live in LDS while a ladder runs: every lane fetches its own instruction each phase, and a divergent read
// of __constant__ memory is a vector load with global-memory latency.  small[c + 4] = the field element c for |c| <= 4:
// integer coefficients are applied as one multiplication, the same code for every lane (a switch over the coefficient
// would run all of its cases serially in a wave whose lanes hold different instructions).
struct ProgLds { Instr prog[3][NPHASE][NROLE]; Instr pre[3][NROLE]; fp small[9]; };
__device__ __forceinline__ void load_programs(ProgLds &p, int tid, int nthreads) {
    Instr *dst = &p.prog[0][0][0];
    const Instr *src = &c_prog[0][0][0];
    for (int i = tid; i < 3 * NPHASE * NROLE; i += nthreads) dst[i] = src[i];
    for (int i = tid; i < 3 * NROLE; i += nthreads) (&p.pre[0][0])[i] = (&c_pre[0][0])[i];
    if (tid < 9) {
        const int c = tid - 4;
        const fp m = fp_mul((uint64_t)(c < 0 ? -c : c), FP_R2); // |c| in Montgomery form
        p.small[tid] = c < 0 ? fp_neg(m) : m;
    }
}

// F_p2 component `comp` (0..2) of an integer combination of slots.  SHAPE (lin_shape): only the terms the phase uses are walked, and a
// term whose coefficients are all 0 / +1 / -1 is a conditional negation instead of a product (a product by a small constant costs 52
// issue cycles per component, the negation and selects about 20); other coefficients are applied as one multiplication, the same
// code for every lane.
template <int SHAPE>
__device__ __forceinline__ Fp2 lincomb2(const fp (*slot)[6], const Term (&t)[3], int comp, const fp *small) {
    Fp2 r = {0, 0};
#pragma unroll
    for (int k = 0; k < (SHAPE & 15); k++) { // unused terms have coefficient 0 (and slot 0): they add 0
        const fp *s = slot[t[k].slot];
        const int cf = t[k].coef;
        fp va, vb;
        if ((SHAPE >> (4 + k)) & 1) {
            va = s[2 * comp]; vb = s[2 * comp + 1];
            if (cf < 0) { va = fp_neg(va); vb = fp_neg(vb); }
            if (cf == 0) { va = 0; vb = 0; }
        } else {
            const fp c = small[cf + 4];
            va = fp_mul(s[2 * comp], c); vb = fp_mul(s[2 * comp + 1], c);
        }
        r = fp2_add(r, {va, vb});
    }
    return r;
}

// single-slot product operand: component `comp` (+ comp2) of slot t.slot scaled by t.coef (1 or 2)
__device__ __forceinline__ Fp2 operand2(const fp (*slot)[6], const Term t, int comp, int comp2) {
    const fp *s = slot[t.slot];
    Fp2 v = {s[2 * comp], s[2 * comp + 1]};
    if (comp2 >= 0) v = fp2_add(v, {s[2 * comp2], s[2 * comp2 + 1]});
    if (t.coef == 2) v = fp2_dbl(v);
    return v;
}

// Runs program `op` on the slots of one point with the 64 lanes of one wave:
//   product phase  lane (m, q), q < 6: the q-th Karatsuba F_p2 product of instruction m  (3 base products per lane)
//   recombination  lane (m, r), r < 3: F_p2 coefficient r of the F_p6 result of instruction m
//   linear phase   lane (m, r): coefficient r of an integer combination of slots
// Called by all lanes of the workgroup (barriers inside).  prod = this point's [6][6] F_p2 scratch.
template <int SHAPE>
__device__ __forceinline__ void run_linear_phase(const Instr (&prog)[NROLE], const fp *small, fp (*slot)[6], int lane, bool enabled) {
    const int m = lane / 3, r = lane % 3;
    Fp2 c = {0, 0};
    int out = -1;
    if (enabled && m < NROLE) {
        const Instr ins = prog[m];
        out = ins.out;
        if (out >= 0) c = lincomb2<SHAPE>(slot, ins.a, r, small);
    }
    __syncthreads();
    if (out >= 0) { slot[out][2 * r] = c.a; slot[out][2 * r + 1] = c.b; }
    __syncthreads();
}

template <int OP, int PH>
__device__ __forceinline__ void run_phases(const ProgLds &pg, fp (*slot)[6], Fp2 (*prod)[6], int lane, bool enabled) {
    if constexpr (PH < NPHASE) {
        if constexpr (PHASE_KIND[OP][PH] >= 0) {
            if constexpr (PHASE_KIND[OP][PH] == 1) {
                const int m = lane / 6, q = lane % 6;
                if (enabled && m < NROLE) {
                    const Instr ins = pg.prog[OP][PH][m];
                    if (ins.out >= 0) {
                        const int c1 = q < 3 ? q : (q == 5 ? 1 : 0), c2 = q < 3 ? -1 : (q == 3 ? 1 : 2);
                        prod[m][q] = fp2_mul(operand2(slot, ins.a[0], c1, c2), operand2(slot, ins.b[0], c1, c2));
                    }
                }
                __syncthreads();
                const int m2 = lane / 3, r = lane % 3;
                if (enabled && m2 < NROLE) {
                    const int out = pg.prog[OP][PH][m2].out;
                    if (out >= 0) {
                        // r = 0: d0 + d1 + d2 - e12;  r = 1: e01 - e12 - d0;  r = 2: e02 + d1 - d0 - 2 d2.  One five-term form
                        // t1 + t2 + t3 - t4 - t5 with the operands selected per lane: the three cases as branches would run one after
                        // the other in the wave (nine F_p2 additions instead of six).
                        const Fp2 zero = {0, 0};
                        const Fp2 d0 = prod[m2][0], d1 = prod[m2][1], d2 = prod[m2][2];
                        const Fp2 t1 = r == 0 ? d0 : prod[m2][r + 2];        // d0 | e01 | e02
                        const Fp2 t2 = r == 1 ? zero : d1;
                        const Fp2 t3 = r == 0 ? d2 : zero;
                        const Fp2 t4 = r == 2 ? d0 : prod[m2][5];            // e12 | e12 | d0
                        const Fp2 t5 = r == 0 ? zero : (r == 1 ? d0 : fp2_dbl(d2));
                        const Fp2 c = fp2_sub(fp2_sub(fp2_add(fp2_add(t1, t2), t3), t4), t5);
                        slot[out][2 * r] = c.a;
                        slot[out][2 * r + 1] = c.b;
                    }
                }
                __syncthreads();
            } else {
                run_linear_phase<lin_shape(k_prog[OP][PH])>(pg.prog[OP][PH], pg.small, slot, lane, enabled);
            }
            run_phases<OP, PH + 1>(pg, slot, prod, lane, enabled);
        }
    }
}
template <int OP>
__device__ __forceinline__ void run_point_op(const ProgLds &pg, fp (*slot)[6], Fp2 (*prod)[6], int lane, bool enabled) {
    if constexpr (OP != OP_DOUBLE) run_linear_phase<lin_shape(k_pre[OP])>(pg.pre[OP], pg.small, slot, lane, enabled);
    run_phases<OP, 0>(pg, slot, prod, lane, enabled);
}

__device__ __forceinline__ int bit_le(const uint8_t *bytes, int i) { return (bytes[i >> 3] >> (i & 7)) & 1; }

// grid = 2 * n_tx workgroups of ONE wave: workgroup 2t is the ladder s*G of transaction t (registers 0..17), 2t+1 the ladder
// h*P (registers 19..36).  With a single wave per workgroup the phase barriers of run_point_op cost nothing and the two
// ladders of a transaction never wait for each other; 2 * n_tx independent waves keep every SIMD busy with two of them.
// The last step (S += h*P, X <- X/Z) needs both ladders: k_trace_schnorr_final, below.
// TR = rows of the staging tile (rows per flush).  64 is the fastest alone (3.99 ms of trace generation against 5.77 ms with 16);
// 16 leaves room in LDS for a transform workgroup beside the eight resident ladders of a CU and is the fastest when the
// ladders run beside the interpolation / extension of the other registers (prove.hip commit_columns: 43.1 vs 43.8 ms per proof).
template <bool STANDALONE, int TR>
__global__ __launch_bounds__(64) void k_trace_schnorr_ec(TxWitnessDev w, fp *__restrict__ trace, size_t n) {
    __builtin_amdgcn_s_setprio(CS_RECUR_PRIO); // latency-bound recurrence: issue ahead of the chip-filling kernels it runs beside
    constexpr int TG = 64 / TR; // column groups of a flush
    __shared__ fp tile[TR][19];
    __shared__ fp slots[NSLOT][6];
    __shared__ Fp2 prods[6][6];
    __shared__ uint8_t sbytes[32];
    __shared__ ProgLds pg;
    const int t = blockIdx.x >> 1, g = blockIdx.x & 1, lane = threadIdx.x;
    fp(*slot)[6] = slots;
    load_programs(pg, lane, 64);

    // scalars: s from the signature, h from the message hash (little-endian bytes, Lsb0 bit order)
    if (lane < 32) sbytes[lane] = g == 0 ? w.sig_s[32 * (size_t)t + lane] : (uint8_t)(w.h_limbs[4 * (size_t)t + (lane >> 3)] >> (8 * (lane & 7)));
    // constant slots and the initial point (0 : 1 : 0), src/schnorr/trace.rs:22-27
    if (lane < 6) {
        slot[SX][lane] = 0;
        slot[SY][lane] = lane == 0 ? FP_ONE : 0;
        slot[SZ][lane] = 0;
        slot[SB3][lane] = c_b3[lane];
        slot[SPX][lane] = g == 0 ? c_generator[lane] : w.s_old[14 * (size_t)t + lane];
        slot[SPY][lane] = g == 0 ? c_generator[6 + lane] : w.s_old[14 * (size_t)t + 6 + lane];
        slot[26][lane] = fp_add(slot[SPX][lane], slot[SPY][lane]); // x2 + y2, operand of the mixed addition
    }
    __syncthreads();
    if (lane < 18) tile[0][lane] = slot[lane / 6][lane % 6];
    const size_t gbase = STANDALONE ? (size_t)t * MERKLE_LEN : (size_t)t * TXC + MERKLE_LEN;

    for (int step = 0; step < MERKLE_LEN - 1; step++) {
        if (step < SCALAR_MUL_LEN) {
            if ((step & 1) == 0) {
                run_point_op<OP_DOUBLE>(pg, slot, prods, lane, true);
            } else {
                const int bit = bit_le(sbytes, 254 - (step >> 1)); // MSB first, src/schnorr/trace.rs:79-82
                if (bit) run_point_op<OP_ADD_MIXED>(pg, slot, prods, lane, true); // uniform over the wave
            }
        } // step == SCALAR_MUL_LEN: the row is a copy here; k_trace_schnorr_final rewrites registers 0..17 of it
        const int r = (step + 1) & (TR - 1);
        if (lane < 18) tile[r][lane] = slot[lane / 6][lane % 6];
        if (r == TR - 1) {
            __syncthreads();
            for (int c = lane / TR; c < 18; c += TG) trace[(size_t)(g * 19 + c) * n + gbase + (step + 1 - (TR - 1)) + (lane & (TR - 1))] = tile[lane & (TR - 1)][c];
            __syncthreads();
        }
    }
}
// S += h*P, then X <- X / Z  (src/schnorr/trace.rs:105-119): last row of the Schnorr block, registers 0..17.
// One wave per transaction; reads both ladders' results from the row before.
template <bool STANDALONE>
__global__ __launch_bounds__(64) void k_trace_schnorr_final(fp *__restrict__ trace, size_t n) {
    __shared__ fp slots[NSLOT][6];
    __shared__ Fp2 prods[6][6];
    __shared__ ProgLds pg;
    const int t = blockIdx.x, lane = threadIdx.x;
    load_programs(pg, lane, 64);
    const size_t last = (STANDALONE ? (size_t)t * MERKLE_LEN : (size_t)t * TXC + MERKLE_LEN) + MERKLE_LEN - 1;
    if (lane < 18) {
        slots[lane / 6][lane % 6] = trace[(size_t)lane * n + last - 1];
        slots[QX + lane / 6][lane % 6] = trace[(size_t)(19 + lane) * n + last - 1];
    }
    if (lane < 6) slots[SB3][lane] = c_b3[lane];
    __syncthreads();
    run_point_op<OP_ADD_FULL>(pg, slots, prods, lane, true);
    if (lane == 0) fp6_store(slots[SX], fp6_mul(fp6_load(slots[SX]), fp6_inv(fp6_load(slots[SZ]))));
    __syncthreads();
    if (lane < 18) trace[(size_t)lane * n + last] = slots[lane / 6][lane % 6];
}

// ---------------------------------------------------------------------------------------------------
// Closed-form registers.  grid = (n_tx, 4), block = 256: one lane per row.
__device__ __forceinline__ fp small_to_fp(uint64_t x) { return fp_mul(x, FP_R2); } // x < p

// WHICH: 0 = every closed-form register; 1 = those that need the witness only; 2 = those that need the message hash h
// (register 37 and the h-limb accumulators 38..41), which k_trace_schnorr_hash leaves in w.h_limbs.
template <int WHICH>
__global__ __launch_bounds__(256) void k_trace_aux(TxWitnessDev w, fp *__restrict__ trace, size_t n) {
    constexpr bool PLAIN = WHICH != 2, WITH_H = WHICH != 1;
    const int t = blockIdx.x;
    const int r = blockIdx.y * 256 + threadIdx.x; // row inside the transaction
    const size_t g = (size_t)t * TXC + r;
    const fp *sv = w.s_old + 14 * (size_t)t, *rv = w.r_old + 14 * (size_t)t;
    const fp delta = w.deltas[t];
    const fp sigma = fp_sub(sv[12], delta);
    // copies: src/trace.rs:43-53 (constant over the whole transaction)
    if (PLAIN) {
        for (int i = 0; i < 12; i++) {
            trace[(size_t)(65 + i) * n + g] = sv[i];
            trace[(size_t)(77 + i) * n + g] = rv[i];
        }
        trace[(size_t)89 * n + g] = delta;
        trace[(size_t)90 * n + g] = sigma;
        trace[(size_t)91 * n + g] = sv[13];
    }
    if (r < MERKLE_LEN) {
        if (PLAIN) {
            trace[(size_t)92 * n + g] = 0;
            trace[(size_t)93 * n + g] = 0;
        }
        return;
    }
    const int q = r - MERKLE_LEN; // q = 0 is the Schnorr init row; row q is produced by step q-1
    const uint64_t dv = fp_to_u64(delta), sg = fp_to_u64(sigma);
    // range accumulators, src/range/prover.rs:65-84: after k = min(q,64) steps acc = value >> (64-k)
    if (PLAIN) {
        const int k = q < 64 ? q : 64;
        uint64_t dacc = k == 0 ? 0 : dv >> (64 - k), sacc = k == 0 ? 0 : sg >> (64 - k);
        trace[(size_t)56 * n + g] = (dacc & 1) ? FP_ONE : 0;
        trace[(size_t)57 * n + g] = small_to_fp(dacc);
        trace[(size_t)92 * n + g] = (sacc & 1) ? FP_ONE : 0;
        trace[(size_t)93 * n + g] = small_to_fp(sacc);
    }
    // scalar bit registers 18 / 37, src/schnorr/trace.rs:79-82 and :110
    const uint8_t *sb = w.sig_s + 32 * (size_t)t;
    const uint64_t *h = w.h_limbs + 4 * (size_t)t;
    fp bs = 0, bh = 0;
    if (q >= 1) {
        const int step = q - 1 < SCALAR_MUL_LEN - 1 ? q - 1 : SCALAR_MUL_LEN - 1;
        const int bi = 254 - (step >> 1);
        bs = bit_le(sb, bi) ? FP_ONE : 0;
        bh = ((h[bi >> 6] >> (bi & 63)) & 1) ? FP_ONE : 0;
        if (q == SCALAR_MUL_LEN + 1) bs = FP_ONE;
    }
    if (PLAIN) trace[(size_t)18 * n + g] = bs;
    if (!WITH_H) return;
    trace[(size_t)37 * n + g] = bh;
    // h-limb accumulators 38..41: k bits consumed MSB-first; chunk 0 = 63 bits into reg 41, then 64-bit chunks
    const int k = (q + 1) / 2 < 255 ? (q + 1) / 2 : 255;
    for (int c = 0; c < 4; c++) {
        const int width = c == 0 ? 63 : 64;
        const int start = c == 0 ? 0 : 63 + 64 * (c - 1);
        int kc = k - start;
        kc = kc < 0 ? 0 : kc > width ? width : kc;
        const uint64_t limb = h[3 - c];
        const uint64_t acc = kc == 0 ? 0 : (c == 0 ? (limb >> (63 - kc)) : (kc == 64 ? limb : limb >> (64 - kc)));
        trace[(size_t)(41 - c) * n + g] = small_to_fp(acc);
    }
}

} // namespace

// Unused dynamic LDS added to every ladder workgroup of the overlapped path (tuning: CSTARK_EC_LDS_PAD); measured best at 0.
static unsigned ladder_lds_pad() {
    static const unsigned pad = [] { const char *e = getenv("CSTARK_EC_LDS_PAD"); return e ? (unsigned)atoi(e) : LADDER_LDS_PAD; }();
    return pad;
}
hipError_t launch_trace_gen(const TxWitnessDev &w, fp *d_trace, hipStream_t stream, hipStream_t side, hipEvent_t fork, hipEvent_t join) {
    const size_t n = (size_t)w.n_tx * TXC;
    // the Merkle-phase recurrence is independent of the Schnorr half: run it beside the curve ladders (both are
    // latency-bound, one wave per transaction) on a side stream forked from / joined back into `stream`
    hipError_t e;
    if ((e = hipEventRecord(fork, stream)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(side, fork, 0)) != hipSuccess) return e;
    hipLaunchKernelGGL((k_trace_merkle<false, 64>), dim3(w.n_tx), dim3(64), 0, side, w, d_trace, n);
    if ((e = hipEventRecord(join, side)) != hipSuccess) return e;
    hipLaunchKernelGGL((k_trace_schnorr_hash<false, 64>), dim3(w.n_tx), dim3(64), 0, stream, w, d_trace, n);
    hipLaunchKernelGGL((k_trace_schnorr_ec<false, 64>), dim3(2 * w.n_tx), dim3(64), 0, stream, w, d_trace, n);
    hipLaunchKernelGGL(k_trace_schnorr_final<false>, dim3(w.n_tx), dim3(64), 0, stream, d_trace, n);
    hipLaunchKernelGGL(k_trace_aux<0>, dim3(w.n_tx, 4), dim3(256), 0, stream, w, d_trace, n);
    if ((e = hipStreamWaitEvent(stream, join, 0)) != hipSuccess) return e;
    return hipGetLastError();
}

hipError_t launch_trace_gen_split(const TxWitnessDev &w, fp *d_trace, hipStream_t stream, hipStream_t side_a, hipStream_t side_b, hipEvent_t fork,
                                  hipEvent_t join_a, hipEvent_t mid_b, hipEvent_t join_b) {
    const size_t n = (size_t)w.n_tx * TXC;
    hipError_t e;
    if ((e = hipEventRecord(fork, stream)) != hipSuccess) return e;
    // the closed-form registers first: the caller's stream transforms them next, and that chain of transforms is what a proof waits for
    // (launched after the recurrences it started ~45 us later)
    hipLaunchKernelGGL(k_trace_aux<1>, dim3(w.n_tx, 4), dim3(256), 0, stream, w, d_trace, n);
    if ((e = hipStreamWaitEvent(side_a, fork, 0)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(side_b, fork, 0)) != hipSuccess) return e;
    hipLaunchKernelGGL((k_trace_merkle<false, CS_RECUR_TILE_ROWS>), dim3(w.n_tx), dim3(64), 0, side_a, w, d_trace, n);
    if ((e = hipEventRecord(join_a, side_a)) != hipSuccess) return e;
    // the message hash produces the scalar h of the second ladder (w.h_limbs): the ladders run behind it
    hipLaunchKernelGGL((k_trace_schnorr_hash<false, CS_RECUR_TILE_ROWS>), dim3(w.n_tx), dim3(64), 0, side_b, w, d_trace, n);
    hipLaunchKernelGGL(k_trace_aux<2>, dim3(w.n_tx, 4), dim3(256), 0, side_b, w, d_trace, n);
    if ((e = hipEventRecord(mid_b, side_b)) != hipSuccess) return e;
    static const int ec_rows = [] { const char *e = getenv("CSTARK_EC_TILE_ROWS"); return e ? atoi(e) : 16; }(); // tuning: rows per flush of the ladders
    if (ec_rows == 64) hipLaunchKernelGGL((k_trace_schnorr_ec<false, 64>), dim3(2 * w.n_tx), dim3(64), ladder_lds_pad(), side_b, w, d_trace, n);
    else if (ec_rows == 32) hipLaunchKernelGGL((k_trace_schnorr_ec<false, 32>), dim3(2 * w.n_tx), dim3(64), ladder_lds_pad(), side_b, w, d_trace, n);
    else hipLaunchKernelGGL((k_trace_schnorr_ec<false, 16>), dim3(2 * w.n_tx), dim3(64), ladder_lds_pad(), side_b, w, d_trace, n);
    hipLaunchKernelGGL(k_trace_schnorr_final<false>, dim3(w.n_tx), dim3(64), 0, side_b, d_trace, n);
    if ((e = hipEventRecord(join_b, side_b)) != hipSuccess) return e;
    return hipGetLastError();
}

// standalone SchnorrAir: bit registers 18 / 37 and h-limb accumulators 38..41 (closed forms, as in k_trace_aux); grid (n, 2) x 256
__global__ __launch_bounds__(256) void k_trace_schnorr_bits(TxWitnessDev w, fp *__restrict__ trace, size_t n) {
    const int t = blockIdx.x;
    const int q = blockIdx.y * 256 + threadIdx.x; // row inside the signature block; row q is produced by step q-1
    const size_t g = (size_t)t * MERKLE_LEN + q;
    const uint8_t *sb = w.sig_s + 32 * (size_t)t;
    const uint64_t *h = w.h_limbs + 4 * (size_t)t;
    fp bs = 0, bh = 0;
    if (q >= 1) {
        const int step = q - 1 < SCALAR_MUL_LEN - 1 ? q - 1 : SCALAR_MUL_LEN - 1;
        const int bi = 254 - (step >> 1);
        bs = bit_le(sb, bi) ? FP_ONE : 0;
        bh = ((h[bi >> 6] >> (bi & 63)) & 1) ? FP_ONE : 0;
        if (q == SCALAR_MUL_LEN + 1) bs = FP_ONE;
    }
    trace[(size_t)18 * n + g] = bs;
    trace[(size_t)37 * n + g] = bh;
    const int k = (q + 1) / 2 < 255 ? (q + 1) / 2 : 255;
    for (int c = 0; c < 4; c++) {
        const int width = c == 0 ? 63 : 64, start = c == 0 ? 0 : 63 + 64 * (c - 1);
        int kc = k - start;
        kc = kc < 0 ? 0 : kc > width ? width : kc;
        const uint64_t limb = h[3 - c];
        const uint64_t acc = kc == 0 ? 0 : (c == 0 ? (limb >> (63 - kc)) : (kc == 64 ? limb : limb >> (64 - kc)));
        trace[(size_t)(41 - c) * n + g] = small_to_fp(acc);
    }
}
__global__ __launch_bounds__(256) void k_schnorr_aux_columns(TxWitnessDev w, fp *__restrict__ out, size_t n) {
    const int t = blockIdx.x;
    const int q = blockIdx.y * 256 + threadIdx.x;
    const size_t g = (size_t)t * MERKLE_LEN + q;
    const fp *sv = w.s_old + 14 * (size_t)t, *rv = w.r_old + 14 * (size_t)t;
    for (int j = 0; j < 12; j++) out[(size_t)j * n + g] = sv[j];
    const bool chunk_row = q < 32 && (q & 7) == 7;
    for (int j = 0; j < 7; j++) {
        fp v = 0;
        if (chunk_row) {
            const int m = j + 7 * (q >> 3);
            v = m < 12 ? sv[m] : m < 24 ? rv[m - 12] : m == 24 ? w.deltas[t] : m == 25 ? sv[13] : w.msg_tail[2 * (size_t)t + m - 26];
        }
        out[(size_t)(12 + j) * n + g] = v;
    }
}
hipError_t launch_schnorr_trace(const TxWitnessDev &w, fp *d_trace, hipStream_t stream) {
    const size_t n = (size_t)w.n_tx * MERKLE_LEN;
    hipLaunchKernelGGL((k_trace_schnorr_hash<true, 64>), dim3(w.n_tx), dim3(64), 0, stream, w, d_trace, n);
    hipLaunchKernelGGL((k_trace_schnorr_ec<true, 64>), dim3(2 * w.n_tx), dim3(64), 0, stream, w, d_trace, n);
    hipLaunchKernelGGL(k_trace_schnorr_final<true>, dim3(w.n_tx), dim3(64), 0, stream, d_trace, n);
    hipLaunchKernelGGL(k_trace_schnorr_bits, dim3(w.n_tx, 2), dim3(256), 0, stream, w, d_trace, n);
    return hipGetLastError();
}
// The same kernels on an internal stream, so that the caller's stream can do work that does not depend on the trace (the public-input
// columns of SchnorrAir, then the interpolation and extension of registers 37..55) beside the latency-bound ladders (one wave per SIMD
// at 512 signatures).  join_a: registers 37..55 are complete (message hash, bit registers, limb accumulators); join_b: all of them.
// The ladders' staging tile: 64 rows while four ladders per CU leave room in LDS for a transform workgroup (13.5 KB each), else 16.
hipError_t launch_schnorr_trace_split(const TxWitnessDev &w, fp *d_trace, hipStream_t stream, hipStream_t side, hipEvent_t fork, hipEvent_t join_a,
                                      hipEvent_t join_b) {
    const size_t n = (size_t)w.n_tx * MERKLE_LEN;
    hipError_t e;
    if ((e = hipEventRecord(fork, stream)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(side, fork, 0)) != hipSuccess) return e;
    hipLaunchKernelGGL((k_trace_schnorr_hash<true, 64>), dim3(w.n_tx), dim3(64), 0, side, w, d_trace, n);
    hipLaunchKernelGGL(k_trace_schnorr_bits, dim3(w.n_tx, 2), dim3(256), 0, side, w, d_trace, n);
    if ((e = hipEventRecord(join_a, side)) != hipSuccess) return e;
    if (w.n_tx <= 512) hipLaunchKernelGGL((k_trace_schnorr_ec<true, 64>), dim3(2 * w.n_tx), dim3(64), 0, side, w, d_trace, n);
    else hipLaunchKernelGGL((k_trace_schnorr_ec<true, 16>), dim3(2 * w.n_tx), dim3(64), 0, side, w, d_trace, n);
    hipLaunchKernelGGL(k_trace_schnorr_final<true>, dim3(w.n_tx), dim3(64), 0, side, d_trace, n);
    if ((e = hipEventRecord(join_b, side)) != hipSuccess) return e;
    return hipGetLastError();
}
hipError_t launch_schnorr_aux_columns(const TxWitnessDev &w, fp *d_out, hipStream_t stream) {
    const size_t n = (size_t)w.n_tx * MERKLE_LEN;
    hipLaunchKernelGGL(k_schnorr_aux_columns, dim3(w.n_tx, 2), dim3(256), 0, stream, w, d_out, n);
    return hipGetLastError();
}

// RescueProver::build_trace (benches/rescue.rs:277-322): ONE chain -- 8 rows per link: seven Rescue rounds, then the capacity half reset
// to zero -- is a single sequential recurrence: one wave, lane e < 14 = state element e, 64 rows staged in LDS per coalesced flush.
// n = 8 * iterations rows, a multiple of 64.
struct RescueSeed { fp s[7]; };
__global__ __launch_bounds__(64) void k_trace_rescue_chain(RescueSeed seed, fp *__restrict__ trace, size_t n) {
    __shared__ fp tile[64][15];
    __shared__ fp xch[14];
    const int lane = threadIdx.x, e = lane < 14 ? lane : 0;
    const bool active = lane < 14;
    fp mrow[14];
#pragma unroll
    for (int j = 0; j < 14; j++) mrow[j] = c_mds[e * 14 + j];
    fp v = (active && lane < 7) ? seed.s[lane] : 0;
    if (active) tile[0][e] = v;
    for (size_t step = 0; step + 1 < n; step++) {
        const int cyc = (int)(step & 7);
        if (cyc < 7) v = rescue_round_lane(v, xch, mrow, e, cyc, active); // rescue::apply_round(state, step)
        else if (lane >= 7) v = 0;
        const int r = (int)((step + 1) & 63);
        if (active) tile[r][e] = v;
        if (r == 63) {
            __syncthreads();
            for (int c = 0; c < 14; c++) trace[(size_t)c * n + (step + 1 - 63) + lane] = tile[lane][c];
            __syncthreads();
        }
    }
}
hipError_t launch_rescue_chain_trace(const uint64_t seed[7], unsigned iterations, fp *d_trace, hipStream_t stream) {
    RescueSeed s;
    for (int i = 0; i < 7; i++) s.s[i] = seed[i];
    hipLaunchKernelGGL(k_trace_rescue_chain, dim3(1), dim3(64), 0, stream, s, d_trace, (size_t)iterations * 8);
    return hipGetLastError();
}

hipError_t launch_merkle_trace(const TxWitnessDev &w, fp *d_trace, hipStream_t stream) {
    const size_t n = (size_t)w.n_tx * MERKLE_LEN;
    hipLaunchKernelGGL((k_trace_merkle<true, 64>), dim3(w.n_tx), dim3(64), 0, stream, w, d_trace, n);
    return hipGetLastError();
}

// RangeProver::build_trace (src/range/prover.rs:24-43): row q holds bit (62 - (q-1)) and the top q bits of the 63-bit value
__global__ void k_trace_range(uint64_t number, fp *trace) {
    const int q = threadIdx.x; // 64 rows
    const uint64_t v63 = number & 0x7FFFFFFFFFFFFFFFULL; // bits 62..0 are consumed, MSB first
    const uint64_t acc = q == 0 ? 0 : v63 >> (63 - q);
    trace[q] = (q >= 1 && (acc & 1)) ? FP_ONE : 0;
    trace[64 + q] = fp_from_u64(acc);
}
hipError_t launch_range_trace(uint64_t number_canonical, fp *d_trace, hipStream_t stream) {
    hipLaunchKernelGGL(k_trace_range, dim3(1), dim3(64), 0, stream, number_canonical, d_trace);
    return hipGetLastError();
}

// The same accumulator over n = 2^log_n rows (synthetic long form of BASELINE.json's "range, 2^16 steps"): V is an (n-1)-bit integer,
// words[] its n/64 little-endian words; row q >= 1 holds bit (n-1-q) of V and acc_q = (V >> (n-1-q)) mod p.  With the MSB-first
// 64-bit chunks U_c = words[n/64 - 1 - c] the accumulator at the end of chunk c is E_c = sum_{c' <= c} U_c' (2^64)^(c-c') and inside
// the chunk acc_(64c+r) = E_(c-1) 2^(r+1) + (U_c >> (63-r)).
// Pass 1 (one workgroup): E_c for every chunk -- per-thread Horner over a segment, a scan of the 256 segment values, a second
// Horner pass.  Pass 2: one thread per row, coalesced stores.
__global__ __launch_bounds__(256) void k_range_chunk_prefix(const uint64_t *__restrict__ words, fp *__restrict__ prefix, unsigned n_chunks) {
    __shared__ fp seg[256], lead[256];
    const unsigned t = threadIdx.x, per = (n_chunks + 255) / 256, c0 = t * per, c1 = min(c0 + per, n_chunks);
    const fp B = FP_R2; // 2^64 in memory form is 2^128 mod p
    fp v = 0;
    for (unsigned c = c0; c < c1; c++) v = fp_add(fp_mul(v, B), fp_from_u64(words[n_chunks - 1 - c]));
    seg[t] = v;
    __syncthreads();
    if (t == 0) { // value carried into every segment: 256 dependent products, negligible
        const fp Bper = fp_pow(B, per);
        fp carry = 0;
        for (unsigned i = 0; i < 256; i++) { lead[i] = carry; carry = fp_add(fp_mul(carry, Bper), seg[i]); }
    }
    __syncthreads();
    v = lead[t];
    for (unsigned c = c0; c < c1; c++) { v = fp_add(fp_mul(v, B), fp_from_u64(words[n_chunks - 1 - c])); prefix[c] = v; }
}
__global__ __launch_bounds__(256) void k_trace_range_bits(const uint64_t *__restrict__ words, const fp *__restrict__ prefix, fp *__restrict__ trace,
                                                          unsigned log_n) {
    const size_t n = (size_t)1 << log_n, q = blockIdx.x * (size_t)256 + threadIdx.x;
    if (q >= n) return;
    const unsigned c = (unsigned)(q >> 6), r = (unsigned)(q & 63), n_chunks = (unsigned)(n >> 6);
    const uint64_t u = words[n_chunks - 1 - c];
    const fp before = c ? prefix[c - 1] : 0;
    const fp two_r1 = fp_pow(fp_from_u64(2), r + 1);
    trace[q] = ((u >> (63 - r)) & 1) ? FP_ONE : 0;
    trace[n + q] = fp_add(fp_mul(before, two_r1), fp_from_u64(u >> (63 - r)));
}
hipError_t launch_range_trace_bits(const uint64_t *d_words, fp *d_prefix, fp *d_trace, unsigned log_n, hipStream_t stream) {
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(k_range_chunk_prefix, dim3(1), dim3(256), 0, stream, d_words, d_prefix, (unsigned)(n >> 6));
    hipLaunchKernelGGL(k_trace_range_bits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_words, (const fp *)d_prefix, d_trace, log_n);
    return hipGetLastError();
}

} // namespace cs
