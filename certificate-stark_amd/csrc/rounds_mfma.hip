// TransactionAir's five Rescue windows (split evaluation; one to three coefficient sets) and the folded round gadgets of MerkleAir /
// SchnorrAir with EVERY constant-matrix product on the matrix cores:
//   y = INV_MDS (next - ark2)                                    14 x 14, constant                     (src/utils/rescue.rs:345-375)
//   v_sec = gamma_sec . cube(y) - U_sec . cube(cur),  U = MDS^T gamma   up to 8 sections x 28, per proof   (rescue.rs:269-300 folded)
// Same values as k_rounds_split / k_merkle_rounds (constraints.hip), bit for bit: the 64-bit integer products are byte-decomposed exactly as in
// mds_mfma.cuh (signed base-256 digits, int8 GEMM of the byte diagonals, carry-free recombination, one Montgomery reduction).
//
// What k_rounds_split spends per point: 980 + 322 limb multiply-adds of six v_mad_u64_u32 each (the inverse matrix and the forward
// dot products), 322 carry-chained 128-bit multiply-adds (the sections' sums over the cubes), 280 field products for the cubes.  Here
// only the cubes, one recombination per output and the flag products are left on the vector ALU; the matrix pipe runs beside them.
//
// Layout of a wave (64 consecutive points): lane (n, h) = (lane & 31, lane >> 5) works for the TWO points n and n + 32.
//   * v_mfma_i32_32x32x32_i8 takes 16 bytes of the contraction index per lane: half h of a k-step = two field elements.  The lane
//     reads its two elements of both points straight from the window image in LDS (no shuffles): columns 4 s + 2 h, + 1.
//   * A 32-row tile = two outputs; all 15 byte diagonals of output g of point n arrive in lane (n, g).  That lane cubes the output and
//     keeps it: it IS the lane's half of the next product's operand (contraction order of the section tables: m -> i = 2 m + h), so
//     nothing moves between the two GEMMs either.  The forward cubes cube(cur_j), j = 2 t + h, are formed by the same lane.
//   * A section's value of point n lands in lane (n, g) for section slot 2 u + g; every lane accumulates its sections' terms for its
//     two points and the two halves are added once at the end (one cross-half exchange of four values per point).
// Tables: compact (one word of eight signed digits per matrix entry, rounds_layout.h); the Toeplitz fragments of the inverse matrix
// are expanded into LDS once per workgroup (28 KB), those of the sections per use (two v_perm_b32 per word).
// The 28-term section sums would exceed 2^128 with operands in [0, p): both factors are centred (coefficients in (-p/2, p/2) as
// signed digits, operands x - (p-1)/2), the constant (X0 + (p-1)/2) sum(c) mod p rides in the row constant with the 2p 2^64 that
// keeps the value positive: |sum| < 7.1 p^2 around 2p 2^64, inside [0, 2^128).
#include "rounds_layout.h"
#include "rescue.cuh"

namespace cs {
namespace {

using mdsmfma::v16i;
using mdsmfma::v4i;
typedef unsigned __int128 u128;

#ifndef RM_NT_
#define RM_NT_ 256
#endif
#ifndef RM_WAVES
#define RM_WAVES 2
#endif
constexpr int RM_NT = RM_NT_;                    // threads per workgroup (one coefficient set): its waves share one expanded inverse table
// window image of a wave of 32 PTS points: 14 columns x rows j0 .. j0 + 32 PTS + 1 (PTS = 2: k_rounds_split's)
template <int PTS> constexpr int rw_rows = 32 * PTS + 2;
template <int PTS> constexpr int rw_img = 14 * rw_rows<PTS>;
constexpr uint64_t X0 = mdsmfma::X0;
constexpr uint64_t HALF_P = (FP_P - 1) / 2;
enum { P_SETUP = 0, P_HASH = 4, P_SCHNORR_HASH = 12, P_ARK = 20 }; // periodic columns (constraints.hip)

__constant__ RoundWindow c_windows[5] = CS_ROUND_WINDOWS_INIT;
__constant__ int c_window_groups[5][2][3] = CS_WINDOW_GROUPS_INIT;

// signed base-256 digits of a two's-complement value |v| < 2^62: v = sum_a dig_a 2^(8a), dig_a in [-128, 127]; digit a at byte a
__device__ inline uint64_t digit_word(int64_t v) {
    uint64_t w = 0;
    int carry = 0;
    for (int a = 0; a < 8; a++) {
        int b = (a < 7 ? (int)(((uint64_t)v >> (8 * a)) & 0xff) : (int)(v >> 56)) + carry; // the top byte is signed as it stands
        carry = (a < 7 && b >= 128) ? 1 : 0;
        w |= (uint64_t)(uint8_t)(int8_t)(b - 256 * carry) << (8 * a);
    }
    return w;
}
__device__ __forceinline__ u128 acc0_offset() { // sum_d ACC0 2^(8d): what the accumulator start values add to the recombined sum
    u128 off = 0;
    for (int d = 0; d < 15; d++) off += (u128)mdsmfma::ACC0 << (8 * d);
    return off;
}

// inverse matrix, tile T: compact entries d_out[g][h][m] (m < 8) and the row constants k_out[2 T + g][4]; threads t < 34
__device__ inline void build_inverse_tile(int T, int t, uint64_t *d_out, uint64_t *k_out) {
    if (t < 32) {
        const int g = t >> 4, h = (t >> 3) & 1, m = t & 7, i = 2 * T + g, j = 4 * (m >> 1) + 2 * h + (m & 1);
        d_out[(g * 2 + h) * 8 + m] = j < 14 ? digit_word((int64_t)c_inv_mds[i * 14 + j]) : 0;
    } else if (t < 34) { // row constant of output i: X0 sum_j M_ij - offset (mod 2^128), the sum itself is below 14 p^2
        const int i = 2 * T + (t - 32);
        u128 k = 0;
        for (int j = 0; j < 14; j++) k += (u128)c_inv_mds[i * 14 + j] * X0;
        k -= acc0_offset();
        for (int q = 0; q < 4; q++) k_out[4 * (t - 32) + q] = (uint32_t)(k >> (32 * q));
    }
}
// one section tile: outputs g = 0, 1 are the sections with coefficient vectors gam[g][14] and MDS-folded limbs ul[g] (4 dwords per
// entry, k_rounds_setup); compact entries d_out[g][h][m] (m < 16) and row constants k_out[g][4].  64 threads, all of them call.
__device__ inline void build_section_tile(const fp *gam0, const fp *gam1, const uint32_t *ul0, const uint32_t *ul1, int t, uint64_t *d_out,
                                          uint64_t *k_out, fp (*csum)[32]) {
    if (t < 56) { // [g][h][m], m < 14
        const int g = t / 28, h = (t / 14) & 1, m = t % 14;
        fp c;
        if (m < 7) c = (g ? gam1 : gam0)[2 * m + h];
        else {
            const uint32_t *l = (g ? ul1 : ul0) + (2 * (m - 7) + h) * 4;
            c = fp_neg((fp)l[0] | ((fp)l[1] << 21) | ((fp)l[2] << 42));
        }
        d_out[(g * 2 + h) * 16 + m] = digit_word(c > HALF_P ? (int64_t)(c - FP_P) : (int64_t)c);
        csum[g][h * 14 + m] = c;
    } else { // m = 14, 15 of every (g, h): padding of the last k-step's rows
        const int q = t - 56;
        d_out[(q >> 1) * 16 + 14 + (q & 1)] = 0;
    }
    __syncthreads();
    if (t < 2) { // row constant: (X0 + (p-1)/2) sum(c) mod p  -  offset  +  2p 2^64
        fp sum = 0;
        for (int e = 0; e < 28; e++) sum = fp_add(sum, csum[t][e]);
        const fp kc = fp_mul(fp_mul(X0 + HALF_P, sum), FP_R2); // the plain product of the two integers mod p
        u128 k = (u128)kc + ((u128)(2 * FP_P) << 64);
        k -= acc0_offset();
        for (int q = 0; q < 4; q++) k_out[4 * t + q] = (uint32_t)(k >> (32 * q));
    }
}

// Compact tables of one TransactionAir proof from what k_rounds_setup left in rtab (RT_G: the sections' coefficient vectors; RT_UL:
// limbs of their MDS-folded forms).  grid = 7 inverse tiles + 13 section tiles, 64 threads.
__global__ void k_rounds_mfma_tables(fp *__restrict__ rtab) {
    rtab += (size_t)blockIdx.y * CE_RTAB_WORDS; // grid.y = coefficient set
    const int t = threadIdx.x;
    if (blockIdx.x < MF_TILES_INV) {
        build_inverse_tile(blockIdx.x, t, rtab + MF_INV_D + blockIdx.x * 32, rtab + MF_K + 8 * blockIdx.x);
        return;
    }
    const int tile = blockIdx.x - MF_TILES_INV, w = mf_tile_window(tile), local = tile - mf_tile_base(w);
    const int sec = (w * 2 + mf_tile_fs(w, local)) * 4 + 2 * mf_tile_pair(w, local);
    __shared__ fp csum[2][32];
    const uint32_t *ul = (const uint32_t *)(rtab + RT_UL);
    build_section_tile(rtab + RT_G + sec * 14, rtab + RT_G + (sec + 1) * 14, ul + sec * 56, ul + (sec + 1) * 56, t, rtab + MF_SEC_D + tile * 64,
                       rtab + MF_K + 4 * (14 + 2 * tile), csum);
}
// The same for the folded round gadgets of MerkleAir / SchnorrAir's message hash (k_merkle_rounds_setup's rtab, MR_* of
// rounds_layout.h): one section tile per window (outputs: alpha, beta).  grid = 7 + nwin, 64 threads.
__global__ void k_merkle_rounds_mfma_tables(fp *__restrict__ rtab) {
    const int t = threadIdx.x;
    if (blockIdx.x < MF_TILES_INV) {
        build_inverse_tile(blockIdx.x, t, rtab + MRF_INV_D + blockIdx.x * 32, rtab + MRF_K + 8 * blockIdx.x);
        return;
    }
    const int w = blockIdx.x - MF_TILES_INV;
    __shared__ fp csum[2][32];
    const uint32_t *ul = (const uint32_t *)(rtab + MR_UL);
    build_section_tile(rtab + MR_G + (2 * w) * 14, rtab + MR_G + (2 * w + 1) * 14, ul + (2 * w) * 56, ul + (2 * w + 1) * 56, t, rtab + MRF_SEC_D + w * 64,
                       rtab + MRF_K + 4 * (14 + 2 * w), csum);
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
// LDS-DMA of one window: lane l < 16 PTS + 1 moves rows j0 + 2l, j0 + 2l + 1 of the 14 columns into the wave's image (k_rounds_split)
template <int PTS>
__device__ __forceinline__ void fetch_window(const fp *rows, size_t n, int reg, int lane, fp *img) {
    if (lane < 16 * PTS + 1) {
#pragma unroll
        for (int j = 0; j < 14; j++)
            __builtin_amdgcn_global_load_lds((glb_void *)(rows + (size_t)(reg + j) * n), (lds_void *)(img + j * rw_rows<PTS>), 16, 0, 0);
    }
}
__device__ __forceinline__ v4i pack2(uint64_t a, uint64_t b) {
    v4i r;
    r[0] = (int)(uint32_t)a; r[1] = (int)(uint32_t)(a >> 32); r[2] = (int)(uint32_t)b; r[3] = (int)(uint32_t)(b >> 32);
    return r;
}
// this lane's 16 bytes of a Toeplitz fragment from the digit words of the k-step's two entries: byte b of a word's part = digit d - b
__device__ __forceinline__ v4i expand_frag(uint64_t d0, uint64_t d1, uint32_t sel_lo, uint32_t sel_hi) {
    v4i r;
    r[0] = (int)__builtin_amdgcn_perm((uint32_t)(d0 >> 32), (uint32_t)d0, sel_lo);
    r[1] = (int)__builtin_amdgcn_perm((uint32_t)(d0 >> 32), (uint32_t)d0, sel_hi);
    r[2] = (int)__builtin_amdgcn_perm((uint32_t)(d1 >> 32), (uint32_t)d1, sel_lo);
    r[3] = (int)__builtin_amdgcn_perm((uint32_t)(d1 >> 32), (uint32_t)d1, sel_hi);
    return r;
}
__device__ __forceinline__ v16i acc_start() {
    v16i a;
#pragma unroll
    for (int v = 0; v < 16; v++) a[v] = mdsmfma::ACC0;
    return a;
}
// A constant in a scalar register that the compiler cannot see through: x * c + t stays ONE v_mad_u64_u32 where the compiler's own
// form of a multiplication by 2^8 / 2^16 / 2^24 is a 64-bit shift and a 64-bit add (two instructions of the same issue cost each).
template <uint32_t V>
__device__ __forceinline__ uint32_t opaque_const() {
    uint32_t r;
    asm("s_mov_b32 %0, %1" : "=s"(r) : "n"(V));
    return r;
}
struct Shifts { uint32_t one, s8, s16, s24; };
// The 15 byte diagonals of one output (non-negative, below 2^24) and the row constant -> Montgomery-reduced field element.
// Diagonal d sits at bit 8 d: the four diagonals 4 q .. 4 q + 3 of 32-bit word q are summed with their shifts by three
// v_mad_u64_u32 on top of the constant's word q (a fourth carries diagonal 4 q into the 64-bit sum: multiplication by an opaque
// one), and the four sums S_q < 2^49 overlap by their high words only: three carry additions.  18 instructions, where packing the
// diagonals into three carry-free 128-bit words and adding those (mdsmfma::recombine) takes 29.
__device__ __forceinline__ fp recombine_mad(const v16i &acc, const uint64_t (&k)[4], const Shifts &c) {
#ifdef RM_OLD_RECOMBINE
    return mdsmfma::recombine(acc, k[0] | (k[1] << 32), k[2] | (k[3] << 32));
#endif
    uint64_t s[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint64_t t = mad_u64_u32((uint32_t)acc[4 * q], c.one, k[q]);
        t = mad_u64_u32((uint32_t)acc[4 * q + 1], c.s8, t);
        t = mad_u64_u32((uint32_t)acc[4 * q + 2], c.s16, t);
        if (4 * q + 3 < 15) t = mad_u64_u32((uint32_t)acc[4 * q + 3], c.s24, t);
        s[q] = t;
    }
    unsigned cy;
    const uint32_t w0 = (uint32_t)s[0];
    const uint32_t w1 = __builtin_addc((uint32_t)(s[0] >> 32), (uint32_t)s[1], 0u, &cy);
    const uint32_t w2 = __builtin_addc((uint32_t)(s[1] >> 32), (uint32_t)s[2], cy, &cy);
    const uint32_t w3 = (uint32_t)(s[2] >> 32) + (uint32_t)s[3] + cy;
    Acc128 a{((uint64_t)w1 << 32) | w0, ((uint64_t)w3 << 32) | w2};
    acc_fold(a);
    return acc_reduce(a);
}
#ifndef RM_DBUF
// vector instructions per MFMA of the NEXT tile in the inverse half (two accumulator sets); 0: one tile at a time.  Measured on three
// boxes: 16 gave 1.745 against 1.82 ms on one, 1.94 against 1.87-1.92 on two others (243-256 registers, spills in the generic form);
// 12: 1.91, 20: 1.78.  Inside the noise between boxes: the simple form is the default.
#define RM_DBUF 0
#endif
#ifndef RM_RESIDENT_
#define RM_RESIDENT_ 4096
#endif
constexpr unsigned RM_RESIDENT = RM_RESIDENT_; // workgroups of a launch: each takes every gridDim.x-th block of RM_NT rows of its coset (four blocks at 2^20 rows: 1.83 ms; one: 1.86; all resident at once, 32 blocks each: 1.92)

// what a lane is in the tiles: column n = lane & 31 of the points' operand (points n and n + 32 of the wave), half h = lane >> 5 of a
// k-step; as a table row r = lane & 31: output g = (r >> 2) & 1, byte diagonal d = (r & 3) + 4 (r >> 3)
struct Role {
    int lane, nn, h, ag;
    uint32_t sel_lo, sel_hi; // v_perm_b32 selectors: byte b of the fragment word = digit d - b of the entry (0x0c: constant zero)
};
__device__ __forceinline__ void frag_selectors(int d, uint32_t &lo, uint32_t &hi) {
    lo = hi = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int a0 = d - b, a1 = d - 4 - b;
        lo |= (uint32_t)((a0 >= 0 && a0 <= 7 && d < 15) ? a0 : 0x0c) << (8 * b);
        hi |= (uint32_t)((a1 >= 0 && a1 <= 7 && d < 15) ? a1 : 0x0c) << (8 * b);
    }
}
__device__ __forceinline__ Role make_role(int lane) {
    Role r;
    r.lane = lane; r.nn = lane & 31; r.h = lane >> 5; r.ag = (r.nn >> 2) & 1;
    frag_selectors((r.nn & 3) + 4 * (r.nn >> 3), r.sel_lo, r.sel_hi);
    return r;
}
// the inverse matrix's fragments, expanded once per workgroup: fragment (T, s) of lane l from the entries [T][g][h][2 s], [2 s + 1]
__device__ __forceinline__ void expand_inverse_table(v4i *inv_lds, const uint64_t *inv_d, int tid, int nthreads) {
    for (unsigned e = tid; e < MF_TILES_INV * MF_KS_INV * 64; e += nthreads) {
        const unsigned l = e & 63, ts = e >> 6, r = l & 31, hh = l >> 5, g = (r >> 2) & 1;
        uint32_t slo, shi;
        frag_selectors((int)((r & 3) + 4 * (r >> 3)), slo, shi);
        const uint64_t *dw = inv_d + ((ts >> 2) * 4 + g * 2 + hh) * 8 + 2 * (ts & 3);
        inv_lds[e] = expand_frag(dw[0], dw[1], slo, shi);
    }
}
// One window, everything that reads the image: the operands of the sections' product for the lane's PTS points (n, and n + 32) --
// half h of the 28 values (cube(INV_MDS (next - ark2))_i, i = 2 m + h, then cube(cur_j), j = 2 t + h), centred and byte-offset, two
// per k-step.  imgA = the wave's image + n; ark2 = this row's 14 constants; k_lds = the row constants (four words each) of the
// inverse matrix.  DBUF > 0: tile T + 1 on the matrix pipe while the vector ALU recombines and cubes tile T -- a wave issues in
// order, so the MFMAs are spread through the vector work (one per DBUF vector instructions), not put in front of it.
template <int PTS, int DBUF>
__device__ __forceinline__ void window_operands(const fp *imgA, const fp *ark2, const v4i *inv_lds, const uint64_t *k_lds, const Role &ro,
                                                const Shifts &sh, v4i (&c)[PTS][MF_KS_SEC]) {
    constexpr int RWR = rw_rows<PTS>;
    const int h = ro.h, lane = ro.lane;
    // operands of the inverse matrix: (next - ark2) of columns 4 s + 2 h, + 1 (k-step 3 of half 1 is padding: its table bytes are 0)
    v4i b[PTS][MF_KS_INV];
#pragma unroll
    for (int s = 0; s < MF_KS_INV; s++) {
        const int jb = (s == 3 && h) ? 12 : 4 * s + 2 * h;
        const fp k0 = ark2[jb], k1 = ark2[jb + 1];
#pragma unroll
        for (int pt = 0; pt < PTS; pt++)
            b[pt][s] = pack2(fp_sub(imgA[jb * RWR + 32 * pt + 1], k0) ^ X0, fp_sub(imgA[(jb + 1) * RWR + 32 * pt + 1], k1) ^ X0);
    }
    uint64_t held[PTS];
    auto inv_tile = [&](int T, v16i (&a)[PTS]) {
#pragma unroll
        for (int pt = 0; pt < PTS; pt++) a[pt] = acc_start();
#pragma unroll
        for (int s = 0; s < MF_KS_INV; s++) {
            const v4i fr = inv_lds[(T * MF_KS_INV + s) * 64 + lane];
#pragma unroll
            for (int pt = 0; pt < PTS; pt++) a[pt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr, b[pt][s], a[pt], 0, 0, 0);
        }
    };
    auto finish_tile = [&](int T, const v16i (&a)[PTS]) { // recombine, reduce, cube: output i = 2 T + h, entry m = T of the operand
        const uint64_t *kp = k_lds + 4 * (2 * T + h);
        const uint64_t kw[4] = {kp[0], kp[1], kp[2], kp[3]};
#pragma unroll
        for (int pt = 0; pt < PTS; pt++) {
            const uint64_t x = (fp_cube(recombine_mad(a[pt], kw, sh)) - HALF_P) ^ X0;
            if (T & 1) c[pt][T >> 1] = pack2(held[pt], x);
            else held[pt] = x;
        }
    };
    if constexpr (DBUF > 0) {
        v16i acc[2][PTS];
        inv_tile(0, acc[0]);
#pragma unroll
        for (int T = 0; T < MF_TILES_INV; T++) {
            if (T + 1 < MF_TILES_INV) inv_tile(T + 1, acc[(T + 1) & 1]);
            finish_tile(T, acc[T & 1]);
            if (T + 1 < MF_TILES_INV) {
#pragma unroll
                for (int e = 0; e < 4 * PTS; e++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, DBUF, 0); // VALU
                }
            }
        }
    } else {
#pragma unroll
        for (int T = 0; T < MF_TILES_INV; T++) {
            v16i acc[PTS];
            inv_tile(T, acc);
            finish_tile(T, acc);
        }
    }
    // forward half: cube(cur_j), j = 2 t + h: entries m = 7 + t
#pragma unroll
    for (int t = 0; t < 7; t++) {
#pragma unroll
        for (int pt = 0; pt < PTS; pt++) {
            const uint64_t x = (fp_cube(imgA[(2 * t + h) * RWR + 32 * pt]) - HALF_P) ^ X0;
            if (t & 1) held[pt] = x;
            else c[pt][3 + (t >> 1)] = pack2(held[pt], x);
        }
    }
}
// one section tile against the window's operands: r[pt] = the value of this lane's section (output g = h of the tile) at the
// lane's points: gamma . cube(INV_MDS (next - ark2)) - U . cube(cur).  tile_d = the tile's compact entries [g][h][16], kp = the
// four words of the lane's row constant
template <int PTS>
__device__ __forceinline__ void section_tile(const uint64_t *tile_d, const uint64_t *kp, const Role &ro, const Shifts &sh, const v4i (&c)[PTS][MF_KS_SEC],
                                             fp (&r)[PTS]) {
    v16i a[PTS];
#pragma unroll
    for (int pt = 0; pt < PTS; pt++) a[pt] = acc_start();
    const uint64_t *dw = tile_d + (ro.ag * 2 + ro.h) * 16;
#pragma unroll
    for (int s = 0; s < MF_KS_SEC; s++) {
        const v4i fr = expand_frag(dw[2 * s], dw[2 * s + 1], ro.sel_lo, ro.sel_hi);
#pragma unroll
        for (int pt = 0; pt < PTS; pt++) a[pt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr, c[pt][s], a[pt], 0, 0, 0);
    }
    const uint64_t kw[4] = {kp[0], kp[1], kp[2], kp[3]};
#pragma unroll
    for (int pt = 0; pt < PTS; pt++) r[pt] = recombine_mad(a[pt], kw, sh);
}
// value of the other lane half (lane ^ 32)
__device__ __forceinline__ fp other_half(fp v) {
    const uint32_t lo = __shfl_xor((uint32_t)v, 32), hi = __shfl_xor((uint32_t)(v >> 32), 32);
    return ((uint64_t)hi << 32) | lo;
}

#ifndef RM_DBUF1
#define RM_DBUF1 0 // the same for one point per lane
#endif
#ifndef RM_PTS
// points per lane of k_rounds_mfma: 2 (a wave = 64 points, 222 registers, two waves per SIMD) or 1 (32 points, 133 registers, three
// waves per SIMD, 54 KB of LDS per workgroup).  Measured: 1 is SLOWER, 2.14-2.23 against 1.89-1.94 ms -- every table fragment is read and
// expanded for half as many points, and the third wave does not pay for that: occupancy is not what holds the kernel back.
#define RM_PTS 2
#endif
constexpr size_t RM_LDS_INV = (size_t)MF_TILES_INV * MF_KS_INV * 64 * 16;                 // expanded inverse table
template <int PTS, int NT = RM_NT> constexpr size_t rm_lds_img = (size_t)(NT / 64) * rw_img<PTS> * 8;
constexpr size_t RM_LDS_SEC = (size_t)MF_TILES_SEC * 4 * 16 * 8;                          // compact section tables of one coefficient set
constexpr size_t RM_LDS_KINV = 14 * 32, RM_LDS_KSEC = (size_t)2 * MF_TILES_SEC * 32;       // row constants: inverse matrix | sections of a set
constexpr size_t RM_LDS_ARK = 8 * 14 * 8, RM_LDS_ATAB = (size_t)2 * MF_TILES_SEC * 8 * 8; // A[section of (tile, g)][row mod 8] of a set
template <int PTS, int M, int NT> constexpr size_t rm_lds = RM_LDS_INV + rm_lds_img<PTS, NT> + RM_LDS_KINV + RM_LDS_ARK + M * (RM_LDS_SEC + RM_LDS_KSEC + RM_LDS_ATAB);

// out = per coefficient set [13 polynomials: the first four written here][4 even cosets][n], as k_rounds_split<M> writes them.
// grid = (workgroups per coset, even cosets of the window): a workgroup expands its tables once and then takes every gridDim.x-th
// block of NT / 64 * 32 PTS rows of its coset.  M coefficient sets (the components of an extension-field proof): the windows' operands
// are formed once, every set has its own section tables (rtab + c * CE_RTAB_WORDS), constants and sums.
template <int PTS, int M, int NT>
__global__ __launch_bounds__(NT, PTS == 1 ? 3 : RM_WAVES) void k_rounds_mfma(CeParams p, fp *__restrict__ out) {
    constexpr int RWR = rw_rows<PTS>, WROWS = 32 * PTS, BROWS = NT / 64 * WROWS; // rows per wave, per workgroup block
    constexpr int RM_NT = NT; // (shadows the default: the loops below stride by the workgroup size)
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    v4i *inv_lds = (v4i *)lds;
    fp *img_all = (fp *)(lds + RM_LDS_INV);
    uint64_t *kinv_lds = (uint64_t *)(lds + RM_LDS_INV + rm_lds_img<PTS, NT>);
    fp *ark2_lds = kinv_lds + RM_LDS_KINV / 8;
    uint64_t *secd_lds = ark2_lds + RM_LDS_ARK / 8;           // [M][13 tiles][64]
    uint64_t *ksec_lds = secd_lds + M * (RM_LDS_SEC / 8);     // [M][26][4]
    fp *atab_lds = ksec_lds + M * (RM_LDS_KSEC / 8);          // [M][26][8]

    const size_t n = (size_t)1 << p.log_n;
    const unsigned kk = 2 * blockIdx.y, kc = (p.k0 >> 1) + blockIdx.y, ka = 2 * kc; // (k_rounds_split)
    const int tid = threadIdx.x, lane = tid & 63;
    const Role ro = make_role(lane);
    const int nn = ro.nn, h = ro.h;
    const fp *rt = p.rtab;
    expand_inverse_table(inv_lds, rt + MF_INV_D, tid, RM_NT);
    for (unsigned e = tid; e < RM_LDS_KINV / 8; e += RM_NT) kinv_lds[e] = rt[MF_K + e];
    for (unsigned e = tid; e < M * (RM_LDS_SEC / 8); e += RM_NT) secd_lds[e] = rt[(size_t)(e / (RM_LDS_SEC / 8)) * CE_RTAB_WORDS + MF_SEC_D + e % (RM_LDS_SEC / 8)];
    for (unsigned e = tid; e < M * (RM_LDS_KSEC / 8); e += RM_NT) ksec_lds[e] = rt[(size_t)(e / (RM_LDS_KSEC / 8)) * CE_RTAB_WORDS + MF_K + 14 * 4 + e % (RM_LDS_KSEC / 8)];
    if (tid < 8 * 14) { // the round constants' extension has period 8 in the row index: the same 8 x 14 values for every block of rows
        const unsigned r = tid / 14, c = tid % 14;
        ark2_lds[tid] = p.ptab[((size_t)ka * 48 + P_ARK + 14 + c) * 1024 + r];
    }
    for (unsigned e = tid; e < M * 2 * MF_TILES_SEC * 8; e += RM_NT) { // (set, tile, g) -> section (window, flag set, slot 2 u + g)
        const int c = e / (2 * MF_TILES_SEC * 8), r = e % (2 * MF_TILES_SEC * 8), tile = r >> 4, g = (r >> 3) & 1, w = mf_tile_window(tile), local = tile - mf_tile_base(w);
        const int sec = (w * 2 + mf_tile_fs(w, local)) * 4 + 2 * mf_tile_pair(w, local) + g;
        atab_lds[e] = rt[(size_t)c * CE_RTAB_WORDS + RT_A + sec * 64 + ka * 8 + (r & 7)];
    }
    __syncthreads();

    fp *img = img_all + (size_t)(tid >> 6) * rw_img<PTS>;
    const uint64_t *k_lds = kinv_lds;
    const fp *imgA = img + nn; // point n: current row at element n, next row at n + 1; point n + 32: + 32
    const unsigned jrp = (unsigned)(nn & 7); // row of the lane's points mod 8 (blocks and waves start at multiples of 32)
    const fp *ark2 = ark2_lds + jrp * 14;
    const fp *per = p.ptab + (size_t)(p.k0 + kk) * 48 * 1024;
    const fp *colbase = p.lde + (size_t)kk * 94 * n;
    const Shifts sh{opaque_const<1>(), opaque_const<1u << 8>(), opaque_const<1u << 16>(), opaque_const<1u << 24>()};
    const size_t nblk = n / BROWS;
    size_t blk = blockIdx.x;
    // rows jw + 2 lane, + 1 of the wave's points (first row jw); rows n, n + 1 of the coset's last wave wrap to 0, 1
    auto rows_of = [&](size_t b) {
        const size_t jw = b * (size_t)BROWS + (size_t)(tid >> 6) * WROWS;
        return colbase + jw + 2 * lane - ((lane == 16 * PTS && jw + WROWS == n) ? n : 0);
    };
    const fp *rows = rows_of(blk);
    if (blk < nblk) fetch_window<PTS>(rows, n, c_windows[0].reg, lane, img);
#pragma unroll 1
    for (; blk < nblk; blk += gridDim.x) {
        const size_t jw = blk * (size_t)BROWS + (size_t)(tid >> 6) * WROWS; // the wave's first row
        fp fl[PTS][3];
#pragma unroll
        for (int pt = 0; pt < PTS; pt++) {
            const size_t r = (jw + nn + 32 * pt) & 1023;
            fl[pt][0] = per[(size_t)P_SETUP * 1024 + r]; fl[pt][1] = per[(size_t)P_HASH * 1024 + r]; fl[pt][2] = per[(size_t)P_SCHNORR_HASH * 1024 + r];
        }
        // Sums per STREAM, not per polynomial: a lane's sections are slot u = 0 (h = 0: alpha; h = 1: slot 1) and slot u = 1 (h = 0:
        // slot 2; h = 1: slot 3) of every flag set, and which polynomial a slot feeds depends on the window alone -- so the register
        // a term goes to is the same for both lane halves: stream 0 -> polynomial of slot 1 (windows 0..4: 1, 2, 1, 3, 3), stream 1 ->
        // polynomial of slot 2 (2, 3, 3).  Half 0 reads stream 0 as alpha whatever the register, half 1 reads stream 1 as polynomial 1
        // (slot 3 is used in window 1 only, there by group 0).
        fp sm[M][PTS][5]; // per set: stream 0 -> polynomials 1, 2, 3 | stream 1 -> polynomials 2, 3
#pragma unroll
        for (int c = 0; c < M; c++)
#pragma unroll
            for (int pt = 0; pt < PTS; pt++)
#pragma unroll
                for (int q = 0; q < 5; q++) sm[c][pt][q] = 0;
#pragma unroll 1
        for (int wdx = 0; wdx < 5; wdx++) {
            const RoundWindow w = c_windows[wdx];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            v4i c[PTS][MF_KS_SEC];
            window_operands<PTS, PTS == 1 ? RM_DBUF1 : RM_DBUF>(imgA, ark2, inv_lds, k_lds, ro, sh, c);
            // the image is free again: the next window (of this block, or the first one of the workgroup's next block) arrives behind
            // the sections' product
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef RM_EXP_NODMA // measurement builds: the windows are not fetched (the image keeps the first window)
            if (wdx < 4) fetch_window<PTS>(rows, n, c_windows[wdx + 1].reg, lane, img);
            else if (blk + gridDim.x < nblk) { rows = rows_of(blk + gridDim.x); fetch_window<PTS>(rows, n, c_windows[0].reg, lane, img); }
#endif
#pragma unroll
            for (int cs = 0; cs < M; cs++) { // one coefficient set after the other against the same operands
                fp x[PTS][2]; // the window's terms of stream u
#pragma unroll
                for (int pt = 0; pt < PTS; pt++) x[pt][0] = x[pt][1] = 0;
#ifdef RM_EXP_NOSEC // measurement builds: without the sections' product
                const int t0 = 0, t1 = 0;
                x[0][0] = (fp)c[0][0][0] ^ (fp)c[0][6][3];
#else
                const int t0 = mf_tile_base(wdx), t1 = mf_tile_base(wdx + 1);
#endif
#pragma unroll 1
                for (int tile = t0; tile < t1; tile++) {
                    const int local = tile - t0, fs = mf_tile_fs(wdx, local), u = mf_tile_pair(wdx, local);
                    fp r[PTS];
                    section_tile<PTS>(secd_lds + (cs * MF_TILES_SEC + tile) * 64, ksec_lds + 4 * (cs * 2 * MF_TILES_SEC + 2 * tile + h), ro, sh, c, r);
                    const fp at = atab_lds[(cs * 2 * MF_TILES_SEC + 2 * tile + h) * 8 + jrp]; // this lane's section: slot 2 u + h
                    const int fg = fs ? w.flag_b : w.flag_a;
#pragma unroll
                    for (int pt = 0; pt < PTS; pt++) {
                        const fp f = fg == 0 ? fl[pt][0] : fg == 1 ? fl[pt][1] : fg == 2 ? fl[pt][2] : fp_add(fl[pt][0], fl[pt][1]);
                        const fp v = fp_mul(f, fp_sub(r[pt], at));
                        if (u) x[pt][1] = fp_add(x[pt][1], v);
                        else x[pt][0] = fp_add(x[pt][0], v);
                    }
                }
                // stream 0 -> polynomial of slot 1: 1, 2, 1, 3, 3; stream 1 -> polynomial of slot 2: 2, 3, 3 (windows 3, 4: no pair 1)
#pragma unroll
                for (int pt = 0; pt < PTS; pt++) {
                    if (wdx == 0 || wdx == 2) sm[cs][pt][0] = fp_add(sm[cs][pt][0], x[pt][0]);
                    else if (wdx == 1) sm[cs][pt][1] = fp_add(sm[cs][pt][1], x[pt][0]);
                    else sm[cs][pt][2] = fp_add(sm[cs][pt][2], x[pt][0]);
                    if (wdx == 0) sm[cs][pt][3] = fp_add(sm[cs][pt][3], x[pt][1]);
                    else if (wdx < 3) sm[cs][pt][4] = fp_add(sm[cs][pt][4], x[pt][1]);
                }
            }
        }
        // polynomials 0..3 of this lane's sections: half 0: stream 0 is alpha, stream 1 by register; half 1: stream 0 by register,
        // stream 1 is polynomial 1.  The other half of a point's sections sits in lane ^ 32.
#pragma unroll
        for (int cs = 0; cs < M; cs++) {
            fp tot[PTS][4];
#pragma unroll
            for (int pt = 0; pt < PTS; pt++) {
                const fp s012 = fp_add(fp_add(sm[cs][pt][0], sm[cs][pt][1]), sm[cs][pt][2]), s34 = fp_add(sm[cs][pt][3], sm[cs][pt][4]);
                tot[pt][0] = h ? 0 : s012; tot[pt][1] = h ? fp_add(sm[cs][pt][0], s34) : 0;
                tot[pt][2] = h ? sm[cs][pt][1] : sm[cs][pt][3]; tot[pt][3] = h ? sm[cs][pt][2] : sm[cs][pt][4];
            }
            fp *o = out + (size_t)cs * CE_SPLIT_TABLES * 4 * n; // the set's block of polynomials (k_rounds_split)
            if constexpr (PTS == 2) { // lane (n, 0) holds its part of point n in [0] and lane (n, 1) the rest, likewise [1] for point n + 32
                const size_t j = jw + lane;
#pragma unroll
                for (int q = 0; q < 4; q++) { // this lane's own point: its part + what the other half holds of it
                    const fp mine = h ? tot[1][q] : tot[0][q], give = h ? tot[0][q] : tot[1][q];
                    o[((size_t)q * 4 + kc) * n + j] = fp_add(mine, other_half(give)); // table 3: group 2 (k_rounds_split)
                }
            } else { // both halves hold parts of point n: half 0 writes polynomials 0, 1 and half 1 writes 2, 3
                const size_t j = jw + nn;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const fp got = other_half(h ? tot[0][e] : tot[0][2 + e]);
                    const int q = 2 * h + e; // (lane-dependent address only)
                    o[((size_t)q * 4 + kc) * n + j] = fp_add(h ? tot[0][2 + e] : tot[0][e], got);
                }
            }
        }
    }
}

// MerkleAir's four round gadgets / the one of SchnorrAir's message hash (k_merkle_rounds of constraints.hip, same template arguments
// and the same values): per window ONE section tile -- output 0 = the alpha section, output 1 = the beta section (the round slots
// share one declared degree) -- so half 0 of a wave sums alpha terms and half 1 beta terms; flag (alpha sum + x^adj beta sum) at the end.
constexpr size_t MRM_LDS_SEC = 4 * 64 * 8, MRM_LDS_K = (14 + 8) * 32, MRM_LDS_ATAB = MR_SECTIONS * 8 * 8;
constexpr size_t MRM_LDS = RM_LDS_INV + rm_lds_img<2> + MRM_LDS_SEC + MRM_LDS_K + RM_LDS_ARK + MRM_LDS_ATAB;
struct MerkleRoundsParams { // what the kernel reads of AirCombineParams (the whole block costs registers): x^adj = xshift[k] w^(j xadj)
    const fp *lde, *w;
    fp *out;
    fp xshift[8];
    uint64_t xadj;
    uint32_t log_n, k0, stride;
};
template <int W0, int NWIN, int WIDTH, int PCOLS, int FLAGCOL, int ARKCOL, bool ADD>
__global__ __launch_bounds__(RM_NT, RM_WAVES) void k_merkle_rounds_mfma(MerkleRoundsParams p, const fp *__restrict__ ptab, const fp *__restrict__ rtab) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    v4i *inv_lds = (v4i *)lds;
    fp *img_all = (fp *)(lds + RM_LDS_INV);
    uint64_t *secd_lds = (uint64_t *)(lds + RM_LDS_INV + rm_lds_img<2>);
    uint64_t *k_lds = secd_lds + MRM_LDS_SEC / 8;
    fp *ark2_lds = k_lds + MRM_LDS_K / 8;
    fp *atab_lds = ark2_lds + RM_LDS_ARK / 8;

    const size_t n = (size_t)1 << p.log_n;
    const unsigned kk = blockIdx.y, k = p.k0 + kk;
    if (k % p.stride) return; // uniform over the workgroup
    const int tid = threadIdx.x, lane = tid & 63;
    const Role ro = make_role(lane);
    const int nn = ro.nn, h = ro.h;
    expand_inverse_table(inv_lds, rtab + MRF_INV_D, tid, RM_NT);
    for (unsigned e = tid; e < NWIN * 64; e += RM_NT) secd_lds[e] = rtab[MRF_SEC_D + e];
    for (unsigned e = tid; e < (14 + 2 * NWIN) * 4; e += RM_NT) k_lds[e] = rtab[MRF_K + e];
    if (tid < 8 * 14) {
        const unsigned r = tid / 14, c = tid % 14;
        ark2_lds[tid] = ptab[((size_t)k * PCOLS + ARKCOL + 14 + c) * 512 + r];
    }
    if (tid < MR_SECTIONS * 8) atab_lds[tid] = rtab[MR_A + (tid >> 3) * 64 + k * 8 + (tid & 7)];
    __syncthreads();

    fp *img = img_all + (size_t)(tid >> 6) * rw_img<2>;
    const fp *imgA = img + nn;
    const unsigned jrp = (unsigned)(nn & 7);
    const fp *ark2 = ark2_lds + jrp * 14;
    const fp *colbase = p.lde + (size_t)kk * WIDTH * n;
    const Shifts sh{opaque_const<1>(), opaque_const<1u << 8>(), opaque_const<1u << 16>(), opaque_const<1u << 24>()};
    const size_t nblk = n / RM_NT;
    size_t blk = blockIdx.x;
    auto rows_of = [&](size_t b) {
        const size_t jw = b * (size_t)RM_NT + (size_t)(tid >> 6) * 64;
        return colbase + jw + 2 * lane - ((lane == 32 && jw + 64 == n) ? n : 0);
    };
    const fp *rows = rows_of(blk);
    if (blk < nblk) fetch_window<2>(rows, n, c_windows[W0].reg, lane, img);
#pragma unroll 1
    for (; blk < nblk; blk += gridDim.x) {
        const size_t j = blk * (size_t)RM_NT + (size_t)(tid >> 6) * 64 + lane; // this lane's own point
        fp tA = 0, tB = 0; // half 0: alpha sums of points n, n + 32; half 1: beta sums
#pragma unroll 1
        for (int wdx = 0; wdx < NWIN; wdx++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            v4i c[2][MF_KS_SEC];
            window_operands<2, 0>(imgA, ark2, inv_lds, k_lds, ro, sh, c); // one tile at a time: 190-198 registers (double-buffered: 256 and 16 spills)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (wdx + 1 < NWIN) fetch_window<2>(rows, n, c_windows[W0 + wdx + 1].reg, lane, img);
            else if (blk + gridDim.x < nblk) { rows = rows_of(blk + gridDim.x); fetch_window<2>(rows, n, c_windows[W0].reg, lane, img); }
            __builtin_amdgcn_sched_barrier(0);
            fp r[2];
            section_tile<2>(secd_lds + wdx * 64, k_lds + 4 * (14 + 2 * wdx + h), ro, sh, c, r);
            const fp at = atab_lds[(2 * wdx + h) * 8 + jrp];
            tA = fp_add(tA, fp_sub(r[0], at));
            tB = fp_add(tB, fp_sub(r[1], at));
        }
        const fp got = other_half(h ? tA : tB); // half 0 hands the alpha sum of point n + 32 over, half 1 the beta sum of point n
        const fp ta = h ? got : tA, tb = h ? tB : got;
        const fp flag = ptab[((size_t)k * PCOLS + FLAGCOL) * 512 + (j & 511)];
        const fp xp = fp_mul(p.xshift[k], p.w[(j * p.xadj) & (n - 1)]);
        const fp v = fp_mul(flag, fp_add(ta, fp_mul(xp, tb)));
        fp *o = p.out + (size_t)kk * n + j;
        *o = ADD ? fp_add(*o, v) : v;
    }
}

} // namespace

template <int PTS, int M, int NT>
static hipError_t launch_rounds_mfma_as(const CeParams &p, uint64_t *d_even, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    constexpr int BROWS = NT / 64 * 32 * PTS;
    constexpr size_t LDS = rm_lds<PTS, M, NT>;
    if (n % BROWS) return hipErrorInvalidValue;
    static const hipError_t attr = hipFuncSetAttribute((const void *)k_rounds_mfma<PTS, M, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(k_rounds_mfma_tables, dim3(MF_TILES_INV + MF_TILES_SEC, M), dim3(64), 0, stream, p.rtab);
    const unsigned ny = p.nkc ? p.nkc : 4, nblk = (unsigned)(n / BROWS), gx = RM_RESIDENT / ny < nblk ? RM_RESIDENT / ny : nblk;
    hipLaunchKernelGGL((k_rounds_mfma<PTS, M, NT>), dim3(gx, ny), dim3(NT), LDS, stream, p, d_even);
    return hipGetLastError();
}
// one coefficient set: two workgroups of 256 per CU; two sets: the same with 76 KB of LDS each; three: one workgroup of 512 (114 KB)
hipError_t launch_rounds_mfma(const CeParams &p, uint64_t *d_even, hipStream_t stream) {
    const unsigned m = p.m ? p.m : 1;
    if (m == 1) return launch_rounds_mfma_as<RM_PTS, 1, RM_NT>(p, d_even, stream);
    if (m == 2) return launch_rounds_mfma_as<2, 2, 256>(p, d_even, stream);
    if (m == 3) return launch_rounds_mfma_as<2, 3, 512>(p, d_even, stream);
    return hipErrorInvalidValue;
}

// constraints.hip: after k_merkle_rounds_setup (same stream); which = 0: MerkleAir's four windows (written), 1: SchnorrAir's message hash (added)
hipError_t launch_merkle_rounds_mfma(const AirCombineParams &p, const uint64_t *ptab, uint64_t *d_rtab, unsigned nk, unsigned round_group, int which,
                                     hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    if (n % RM_NT) return hipErrorInvalidValue;
    const unsigned nblk = (unsigned)(n / RM_NT), gx = RM_RESIDENT / nk ? (RM_RESIDENT / nk < nblk ? RM_RESIDENT / nk : nblk) : 1;
    if (round_group >= AIR_MAX_GROUPS) return hipErrorInvalidValue;
    MerkleRoundsParams q;
    q.lde = p.lde; q.w = p.w; q.out = p.out; q.xadj = p.tgrp_adj[round_group]; q.log_n = p.log_n; q.k0 = p.k0; q.stride = p.stride;
    for (int k = 0; k < 8; k++) q.xshift[k] = p.tgrp_shift[k][round_group];
    if (which == 0) {
        auto kern = k_merkle_rounds_mfma<0, 4, 65, 33, 4, 5, false>;
        static const hipError_t attr = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MRM_LDS);
        if (attr != hipSuccess) return attr;
        hipLaunchKernelGGL(k_merkle_rounds_mfma_tables, dim3(MF_TILES_INV + 4), dim3(64), 0, stream, d_rtab);
        hipLaunchKernelGGL(kern, dim3(gx, nk), dim3(RM_NT), MRM_LDS, stream, q, ptab, (const fp *)d_rtab);
    } else {
        auto kern = k_merkle_rounds_mfma<4, 1, 56, 36, 7, 8, true>;
        static const hipError_t attr = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MRM_LDS);
        if (attr != hipSuccess) return attr;
        hipLaunchKernelGGL(k_merkle_rounds_mfma_tables, dim3(MF_TILES_INV + 1), dim3(64), 0, stream, d_rtab);
        hipLaunchKernelGGL(kern, dim3(gx, nk), dim3(RM_NT), MRM_LDS, stream, q, ptab, (const fp *)d_rtab);
    }
    return hipGetLastError();
}

} // namespace cs
