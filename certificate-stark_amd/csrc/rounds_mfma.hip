// TransactionAir's five Rescue windows (split evaluation, one coefficient set) with EVERY constant-matrix product on the matrix cores:
//   y = INV_MDS (next - ark2)                                    14 x 14, constant                     (src/utils/rescue.rs:345-375)
//   v_sec = gamma_sec . cube(y) - U_sec . cube(cur),  U = MDS^T gamma   up to 8 sections x 28, per proof   (rescue.rs:269-300 folded)
// Same values as k_rounds_split (constraints.hip), bit for bit: the 64-bit integer products are byte-decomposed exactly as in
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
//     two points and the two halves are added once at the end (one cross-half exchange of eight values).
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

constexpr int RM_NT = 512;                      // threads per workgroup: eight waves share one expanded inverse table
constexpr int RW_ROWS = 66, RW_IMG = 14 * RW_ROWS; // window image of a wave: 14 columns x rows j0 .. j0 + 65 (k_rounds_split's)
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

// Compact tables of one proof from what k_rounds_setup left in rtab (RT_G: the sections' coefficient vectors; RT_UL: limbs of their
// MDS-folded forms).  grid = 7 inverse tiles + 13 section tiles, 64 threads.
__global__ void k_rounds_mfma_tables(fp *__restrict__ rtab) {
    const int t = threadIdx.x;
    if (blockIdx.x < MF_TILES_INV) { // inverse matrix, tile T: entries [g][h][m], m < 8
        const int T = blockIdx.x;
        if (t < 32) {
            const int g = t >> 4, h = (t >> 3) & 1, m = t & 7, i = 2 * T + g, j = 4 * (m >> 1) + 2 * h + (m & 1);
            rtab[MF_INV_D + (T * 4 + g * 2 + h) * 8 + m] = j < 14 ? digit_word((int64_t)c_inv_mds[i * 14 + j]) : 0;
        } else if (t < 34) { // row constant of output i: X0 sum_j M_ij - offset (mod 2^128), the sum itself is below 14 p^2
            const int i = 2 * T + (t - 32);
            u128 k = 0;
            for (int j = 0; j < 14; j++) k += (u128)c_inv_mds[i * 14 + j] * X0;
            k -= acc0_offset();
            rtab[MF_K + 2 * i] = (uint64_t)k;
            rtab[MF_K + 2 * i + 1] = (uint64_t)(k >> 64);
        }
        return;
    }
    const int tile = blockIdx.x - MF_TILES_INV, w = mf_tile_window(tile), local = tile - mf_tile_base(w);
    const int fs = mf_tile_fs(w, local), u = mf_tile_pair(w, local);
    __shared__ fp csum[2][32];
    if (t < 56) { // [g][h][m], m < 14
        const int g = t / 28, h = (t / 14) & 1, m = t % 14, sec = (w * 2 + fs) * 4 + 2 * u + g;
        fp c;
        if (m < 7) c = rtab[RT_G + sec * 14 + 2 * m + h];
        else {
            const uint32_t *l = (const uint32_t *)(rtab + RT_UL) + (sec * 14 + 2 * (m - 7) + h) * 4;
            c = fp_neg((fp)l[0] | ((fp)l[1] << 21) | ((fp)l[2] << 42));
        }
        rtab[MF_SEC_D + (tile * 4 + g * 2 + h) * 16 + m] = digit_word(c > HALF_P ? (int64_t)(c - FP_P) : (int64_t)c);
        csum[g][h * 14 + m] = c;
    } else if (t < 64) { // m = 14, 15 of every (g, h): padding of the last k-step's rows
        const int q = t - 56;
        rtab[MF_SEC_D + (tile * 4 + (q >> 1)) * 16 + 14 + (q & 1)] = 0;
    }
    __syncthreads();
    if (t < 2) { // row constant: (X0 + (p-1)/2) sum(c) mod p  -  offset  +  2p 2^64
        fp s = 0;
        for (int e = 0; e < 28; e++) s = fp_add(s, csum[t][e]);
        const fp kc = fp_mul(fp_mul(X0 + HALF_P, s), FP_R2); // the plain product of the two integers mod p
        u128 k = (u128)kc + ((u128)(2 * FP_P) << 64);
        k -= acc0_offset();
        rtab[MF_K + 2 * (14 + 2 * tile + t)] = (uint64_t)k;
        rtab[MF_K + 2 * (14 + 2 * tile + t) + 1] = (uint64_t)(k >> 64);
    }
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
// LDS-DMA of one window: lane l < 33 moves rows j0 + 2l, j0 + 2l + 1 of the 14 columns into the wave's image (k_rounds_split)
__device__ __forceinline__ void fetch_window(const fp *rows, size_t n, int reg, int lane, fp *img) {
    if (lane < 33) {
#pragma unroll
        for (int j = 0; j < 14; j++)
            __builtin_amdgcn_global_load_lds((glb_void *)(rows + (size_t)(reg + j) * n), (lds_void *)(img + j * RW_ROWS), 16, 0, 0);
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
__device__ __forceinline__ void tot_add(fp (&t)[4], int q, fp v) { // q uniform
    if (q == 0) t[0] = fp_add(t[0], v);
    else if (q == 1) t[1] = fp_add(t[1], v);
    else if (q == 2) t[2] = fp_add(t[2], v);
    else t[3] = fp_add(t[3], v);
}

constexpr size_t RM_LDS_INV = (size_t)MF_TILES_INV * MF_KS_INV * 64 * 16;                 // expanded inverse table
constexpr size_t RM_LDS_IMG = (size_t)(RM_NT / 64) * RW_IMG * 8;
constexpr size_t RM_LDS_SEC = (size_t)MF_TILES_SEC * 4 * 16 * 8;                          // compact section tables
constexpr size_t RM_LDS_K = (size_t)(14 + 2 * MF_TILES_SEC) * 16;
constexpr size_t RM_LDS_ARK = 8 * 14 * 8 + 16, RM_LDS_ATAB = (size_t)RT_SECTIONS * 8 * 8; // + 16: the padded columns 14, 15 of row 7
constexpr size_t RM_LDS = RM_LDS_INV + RM_LDS_IMG + RM_LDS_SEC + RM_LDS_K + RM_LDS_ARK + RM_LDS_ATAB;

// out = [4 polynomials][4 even cosets][n] as k_rounds_split<1> writes them.  grid = (n / RM_NT, even cosets of the window)
__global__ __launch_bounds__(RM_NT, 2) void k_rounds_mfma(CeParams p, fp *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    v4i *inv_lds = (v4i *)lds;
    fp *img_all = (fp *)(lds + RM_LDS_INV);
    uint64_t *secd_lds = (uint64_t *)(lds + RM_LDS_INV + RM_LDS_IMG);
    uint64_t *k_lds = secd_lds + RM_LDS_SEC / 8;
    fp *ark2_lds = k_lds + RM_LDS_K / 8;
    fp *atab_lds = ark2_lds + RM_LDS_ARK / 8;

    const size_t n = (size_t)1 << p.log_n;
    const unsigned kk = 2 * blockIdx.y, kc = (p.k0 >> 1) + blockIdx.y, ka = 2 * kc; // (k_rounds_split)
    const int tid = threadIdx.x, lane = tid & 63, nn = lane & 31, h = lane >> 5;
    const size_t jw = blockIdx.x * (size_t)RM_NT + (size_t)(tid >> 6) * 64; // the wave's first row
    // tile row of this lane as an A operand: r = lane & 31 -> output g = (r >> 2) & 1, byte diagonal d = (r & 3) + 4 (r >> 3)
    const int ag = (nn >> 2) & 1, ad = (nn & 3) + 4 * (nn >> 3);
    uint32_t sel_lo = 0, sel_hi = 0; // byte b of the fragment word = digit ad - b of the entry (0x0c: constant zero)
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int a0 = ad - b, a1 = ad - 4 - b;
        sel_lo |= (uint32_t)((a0 >= 0 && a0 <= 7 && ad < 15) ? a0 : 0x0c) << (8 * b);
        sel_hi |= (uint32_t)((a1 >= 0 && a1 <= 7 && ad < 15) ? a1 : 0x0c) << (8 * b);
    }
    const fp *rt = p.rtab;
    for (unsigned e = tid; e < MF_TILES_INV * MF_KS_INV * 64; e += RM_NT) { // fragment (T, s) of lane l: entries [T][g][h][2s], [2s + 1]
        const unsigned l = e & 63, ts = e >> 6, r = l & 31, hh = l >> 5, g = (r >> 2) & 1, d = (r & 3) + 4 * (r >> 3);
        uint32_t slo = 0, shi = 0;
        for (int b = 0; b < 4; b++) {
            const int a0 = (int)d - b, a1 = (int)d - 4 - b;
            slo |= (uint32_t)((a0 >= 0 && a0 <= 7 && d < 15) ? a0 : 0x0c) << (8 * b);
            shi |= (uint32_t)((a1 >= 0 && a1 <= 7 && d < 15) ? a1 : 0x0c) << (8 * b);
        }
        const fp *dw = rt + MF_INV_D + ((ts >> 2) * 4 + g * 2 + hh) * 8 + 2 * (ts & 3);
        inv_lds[e] = expand_frag(dw[0], dw[1], slo, shi);
    }
    for (unsigned e = tid; e < RM_LDS_SEC / 8; e += RM_NT) secd_lds[e] = rt[MF_SEC_D + e];
    for (unsigned e = tid; e < RM_LDS_K / 8; e += RM_NT) k_lds[e] = rt[MF_K + e];
    if (tid < 8 * 14) {
        const unsigned r = tid / 14, c = tid % 14;
        ark2_lds[tid] = p.ptab[((size_t)ka * 48 + P_ARK + 14 + c) * 1024 + ((blockIdx.x * (size_t)RM_NT + r) & 1023)];
    }
    if (tid < 2) ark2_lds[8 * 14 + tid] = 0;
    for (unsigned e = tid; e < RT_SECTIONS * 8; e += RM_NT) atab_lds[e] = rt[RT_A + (e >> 3) * 64 + ka * 8 + (e & 7)];
    __syncthreads();

    fp *img = img_all + (size_t)(tid >> 6) * RW_IMG;
    const unsigned jr = (unsigned)(jw & 7); // the rows of both points mod 8 (nn + 32 = nn mod 8): jw + nn
    const fp *ark2 = ark2_lds + ((jr + nn) & 7) * 14;
    const unsigned jrp = (jr + nn) & 7;
    const fp *per = p.ptab + (size_t)(p.k0 + kk) * 48 * 1024;
    const size_t rA = (jw + nn) & 1023, rB = (jw + nn + 32) & 1023;
    const fp flA[3] = {per[(size_t)P_SETUP * 1024 + rA], per[(size_t)P_HASH * 1024 + rA], per[(size_t)P_SCHNORR_HASH * 1024 + rA]};
    const fp flB[3] = {per[(size_t)P_SETUP * 1024 + rB], per[(size_t)P_HASH * 1024 + rB], per[(size_t)P_SCHNORR_HASH * 1024 + rB]};
    const fp *colbase = p.lde + (size_t)kk * 94 * n;
    const fp *rows = colbase + jw + 2 * lane; // rows jw + 2 lane, + 1; rows n, n + 1 of the coset's last wave wrap to 0, 1
    if (lane == 32 && jw + 64 == n) rows -= n;
    fetch_window(rows, n, c_windows[0].reg, lane, img);
    fp totA[4] = {0, 0, 0, 0}, totB[4] = {0, 0, 0, 0};
    const fp *imgA = img + nn; // point n: current row at element n, next row at n + 1; point n + 32: + 32
#pragma unroll 1
    for (int wdx = 0; wdx < 5; wdx++) {
        const RoundWindow w = c_windows[wdx];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // operands of the inverse matrix: (next - ark2) of columns 4 s + 2 h, + 1 (k-step 3 of half 1 is padding: any finite bytes)
        v4i bA[MF_KS_INV], bB[MF_KS_INV];
#pragma unroll
        for (int s = 0; s < MF_KS_INV; s++) {
            const int jb = (s == 3 && h) ? 12 : 4 * s + 2 * h;
            const fp k0 = ark2[jb], k1 = ark2[jb + 1];
            bA[s] = pack2(fp_sub(imgA[jb * RW_ROWS + 1], k0) ^ X0, fp_sub(imgA[(jb + 1) * RW_ROWS + 1], k1) ^ X0);
            bB[s] = pack2(fp_sub(imgA[jb * RW_ROWS + 33], k0) ^ X0, fp_sub(imgA[(jb + 1) * RW_ROWS + 33], k1) ^ X0);
        }
        // cube(INV_MDS d)_i, i = 2 T + h, of both points; operands of the sections' product as they come
        v4i cA[MF_KS_SEC], cB[MF_KS_SEC];
        uint64_t heldA = 0, heldB = 0;
#pragma unroll
        for (int T = 0; T < MF_TILES_INV; T++) {
            v16i a0 = acc_start(), a1 = acc_start();
#pragma unroll
            for (int s = 0; s < MF_KS_INV; s++) {
                const v4i a = inv_lds[(T * MF_KS_INV + s) * 64 + lane];
                a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bA[s], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bB[s], a1, 0, 0, 0);
            }
            const uint64_t klo = k_lds[2 * (2 * T + h)], khi = k_lds[2 * (2 * T + h) + 1];
            const uint64_t xa = (fp_cube(mdsmfma::recombine(a0, klo, khi)) - HALF_P) ^ X0;
            const uint64_t xb = (fp_cube(mdsmfma::recombine(a1, klo, khi)) - HALF_P) ^ X0;
            if (T & 1) { cA[T >> 1] = pack2(heldA, xa); cB[T >> 1] = pack2(heldB, xb); }
            else { heldA = xa; heldB = xb; }
        }
        // forward half: cube(cur_j), j = 2 t + h
#pragma unroll
        for (int t = 0; t < 7; t++) {
            const uint64_t xa = (fp_cube(imgA[(2 * t + h) * RW_ROWS]) - HALF_P) ^ X0;
            const uint64_t xb = (fp_cube(imgA[(2 * t + h) * RW_ROWS + 32]) - HALF_P) ^ X0;
            if (t & 1) { heldA = xa; heldB = xb; }
            else { cA[3 + (t >> 1)] = pack2(heldA, xa); cB[3 + (t >> 1)] = pack2(heldB, xb); }
        }
        if (wdx < 4) { // the image is free again: the next window arrives behind the sections' product
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            fetch_window(rows, n, c_windows[wdx + 1].reg, lane, img);
        }
        const int t0 = mf_tile_base(wdx), t1 = mf_tile_base(wdx + 1);
#pragma unroll 1
        for (int tile = t0; tile < t1; tile++) {
            const int local = tile - t0, fs = mf_tile_fs(wdx, local), u = mf_tile_pair(wdx, local);
            v16i a0 = acc_start(), a1 = acc_start();
            const uint64_t *dw = secd_lds + (tile * 4 + ag * 2 + h) * 16;
#pragma unroll
            for (int s = 0; s < MF_KS_SEC; s++) {
                const v4i a = expand_frag(dw[2 * s], dw[2 * s + 1], sel_lo, sel_hi);
                a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, cA[s], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, cB[s], a1, 0, 0, 0);
            }
            const uint64_t klo = k_lds[2 * (14 + 2 * tile + h)], khi = k_lds[2 * (14 + 2 * tile + h) + 1];
            const int sec = (wdx * 2 + fs) * 4 + 2 * u + h; // this lane's section: slot 2 u + h
            const fp at = atab_lds[sec * 8 + jrp];
            const int fl = fs ? w.flag_b : w.flag_a;
            const fp fa = fl == 0 ? flA[0] : fl == 1 ? flA[1] : fl == 2 ? flA[2] : fp_add(flA[0], flA[1]);
            const fp fb = fl == 0 ? flB[0] : fl == 1 ? flB[1] : fl == 2 ? flB[2] : fp_add(flB[0], flB[1]);
            const fp va = fp_mul(fa, fp_sub(mdsmfma::recombine(a0, klo, khi), at));
            const fp vb = fp_mul(fb, fp_sub(mdsmfma::recombine(a1, klo, khi), at));
            // polynomial of slot 0: alpha; of slot s >= 1: beta of group c_window_groups[.][.][s - 1] (an unused slot's rows are zero)
            const int g0 = u ? c_window_groups[wdx][fs][1] : -2, g1 = c_window_groups[wdx][fs][2 * u];
            const int q0 = g0 == -2 ? 0 : g0 < 0 ? 0 : 1 + g0, q1 = g1 < 0 ? 0 : 1 + g1;
            if (q0 == q1) { tot_add(totA, q0, va); tot_add(totB, q0, vb); }
            else {
                tot_add(totA, q0, h ? 0 : va); tot_add(totB, q0, h ? 0 : vb);
                tot_add(totA, q1, h ? va : 0); tot_add(totB, q1, h ? vb : 0);
            }
        }
    }
    // the two halves of a point: lane (n, 0) holds point n's slots 0, 2 in totA and lane (n, 1) its slots 1, 3; likewise totB for n + 32
    const size_t j = jw + lane;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const fp mine = h ? totB[q] : totA[q], give = h ? totA[q] : totB[q]; // what the other half needs from this lane
        const uint32_t lo = __shfl_xor((uint32_t)give, 32), hi = __shfl_xor((uint32_t)(give >> 32), 32);
        out[((size_t)q * 4 + kc) * n + j] = fp_add(mine, ((uint64_t)hi << 32) | lo); // table 3: group 2 (k_rounds_split)
    }
}

} // namespace

hipError_t launch_rounds_mfma(const CeParams &p, uint64_t *d_even, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    if ((p.m ? p.m : 1) != 1 || n % RM_NT) return hipErrorInvalidValue;
    static const hipError_t attr = hipFuncSetAttribute((const void *)k_rounds_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RM_LDS);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(k_rounds_mfma_tables, dim3(MF_TILES_INV + MF_TILES_SEC), dim3(64), 0, stream, p.rtab);
    hipLaunchKernelGGL(k_rounds_mfma, dim3((unsigned)(n / RM_NT), p.nkc ? p.nkc : 4), dim3(RM_NT), RM_LDS, stream, p, d_even);
    return hipGetLastError();
}

} // namespace cs
