// K2/K3 -- radix-2 NTT over f63 for column-major tables: interpolation (inverse transform) and coset
// low-degree extension.  Engine stage behind `prover.prove(trace)` (/root/reference/src/lib.rs:140;
// winterfell trace.extend [UPSTREAM-RECALL]); the algorithm is the textbook four-step decomposition
// n = R * C, mapped to gfx950 as two kernels per transform:
//
//   k_ntt_cols  one workgroup owns L adjacent matrix columns (all R rows): global accesses are L*8-byte
//               segments at stride C, the R-point sub-transforms run in LDS (DIF, natural in, bit-reversed
//               positions out), the inter-step twiddle w_n^(k1*c) is applied on the way out.
//   k_ntt_rows  one workgroup owns L adjacent rows of the intermediate (each row contiguous in HBM): loads
//               are fully coalesced, the C-point sub-transforms run in LDS, and the store performs the
//               transposition back to natural order (k = k1 + R*k2) in L*8-byte segments.
//
// HBM traffic per transform: read n + write n + read n + write n elements (8 bytes each); twiddle and
// coset-scaling tables (8 MB each at n = 2^20) are shared by all columns and stay in L2 / Infinity Cache.
#include "ntt.h"
#include "fp.cuh"
#include "../../include/cstark_conventions.h"
#include <stdlib.h>
#ifdef CS_NTT_NOMATH // measurement build (tools/build_variant_fast.py): the kernels' memory / LDS traffic without their field arithmetic
#define fp_mul(a, b) ((a) ^ (b))
#define fp_add(a, b) ((a) + (b))
#define fp_sub(a, b) ((a) - (b))
#define fp_sub_lazy(a, b) ((a) - (b))
#endif
// measurement builds only: leave out one memory phase of the v4 / v5 kernels (results are then wrong, the timing shows what the phase costs)
//   1 twiddle-table fill   2 prescale loads   4 output-factor loads   8 global stores (kept alive by an impossible condition)   16 tile loads
#ifndef CS_NTT_SKIP
#define CS_NTT_SKIP 0
#endif

namespace cs {
namespace {

// The measurement hooks, kept out of the loops: in the product build (CS_NTT_SKIP == 0) mb_load is the load, mb_store the store, and
// the MB_* conditions are compile-time false.
__device__ __forceinline__ fp mb_load(const fp *row, unsigned lane_off, unsigned fake) { return (CS_NTT_SKIP & 16) ? (fp)fake : row[lane_off]; }
__device__ __forceinline__ void mb_store(fp *row, unsigned lane_off, fp val) {
    if (!(CS_NTT_SKIP & 8) || val == 0x123456789abcdefull) row[lane_off] = val; // (measurement: kept alive by an impossible condition)
}
constexpr bool MB_NO_TWIDDLE_FILL = (CS_NTT_SKIP & 1) != 0, MB_NO_PRESCALE = (CS_NTT_SKIP & 2) != 0, MB_NO_OUTPUT_FACTOR = (CS_NTT_SKIP & 4) != 0;

constexpr int NT = 256; // threads per workgroup

__device__ __forceinline__ unsigned bitrev(unsigned x, unsigned bits) { return __brev(x) >> (32 - bits); }

// In-LDS decimation-in-frequency transform of M = 2^log_m points for L interleaved sequences
// (element (i, l) at tile[i * L + l]).  Natural order in, bit-reversed positions out.
// tw[e] = w_M^e for e < M/2.  All NT threads of the workgroup take part.
template <int L>
__device__ __forceinline__ void lds_ntt_dif(fp *tile, unsigned log_m, const fp *tw) {
    const unsigned half_total = (1u << (log_m - 1)) * L;
    for (unsigned s = 0; s < log_m; s++) {
        const unsigned hbits = log_m - 1 - s, half = 1u << hbits;
        for (unsigned b = threadIdx.x; b < half_total; b += NT) {
            const unsigned l = b % L, q = b / L;
            const unsigned j = q & (half - 1);
            const unsigned i0 = ((q >> hbits) << (hbits + 1)) + j, i1 = i0 + half;
            const fp u = tile[i0 * L + l], v = tile[i1 * L + l];
            tile[i0 * L + l] = fp_add(u, v);
            tile[i1 * L + l] = fp_mul(fp_sub_lazy(u, v), tw[j << s]);
        }
        __syncthreads();
    }
}

// grid = (C / L, width, batch).  in/out column stride n; batch stride given explicitly.
template <int L>
__global__ __launch_bounds__(NT) void k_ntt_cols(const fp *__restrict__ in, fp *__restrict__ out, unsigned log_n, unsigned log_r,
                                                const fp *__restrict__ w, const fp *__restrict__ prescale, size_t in_batch_stride,
                                                size_t out_batch_stride, size_t prescale_batch_stride) {
    extern __shared__ __attribute__((aligned(16))) fp smem[];
    const unsigned log_c = log_n - log_r, R = 1u << log_r;
    const size_t n = (size_t)1 << log_n;
    fp *tile = smem, *tw = smem + (size_t)R * L;
    const unsigned c0 = blockIdx.x * L;
    const fp *src = in + blockIdx.z * in_batch_stride + (size_t)blockIdx.y * n;
    fp *dst = out + blockIdx.z * out_batch_stride + (size_t)blockIdx.y * n;
    const fp *ps = prescale ? prescale + blockIdx.z * prescale_batch_stride : nullptr;

    for (unsigned e = threadIdx.x; e < R / 2; e += NT) tw[e] = w[(size_t)e << log_c];
    for (unsigned idx = threadIdx.x; idx < R * L; idx += NT) {
        const unsigned r = idx / L, l = idx % L;
        const size_t m = ((size_t)r << log_c) + c0 + l;
        fp v = src[m];
        if (ps) v = fp_mul(v, ps[m]);
        tile[idx] = v;
    }
    __syncthreads();
    lds_ntt_dif<L>(tile, log_r, tw);
    for (unsigned idx = threadIdx.x; idx < R * L; idx += NT) {
        const unsigned p = idx / L, l = idx % L;
        const unsigned k1 = bitrev(p, log_r), c = c0 + l;
        dst[((size_t)k1 << log_c) + c] = fp_mul(tile[idx], w[(size_t)k1 * c]);
    }
}

// grid = (R / L, width, batch).  in: rows [k1][c]; out: natural order k = k1 + R * k2.
template <int L>
__global__ __launch_bounds__(NT) void k_ntt_rows(const fp *__restrict__ in, fp *__restrict__ out, unsigned log_n, unsigned log_r,
                                                const fp *__restrict__ w, fp post_scale, int do_scale, size_t in_batch_stride,
                                                size_t out_batch_stride) {
    extern __shared__ __attribute__((aligned(16))) fp smem[];
    const unsigned log_c = log_n - log_r, C = 1u << log_c;
    const size_t n = (size_t)1 << log_n;
    fp *tile = smem, *tw = smem + (size_t)C * L;
    const unsigned k10 = blockIdx.x * L;
    const fp *src = in + blockIdx.z * in_batch_stride + (size_t)blockIdx.y * n;
    fp *dst = out + blockIdx.z * out_batch_stride + (size_t)blockIdx.y * n;

    for (unsigned e = threadIdx.x; e < C / 2; e += NT) tw[e] = w[(size_t)e << log_r];
    for (unsigned c = threadIdx.x; c < C; c += NT) { // coalesced row reads, transposed into [c][l] through registers
        fp v[L];
#pragma unroll
        for (int l = 0; l < L; l++) v[l] = src[((size_t)(k10 + l) << log_c) + c];
#pragma unroll
        for (int l = 0; l < L; l++) tile[c * L + l] = v[l];
    }
    __syncthreads();
    lds_ntt_dif<L>(tile, log_c, tw);
    for (unsigned idx = threadIdx.x; idx < C * L; idx += NT) {
        const unsigned p = idx / L, l = idx % L;
        const unsigned k2 = bitrev(p, log_c);
        fp v = tile[idx];
        if (do_scale) v = fp_mul(v, post_scale);
        dst[((size_t)k2 << log_r) + k10 + l] = v;
    }
}

// =====================================================================================================
// v2 kernels for sub-transform sizes M = 2^(LA+LB) with LA, LB <= 5: every thread runs 2^LA- and 2^LB-point
// transforms entirely in registers (compile-time twiddles, fully unrolled), with ONE exchange through LDS in
// between.  L = 16 adjacent columns / rows per workgroup make every global access a 128-byte segment.
// -----------------------------------------------------------------------------------------------------
// compile-time field arithmetic for the small twiddle tables
__host__ __device__ constexpr uint64_t cx_mul(uint64_t a, uint64_t b) {
    unsigned __int128 t = (unsigned __int128)a * b;
    uint64_t m = (uint64_t)t * 0x417fffffffffffffULL;
    uint64_t r = (uint64_t)((t + (unsigned __int128)m * FP_P) >> 64);
    return r >= FP_P ? r - FP_P : r;
}
__host__ __device__ constexpr uint64_t cx_pow(uint64_t b, uint64_t e) {
    uint64_t r = FP_ONE;
    while (e) { if (e & 1) r = cx_mul(r, b); b = cx_mul(b, b); e >>= 1; }
    return r;
}
// w_{2^log}^k (INV: its inverse); 2^55-th root of unity = generator^131 [UPSTREAM-RECALL, include/cstark_conventions.h, same as hostfield.h]
__host__ __device__ constexpr uint64_t cx_root(int log, bool inv) {
    uint64_t g = cx_pow(cx_mul(CSTARK_CONV_FIELD_GENERATOR, FP_R2), CSTARK_CONV_TWO_ADIC_ROOT_EXP);
    for (int i = log; i < 55; i++) g = cx_mul(g, g);
    return inv ? cx_pow(g, FP_P - 2) : g;
}
template <int LOG, bool INV>
struct SmallTw {
    uint64_t v[LOG == 0 ? 1 : (1 << LOG) / 2 + 1];
    constexpr SmallTw() : v{} {
        const uint64_t w = cx_root(LOG, INV);
        uint64_t x = FP_ONE;
        for (int i = 0; i < (1 << LOG) / 2 + (LOG == 0); i++) { v[i] = x; x = cx_mul(x, w); }
    }
};

// 2^LOG-point DIF transform in registers: natural order in, x[p] = X[bitrev(p)] out
template <int LOG, bool INV>
__device__ __forceinline__ void reg_ntt_dif(fp (&x)[1 << LOG]) {
    constexpr SmallTw<LOG, INV> tw{};
    constexpr int N = 1 << LOG;
#pragma unroll
    for (int s = 0; s < LOG; s++) {
        const int half = N >> (s + 1);
#pragma unroll
        for (int blk = 0; blk < (1 << s); blk++) {
#pragma unroll
            for (int j = 0; j < half; j++) {
                const int i0 = blk * 2 * half + j, i1 = i0 + half;
                const fp u = x[i0], v = x[i1];
                x[i0] = fp_add(u, v);
                x[i1] = (j == 0) ? fp_sub(u, v) : fp_mul(fp_sub_lazy(u, v), tw.v[j << s]); // the product takes a first factor < 2p
            }
        }
    }
}
__host__ __device__ constexpr unsigned cx_brev(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

#ifndef CS_NTT_L
#define CS_NTT_L 8
#endif
constexpr int L2 = CS_NTT_L; // columns / rows per workgroup in the v2 kernels
// Workgroups are dealt to the 8 XCDs round-robin by linear id, and each XCD has its own L2.  With 64-byte tiles two
// neighbouring tiles share every 128-byte line, so neighbours are given ids 8 apart: same XCD, dispatched back to back.
// (The tile counts of the v2 sizes are multiples of 16.)
__device__ __forceinline__ unsigned xcd_pair_tile(unsigned y) {
    return L2 >= 16 ? y : ((y & ~15u) | ((y & 7u) << 1) | ((y >> 3) & 1u));
}

// grid = (batch, C / L2, width): batch (coset) is the fastest grid dimension so that the workgroups re-reading the
// same coefficient tile for different cosets run close in time (the re-reads are served by L2 / Infinity Cache).
template <int LA, int LB, bool INV>
__global__ __launch_bounds__(L2 << (LA > LB ? LA : LB)) void k_ntt_cols_v2(const fp *__restrict__ in, fp *__restrict__ out, unsigned log_n,
                                                                          const fp *__restrict__ w, const fp *__restrict__ prescale,
                                                                          size_t in_batch_stride, size_t out_batch_stride,
                                                                          size_t prescale_batch_stride) {
    constexpr int A = 1 << LA, B = 1 << LB, M = A * B, LOGM = LA + LB;
    extern __shared__ __attribute__((aligned(16))) fp smem[];
    fp *tile = smem;                           // [A][B + 1][L2]
    fp *tw = smem + (size_t)A * (B + 1) * L2;  // [M] powers of w_M
    const unsigned log_c = log_n - LOGM;
    const size_t n = (size_t)1 << log_n;
    const unsigned c0 = xcd_pair_tile(blockIdx.y) * L2;
    const fp *src = in + blockIdx.x * in_batch_stride + (size_t)blockIdx.z * n;
    fp *dst = out + blockIdx.x * out_batch_stride + (size_t)blockIdx.z * n;
    const fp *ps = prescale ? prescale + blockIdx.x * prescale_batch_stride : nullptr;
    const unsigned l = threadIdx.x % L2, t = threadIdx.x / L2;

    for (unsigned e = threadIdx.x; e < M; e += blockDim.x) tw[e] = w[(size_t)e << log_c];
    if (t < B) { // step 1: A-point transforms over r1 for fixed r2 = t (rows r = r1 * B + r2)
        fp a[A];
#pragma unroll
        for (int r1 = 0; r1 < A; r1++) {
            const size_t m = ((size_t)(r1 * B + t) << log_c) + c0 + l;
            a[r1] = src[m];
        }
        if (ps) {
            // coset scaling shift^m, m = r * C + c: the row part shift^(r*C) here (M-entry slice of the power table,
            // cache resident); the column part shift^c commutes with the transform over r and is folded into the
            // output factor below
#pragma unroll
            for (int r1 = 0; r1 < A; r1++) a[r1] = fp_mul(a[r1], ps[(size_t)(r1 * B + t) << log_c]);
        }
        reg_ntt_dif<LA, INV>(a);
        __syncthreads(); // tw[] ready
#pragma unroll
        for (int p = 0; p < A; p++) {
            const unsigned k1 = cx_brev(p, LA);
            const fp v = (k1 == 0) ? a[p] : fp_mul(a[p], tw[k1 * t]);
            tile[((size_t)k1 * (B + 1) + t) * L2 + l] = v;
        }
    } else {
        __syncthreads();
    }
    __syncthreads();
    if (t < A) { // step 2: B-point transforms over r2 for fixed k1 = t; output row k = k1 + A * k2
        fp b[B];
#pragma unroll
        for (int r2 = 0; r2 < B; r2++) b[r2] = tile[((size_t)t * (B + 1) + r2) * L2 + l];
        reg_ntt_dif<LB, INV>(b);
        const unsigned c = c0 + l;
        // output factor shift^c * w_n^(k*c), k = t + A*k2: a geometric sequence in k2 with ratio w_n^(A*c), generated
        // in registers instead of gathering 8-byte entries of the 8 MB twiddle table
        fp g = w[(size_t)t * c];
        if (ps) g = fp_mul(g, ps[c]);
        const fp ratio = w[(size_t)A * c];
#pragma unroll
        for (int k2 = 0; k2 < B; k2++) { // b[] holds the outputs in bit-reversed positions
            dst[((size_t)(t + A * k2) << log_c) + c] = fp_mul(b[cx_brev(k2, LB)], g);
            if (k2 + 1 < B) g = fp_mul(g, ratio);
        }
    }
}

// grid = (batch, R / L2, width).  in: rows [k1][c] (each row M contiguous); out: natural order k = k1 + R * k2.
template <int LA, int LB, bool INV>
__global__ __launch_bounds__(L2 << (LA > LB ? LA : LB)) void k_ntt_rows_v2(const fp *__restrict__ in, fp *__restrict__ out, unsigned log_n,
                                                                          const fp *__restrict__ w, fp post_scale, int do_scale,
                                                                          size_t in_batch_stride, size_t out_batch_stride) {
    constexpr int A = 1 << LA, B = 1 << LB, M = A * B, LOGM = LA + LB;
    extern __shared__ __attribute__((aligned(16))) fp smem[];
    fp *tile = smem;                     // [L2][A][B] with the B index XOR-swizzled
    fp *tw = smem + (size_t)L2 * A * B;  // [M]
    const unsigned log_r = log_n - LOGM;
    const size_t n = (size_t)1 << log_n;
    const unsigned k10 = xcd_pair_tile(blockIdx.y) * L2;
    const fp *src = in + blockIdx.x * in_batch_stride + (size_t)blockIdx.z * n;
    fp *dst = out + blockIdx.x * out_batch_stride + (size_t)blockIdx.z * n;

    for (unsigned e = threadIdx.x; e < M; e += blockDim.x) tw[e] = w[(size_t)e << log_r];
    {   // step 1: task (l, c2) with c2 fastest across lanes (coalesced row reads): A-point transforms over c1
        const unsigned c2 = threadIdx.x % B, l = threadIdx.x / B;
        if (l < L2) {
            fp a[A];
#pragma unroll
            for (int c1 = 0; c1 < A; c1++) a[c1] = src[((size_t)(k10 + l) << LOGM) + c1 * B + c2];
            reg_ntt_dif<LA, INV>(a);
            __syncthreads();
#pragma unroll
            for (int p = 0; p < A; p++) {
                const unsigned j1 = cx_brev(p, LA);
                const fp v = (j1 == 0) ? a[p] : fp_mul(a[p], tw[j1 * c2]);
                const unsigned sw = (l | ((j1 % (32 / L2)) * L2)) & (B - 1);
                tile[((size_t)l * A + j1) * B + (c2 ^ sw)] = v;
            }
        } else {
            __syncthreads();
        }
    }
    __syncthreads();
    {   // step 2: task (j1, l) with l fastest across lanes (128-byte transposed stores): B-point transforms over c2
        const unsigned l = threadIdx.x % L2, j1 = threadIdx.x / L2;
        if (j1 < A) {
            fp b[B];
            const unsigned sw = (l | ((j1 % (32 / L2)) * L2)) & (B - 1);
#pragma unroll
            for (int c2 = 0; c2 < B; c2++) b[c2] = tile[((size_t)l * A + j1) * B + (c2 ^ sw)];
            reg_ntt_dif<LB, INV>(b);
#pragma unroll
            for (int p = 0; p < B; p++) {
                const unsigned k2 = j1 + A * cx_brev(p, LB);
                fp v = b[p];
                if (do_scale) v = fp_mul(v, post_scale);
                dst[((size_t)k2 << log_r) + k10 + l] = v;
            }
        }
    }
}

// =====================================================================================================
// v4 kernels: the same two passes with every 2^LOGM-point sub-transform in THREE register steps (2^LA, 2^LB, 2^LC points,
// LA >= LB >= LC) instead of two.  A thread then holds 2^LA = 16 elements instead of 32, a workgroup of the same tile
// (L2 columns x 2^LOGM points, 64 KB of LDS) has twice the threads, and a CU holds four waves per SIMD instead of two:
// the v2 kernels spent half of their wave cycles waiting (rocprofv3: SQ_WAIT_ANY 22-39 %, SQ_WAIT_INST_ANY 16-30 % of
// SQ_WAVE_CYCLES) because two waves per SIMD cannot cover global-load latency, LDS round trips and the issue gaps of
// dependent v_mad_u64_u32 chains.
//
// Index algebra (w = w_M, M = 2^LOGM, T = 2^(LB+LC)):  r = r1 T + r2 2^LC + r3,  k = k1 + 2^LA k2 + 2^(LA+LB) k3
//   step 1  (r2, r3) fixed: Y[k1]  = sum_r1 x[r] w_{2^LA}^(r1 k1),  then  * w^((r2 2^LC + r3) k1)
//   step 2  (k1, r3) fixed: Z[k2]  = sum_r2 Y[k1; r2, r3] w_{2^LB}^(r2 k2),  then  * w^(2^LA r3 k2)
//   step 3  (k1, k2) fixed: X[k3]  = sum_r3 Z[k1, k2; r3] w_{2^LC}^(r3 k3)
// LDS tile: element (k1, q, l) at (k1 T + q) L2 + l (columns kernel) resp. ((k1 L2 + l)(T + 4) + q) (rows kernel), q = r2 2^LC + r3
// after step 1; step 2 works in place and stores Z[k2] of task (k1, r3) at q = k2 2^LC + (r3 ^ (k2 mod 2^LC)) so that step 3's
// reads (fixed r3, lanes over k2) fall into distinct banks.
// -----------------------------------------------------------------------------------------------------
template <int LA, int LB, int LC>
struct V4 {
    static constexpr int A = 1 << LA, Bn = 1 << LB, Cn = 1 << LC, T = Bn * Cn, M = A * T, LOGM = LA + LB + LC;
    static constexpr int NT = L2 * T;       // threads per workgroup: one step-1 task each
    static constexpr int J2 = A / Bn;       // step-2 tasks (k1, r3) per thread
    static constexpr int J3 = A / Cn;       // step-3 tasks (k1, k2) per thread
    static_assert(LA >= LB && LB >= LC && LC >= 1, "step sizes must not increase");
};

#ifndef CS_NTT_TILES
#define CS_NTT_TILES 1
#endif
// Tiles per workgroup; with more than one the next tile's global loads are issued before the current tile's arithmetic.  Measured on
// MI355X (2^20 x 94 x 8): the register prefetch pushes hipcc past the 128 VGPRs that four waves per SIMD allow (15-34 spilled
// registers even behind scheduling barriers) and the extension takes 11.2 instead of 8.6 ms; the default stays 1.  Also measured and
// dropped: steps 2 and 3 task by task inside one wave (the 64 lanes of a wave are the L2 columns x the 8 tasks of one k1, so no
// workgroup barrier is needed after step 1) -- 9.2 ms: the per-task LDS round trips serialise; all cosets of a tile in one
// workgroup with the coefficients kept in registers -- 83 spilled registers.
constexpr int V4_TILES = CS_NTT_TILES;

// Steps 2 and 3 of the column pass on the tile in LDS (after step 1 stored Y * twiddle), including the output factor and the
// stores.  t, l: this thread's position; kb-dependent factors from w / ps as in the v2 kernel.
template <int LA, int LB, int LC, bool INV>
__device__ __forceinline__ void cols_v4_finish(fp *tile, const fp *tw, const fp *__restrict__ w, const fp *__restrict__ ps, fp *__restrict__ dst,
                                               unsigned log_c, unsigned c, unsigned t, unsigned l, const fp *__restrict__ outf,
                                               const fp *__restrict__ ratio_tab) {
    using G = V4<LA, LB, LC>;
    constexpr int A = G::A, Bn = G::Bn, Cn = G::Cn, T = G::T;
    fp z[G::J2][Bn];
#pragma unroll
    for (int j = 0; j < G::J2; j++) { // step 2: task u = t + T j = k1 Cn + r3
        const unsigned u = t + T * j, k1 = u / Cn, r3 = u % Cn;
#pragma unroll
        for (int r2 = 0; r2 < Bn; r2++) z[j][r2] = tile[((size_t)k1 * T + r2 * Cn + r3) * L2 + l];
    }
    __syncthreads(); // every input of step 2 is in registers: the tile can be overwritten
#pragma unroll
    for (int j = 0; j < G::J2; j++) {
        const unsigned u = t + T * j, k1 = u / Cn, r3 = u % Cn;
        reg_ntt_dif<LB, INV>(z[j]);
#pragma unroll
        for (int p = 0; p < Bn; p++) {
            const unsigned k2 = cx_brev(p, LB);
            const fp v = (k2 == 0) ? z[j][p] : fp_mul(z[j][p], tw[(A * k2) * r3]);
            tile[((size_t)k1 * T + k2 * Cn + (r3 ^ (k2 % Cn))) * L2 + l] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < G::J3; j++) { // step 3: task v = t + T j = k1 Bn + k2; output rows k = k1 + A k2 + A Bn k3
        const unsigned v = t + T * j, k1 = v / Bn, k2 = v % Bn;
        fp x[Cn];
#pragma unroll
        for (int r3 = 0; r3 < Cn; r3++) x[r3] = tile[((size_t)k1 * T + k2 * Cn + (r3 ^ (k2 % Cn))) * L2 + l];
        reg_ntt_dif<LC, INV>(x);
        // output factor shift^c w_n^(k c), k = (k1 + A k2) + A Bn k3: a geometric sequence in k3 with ratio w_n^(A Bn c)
        const unsigned kb = k1 + A * k2;
        fp g, ratio;
        if (MB_NO_OUTPUT_FACTOR) { g = kb + c; ratio = c; }
        else if (outf) { // compact tables (ntt_build_aux_*): one 64-byte segment per (kb, tile) instead of 8 scattered sectors
            g = outf[((size_t)kb << log_c) + c];
            ratio = ratio_tab[c];
        } else {
            g = w[(size_t)kb * c];
            if (ps) g = fp_mul(g, ps[c]);
            ratio = w[(size_t)(A * Bn) * c];
        }
        const unsigned lane_off = (kb << log_c) + c; // uniform row base + 32-bit lane offset: one address register for all stores
#pragma unroll
        for (int k3 = 0; k3 < Cn; k3++) {
            fp *row = dst + ((size_t)(A * Bn * k3) << log_c);
            const fp val = fp_mul(x[cx_brev(k3, LC)], g);
            mb_store(row, lane_off, val);
            if (k3 + 1 < Cn) g = fp_mul(g, ratio);
        }
    }
}

// grid = (C / L2 / V4_TILES, batch, width)
template <int LA, int LB, int LC, bool INV>
__global__ __launch_bounds__((V4<LA, LB, LC>::NT), 4) void k_ntt_cols_v4(const fp *__restrict__ in, fp *__restrict__ out, unsigned log_n,
                                                                        const fp *__restrict__ w, const fp *__restrict__ prescale,
                                                                        size_t in_batch_stride, size_t out_batch_stride, size_t prescale_batch_stride,
                                                                        const fp *__restrict__ aux, const fp *__restrict__ aux_ps,
                                                                        size_t aux_ps_batch_stride) {
    using G = V4<LA, LB, LC>;
    constexpr int A = G::A, T = G::T, M = G::M, LOGM = G::LOGM;
    extern __shared__ __attribute__((aligned(16))) fp smem[];
    fp *tile = smem;                      // [A][T][L2]
    fp *tw = smem + (size_t)M * L2;       // [M] powers of w_M
    const unsigned log_c = log_n - LOGM;
    const size_t n = (size_t)1 << log_n;
    const unsigned bz = blockIdx.y, tile_id = blockIdx.x; // batch (coset), tile: see the grid note at k_ntt_cols_v5
    const fp *src = in + bz * in_batch_stride + (size_t)blockIdx.z * n;
    fp *dst = out + bz * out_batch_stride + (size_t)blockIdx.z * n;
    const fp *ps = prescale ? prescale + bz * prescale_batch_stride : nullptr;
    const unsigned l = threadIdx.x % L2, t = threadIdx.x / L2;
    // compact tables (NttAux): aux = [M] twiddles of this pass | [C] twiddles of the row pass | [C] ratios | [A Bn][C] output factors;
    // aux_ps (per batch) = [M] row part of the prescale | [A Bn][C] output factors times the column part
    const fp *ps_row = (ps && aux_ps) ? aux_ps + bz * aux_ps_batch_stride : nullptr;
    const fp *outf = !aux ? nullptr : ps ? (ps_row ? ps_row + M : nullptr) : aux + M + ((size_t)2 << log_c);
    const fp *ratio_tab = aux ? aux + M + ((size_t)1 << log_c) : nullptr;

    if (MB_NO_TWIDDLE_FILL) { for (unsigned e = threadIdx.x; e < M; e += G::NT) tw[e] = e; }
    else if (aux) for (unsigned e = threadIdx.x; e < M; e += G::NT) tw[e] = aux[e];
    else for (unsigned e = threadIdx.x; e < M; e += G::NT) tw[e] = w[(size_t)e << log_c];
    fp nxt[A];
    {
        const unsigned lane_off = (t << log_c) + xcd_pair_tile(tile_id * V4_TILES) * L2 + l;
#pragma unroll
        for (int r1 = 0; r1 < A; r1++) nxt[r1] = mb_load(src + ((size_t)(r1 * T) << log_c), lane_off, lane_off + r1);
    }
    __syncthreads(); // tw[] ready
#pragma unroll 1
    for (int it = 0; it < V4_TILES; it++) {
        const unsigned c = xcd_pair_tile(tile_id * V4_TILES + it) * L2 + l;
        {   // step 1: rows r = r1 T + t
            fp a[A];
#pragma unroll
            for (int r1 = 0; r1 < A; r1++) a[r1] = nxt[r1];
            if (it + 1 < V4_TILES) {
                const unsigned lane_off = (t << log_c) + xcd_pair_tile(tile_id * V4_TILES + it + 1) * L2 + l;
#pragma unroll
                for (int r1 = 0; r1 < A; r1++) nxt[r1] = (src + ((size_t)(r1 * T) << log_c))[lane_off];
            }
            __builtin_amdgcn_sched_barrier(0); // the prefetch is issued here, ahead of this tile's arithmetic
            if (MB_NO_PRESCALE) {
            } else if (ps_row) { // row part shift^(r C) of the coset power; the column part shift^c is folded into the output factor
#pragma unroll
                for (int r1 = 0; r1 < A; r1++) a[r1] = fp_mul(a[r1], ps_row[r1 * T + t]);
            } else if (ps) {
                const unsigned ps_off = t << log_c;
#pragma unroll
                for (int r1 = 0; r1 < A; r1++) a[r1] = fp_mul(a[r1], (ps + ((size_t)(r1 * T) << log_c))[ps_off]);
            }
            reg_ntt_dif<LA, INV>(a);
#pragma unroll
            for (int p = 0; p < A; p++) {
                const unsigned k1 = cx_brev(p, LA);
                tile[((size_t)k1 * T + t) * L2 + l] = (k1 == 0) ? a[p] : fp_mul(a[p], tw[k1 * t]);
            }
        }
        __syncthreads();
        cols_v4_finish<LA, LB, LC, INV>(tile, tw, w, ps, dst, log_c, c, t, l, outf, ratio_tab);
        __syncthreads(); // the tile is rewritten by the next iteration
    }
}

// grid = (batch, R / L2 / V4_TILES, width).  in: rows [k1][c] (each row M contiguous); out: natural order k = k1 + R * k2.
template <int LA, int LB, int LC, bool INV>
__global__ __launch_bounds__((V4<LA, LB, LC>::NT), 4) void k_ntt_rows_v4(const fp *__restrict__ in, fp *__restrict__ out, unsigned log_n,
                                                                        const fp *__restrict__ w, fp post_scale, int do_scale,
                                                                        size_t in_batch_stride, size_t out_batch_stride, const fp *__restrict__ aux_tw) {
    using G = V4<LA, LB, LC>;
    constexpr int A = G::A, Bn = G::Bn, Cn = G::Cn, T = G::T, M = G::M, LOGM = G::LOGM;
    constexpr int TP = T + 4; // padded run of q per (j1, l): lanes over l then hit distinct banks (T + 4 = 4 mod 32 for T = 32, 64)
    extern __shared__ __attribute__((aligned(16))) fp smem[];
    fp *tile = smem;                              // [A][L2][TP]
    fp *tw = smem + (size_t)A * L2 * TP;          // [M]
    const unsigned log_r = log_n - LOGM;
    const size_t n = (size_t)1 << log_n;
    const fp *src = in + blockIdx.x * in_batch_stride + (size_t)blockIdx.z * n;
    fp *dst = out + blockIdx.x * out_batch_stride + (size_t)blockIdx.z * n;
    const unsigned s1 = threadIdx.x % T, l1 = threadIdx.x / T; // step 1: s = c2 Cn + c3 fastest across lanes (coalesced row reads)
    const unsigned l = threadIdx.x % L2, t = threadIdx.x / L2; // steps 2 and 3: l fastest across lanes (64-byte transposed stores)

    if (MB_NO_TWIDDLE_FILL) { for (unsigned e = threadIdx.x; e < M; e += G::NT) tw[e] = e; }
    else if (aux_tw) for (unsigned e = threadIdx.x; e < M; e += G::NT) tw[e] = aux_tw[e];
    else for (unsigned e = threadIdx.x; e < M; e += G::NT) tw[e] = w[(size_t)e << log_r];
    fp nxt[A];
    const unsigned in_lane = (l1 << LOGM) + s1; // uniform row base + 32-bit lane offset: one address register for all loads
    {
        const unsigned k10 = xcd_pair_tile(blockIdx.y * V4_TILES) * L2;
#pragma unroll
        for (int c1 = 0; c1 < A; c1++) nxt[c1] = mb_load(src + ((size_t)k10 << LOGM) + c1 * T, in_lane, in_lane + c1);
    }
    __syncthreads(); // tw[] ready
#pragma unroll 1
    for (int it = 0; it < V4_TILES; it++) {
        const unsigned k10 = xcd_pair_tile(blockIdx.y * V4_TILES + it) * L2;
        {   // step 1: columns c = c1 T + s
            fp a[A];
#pragma unroll
            for (int c1 = 0; c1 < A; c1++) a[c1] = nxt[c1];
            if (it + 1 < V4_TILES) {
                const unsigned k11 = xcd_pair_tile(blockIdx.y * V4_TILES + it + 1) * L2;
#pragma unroll
                for (int c1 = 0; c1 < A; c1++) nxt[c1] = (src + ((size_t)k11 << LOGM) + c1 * T)[in_lane];
            }
            __builtin_amdgcn_sched_barrier(0); // the prefetch is issued here, ahead of this tile's arithmetic
            reg_ntt_dif<LA, INV>(a);
#pragma unroll
            for (int p = 0; p < A; p++) {
                const unsigned j1 = cx_brev(p, LA);
                tile[((size_t)j1 * L2 + l1) * TP + s1] = (j1 == 0) ? a[p] : fp_mul(a[p], tw[j1 * s1]);
            }
        }
        __syncthreads();
        fp z[G::J2][Bn];
#pragma unroll
        for (int j = 0; j < G::J2; j++) { // step 2: task u = t + T j = j1 Cn + c3
            const unsigned u = t + T * j, j1 = u / Cn, c3 = u % Cn;
#pragma unroll
            for (int c2 = 0; c2 < Bn; c2++) z[j][c2] = tile[((size_t)j1 * L2 + l) * TP + c2 * Cn + c3];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < G::J2; j++) {
            const unsigned u = t + T * j, j1 = u / Cn, c3 = u % Cn;
            reg_ntt_dif<LB, INV>(z[j]);
#pragma unroll
            for (int p = 0; p < Bn; p++) {
                const unsigned j2 = cx_brev(p, LB);
                const fp v = (j2 == 0) ? z[j][p] : fp_mul(z[j][p], tw[(A * j2) * c3]);
                tile[((size_t)j1 * L2 + l) * TP + j2 * Cn + (c3 ^ (j2 % Cn))] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < G::J3; j++) { // step 3: task v = t + T j = j1 Bn + j2; output frequency k2 = j1 + A j2 + A Bn j3
            const unsigned v = t + T * j, j1 = v / Bn, j2 = v % Bn;
            fp x[Cn];
#pragma unroll
            for (int c3 = 0; c3 < Cn; c3++) x[c3] = tile[((size_t)j1 * L2 + l) * TP + j2 * Cn + (c3 ^ (j2 % Cn))];
            reg_ntt_dif<LC, INV>(x);
            const unsigned out_lane = ((j1 + A * j2) << log_r) + l;
#pragma unroll
            for (int p = 0; p < Cn; p++) {
                fp val = x[p];
                if (do_scale) val = fp_mul(val, post_scale);
                mb_store(dst + ((size_t)(A * Bn * cx_brev(p, LC)) << log_r) + k10, out_lane, val);
            }
        }
        __syncthreads(); // the tile is rewritten by the next iteration
    }
}

// =====================================================================================================
// v5 kernels: the v4 passes for shapes with two step-2 and two step-3 tasks per thread (2^LA = 2 * 2^LB = 2 * 2^LC: the
// 16 * 8 * 8 split of 1024 points), with each of the two exchanges done in TWO HALVES through a tile of half the size.  The tasks
// of a thread are task h = 0, 1 with k1 in [8 h, 8 h + 8): step 1 writes its eight outputs of half h, every thread reads the
// inputs of its task h, then the same buffer takes the other half.  32 KB of tile + 8 KB of twiddles per workgroup instead of
// 72: LDS no longer limits residency (registers do: three workgroups per CU in the row pass, two in the column pass), so a CU holds workgroups in
// different phases -- the loads and stores of one run under the arithmetic of the others.  (v4 measured with its arithmetic
// alone 5.55 ms, with its memory traffic alone 5.74 ms, together 8.0 ms for 94 columns x 8 cosets: two resident workgroups
// overlap too little.)  Register need after the cheaper field product: 62 (rows) / 95 (columns) VGPRs.
// -----------------------------------------------------------------------------------------------------
#ifndef CS_NTT_V5_COLS_WAVES
#define CS_NTT_V5_COLS_WAVES 5 // 95 VGPRs, no spills, two workgroups per CU; 6 (80 VGPRs, 8 spilled -> 30 % more HBM writes): LDE 7.80 vs 7.58 ms
#endif
#ifndef CS_NTT_V5_ROWS_WAVES
#define CS_NTT_V5_ROWS_WAVES 6
#endif
#ifndef CS_NTT_V5_ROWS_TILES
#define CS_NTT_V5_ROWS_TILES 1
#endif
// Tiles per workgroup of the row pass; > 1: the next tile's loads are issued once the first exchange has freed the data registers and
// fly behind steps 2 and 3.  Measured (94 columns x 8 cosets): 1 tile, three workgroups per CU 7.85 ms; 4 tiles at four waves per
// SIMD (128 VGPRs, 2 spilled) 8.27 ms; 2 tiles (16 spilled) 8.77 ms -- more resident workgroups beat the software prefetch again.
constexpr int V5_ROWS_TILES = CS_NTT_V5_ROWS_TILES;
template <int LA, int LB, int LC, bool INV>
__global__ __launch_bounds__((V4<LA, LB, LC>::NT), CS_NTT_V5_COLS_WAVES) void k_ntt_cols_v5(
    const fp *__restrict__ in, fp *__restrict__ out, unsigned log_n, const fp *__restrict__ w, const fp *__restrict__ prescale, size_t in_batch_stride,
    size_t out_batch_stride, size_t prescale_batch_stride, const fp *__restrict__ aux, const fp *__restrict__ aux_ps, size_t aux_ps_batch_stride) {
    using G = V4<LA, LB, LC>;
    constexpr int A = G::A, Bn = G::Bn, Cn = G::Cn, T = G::T, M = G::M, LOGM = G::LOGM, AH = A / 2;
    static_assert(G::J2 == 2 && G::J3 == 2 && Bn == AH && Cn == AH, "two tasks per thread and step, one per half");
    extern __shared__ __attribute__((aligned(16))) fp smem[];
    fp *tile = smem;                        // [A / 2][T][L2]
    fp *tw = smem + (size_t)(M / 2) * L2;   // [M] powers of w_M
    const unsigned log_c = log_n - LOGM;
    const size_t n = (size_t)1 << log_n;
    // grid = (C / L2, batch, width): the TILE is the fastest grid dimension.  Workgroups go to the 8 XCDs round-robin by linear id, so
    // the XCD of a workgroup is its tile index mod 8 whatever its coset: the eight cosets of a tile -- which read the SAME coefficients
    // -- share one L2, and the coefficient table leaves HBM once per extension instead of once per coset (round 2: batch fastest put
    // coset k on XCD k, and the wide tables were extended coset by coset, 3 GB of other traffic between two reads of a tile).
    const unsigned bz = blockIdx.y;
    const fp *src = in + bz * in_batch_stride + (size_t)blockIdx.z * n;
    fp *dst = out + bz * out_batch_stride + (size_t)blockIdx.z * n;
    const fp *ps = prescale ? prescale + bz * prescale_batch_stride : nullptr;
    const unsigned l = threadIdx.x % L2, t = threadIdx.x / L2;
    const fp *ps_row = (ps && aux_ps) ? aux_ps + bz * aux_ps_batch_stride : nullptr; // tables as in the v4 kernel
    const fp *outf = !aux ? nullptr : ps ? (ps_row ? ps_row + M : nullptr) : aux + M + ((size_t)2 << log_c);
    const fp *ratio_tab = aux ? aux + M + ((size_t)1 << log_c) : nullptr;
    const unsigned c = xcd_pair_tile(blockIdx.x) * L2 + l;

    if (MB_NO_TWIDDLE_FILL) { for (unsigned e = threadIdx.x; e < M; e += G::NT) tw[e] = e; }
    else if (aux) for (unsigned e = threadIdx.x; e < M; e += G::NT) tw[e] = aux[e];
    else for (unsigned e = threadIdx.x; e < M; e += G::NT) tw[e] = w[(size_t)e << log_c];
    fp a[A];
    {
        const unsigned lane_off = (t << log_c) + c;
#pragma unroll
        for (int r1 = 0; r1 < A; r1++) a[r1] = mb_load(src + ((size_t)(r1 * T) << log_c), lane_off, lane_off + r1);
    }
    if (MB_NO_PRESCALE) {
    } else if (ps_row) { // row part shift^(r C) of the coset power; the column part shift^c is folded into the output factor
#pragma unroll
        for (int r1 = 0; r1 < A; r1++) a[r1] = fp_mul(a[r1], ps_row[r1 * T + t]);
    } else if (ps) {
        const unsigned ps_off = t << log_c;
#pragma unroll
        for (int r1 = 0; r1 < A; r1++) a[r1] = fp_mul(a[r1], (ps + ((size_t)(r1 * T) << log_c))[ps_off]);
    }
    reg_ntt_dif<LA, INV>(a); // step 1: rows r = r1 T + t; a[p] = Y[k1 = brev(p)]
    __syncthreads(); // tw[] ready
    const unsigned kh = t / Cn, r3 = t % Cn; // task h of this thread in step 2: k1 = kh + AH h, r3
    fp z[2][Bn];
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int p = 0; p < A; p++) {
            const unsigned k1 = cx_brev(p, LA);
            if ((int)(k1 / AH) == h) tile[((size_t)(k1 % AH) * T + t) * L2 + l] = (k1 == 0) ? a[p] : fp_mul(a[p], tw[k1 * t]);
        }
        __syncthreads();
#pragma unroll
        for (int r2 = 0; r2 < Bn; r2++) z[h][r2] = tile[((size_t)kh * T + r2 * Cn + r3) * L2 + l];
        __syncthreads(); // the half is rewritten
    }
    const unsigned k1s = t / Bn, k2 = t % Bn; // task h in step 3: k1 = k1s + AH h, k2
#pragma unroll
    for (int h = 0; h < 2; h++) {
        reg_ntt_dif<LB, INV>(z[h]);
#pragma unroll
        for (int p = 0; p < Bn; p++) {
            const unsigned j2 = cx_brev(p, LB);
            const fp v = (j2 == 0) ? z[h][p] : fp_mul(z[h][p], tw[(A * j2) * r3]);
            tile[((size_t)kh * T + j2 * Cn + (r3 ^ (j2 % Cn))) * L2 + l] = v;
        }
        __syncthreads();
        fp x[Cn];
#pragma unroll
        for (int q = 0; q < Cn; q++) x[q] = tile[((size_t)k1s * T + k2 * Cn + (q ^ (k2 % Cn))) * L2 + l];
        if (h == 0) __syncthreads(); // the half is rewritten by task 1
        reg_ntt_dif<LC, INV>(x);
        // output factor shift^c w_n^(k c), k = (k1 + A k2) + A Bn k3: a geometric sequence in k3 with ratio w_n^(A Bn c)
        const unsigned kb = (k1s + AH * h) + A * k2;
        fp g, ratio;
        if (MB_NO_OUTPUT_FACTOR) { g = kb + c; ratio = c; }
        else if (outf) {
            g = outf[((size_t)kb << log_c) + c];
            ratio = ratio_tab[c];
        } else {
            g = w[(size_t)kb * c];
            if (ps) g = fp_mul(g, ps[c]);
            ratio = w[(size_t)(A * Bn) * c];
        }
        const unsigned lane_off = (kb << log_c) + c;
#pragma unroll
        for (int k3 = 0; k3 < Cn; k3++) {
            fp *row = dst + ((size_t)(A * Bn * k3) << log_c);
            const fp val = fp_mul(x[cx_brev(k3, LC)], g);
            mb_store(row, lane_off, val);
            if (k3 + 1 < Cn) g = fp_mul(g, ratio);
        }
    }
}

template <int LA, int LB, int LC, bool INV>
__global__ __launch_bounds__((V4<LA, LB, LC>::NT), CS_NTT_V5_ROWS_WAVES) void k_ntt_rows_v5(const fp *__restrict__ in, fp *__restrict__ out, unsigned log_n,
                                                                                           const fp *__restrict__ w, fp post_scale, int do_scale,
                                                                                           size_t in_batch_stride, size_t out_batch_stride,
                                                                                           const fp *__restrict__ aux_tw) {
    using G = V4<LA, LB, LC>;
    constexpr int A = G::A, Bn = G::Bn, Cn = G::Cn, T = G::T, M = G::M, LOGM = G::LOGM, AH = A / 2;
    static_assert(G::J2 == 2 && G::J3 == 2 && Bn == AH && Cn == AH, "two tasks per thread and step, one per half");
    constexpr int TP = T + 4; // padded run of q per (j1, l), as in the v4 kernel
    extern __shared__ __attribute__((aligned(16))) fp smem[];
    fp *tile = smem;                              // [A / 2][L2][TP]
    fp *tw = smem + (size_t)AH * L2 * TP;         // [M]
    const unsigned log_r = log_n - LOGM;
    const size_t n = (size_t)1 << log_n;
    const fp *src = in + blockIdx.x * in_batch_stride + (size_t)blockIdx.z * n;
    fp *dst = out + blockIdx.x * out_batch_stride + (size_t)blockIdx.z * n;
    const unsigned s1 = threadIdx.x % T, l1 = threadIdx.x / T; // step 1: s = c2 Cn + c3 fastest across lanes (coalesced row reads)
    const unsigned l = threadIdx.x % L2, t = threadIdx.x / L2; // steps 2 and 3: l fastest across lanes (64-byte transposed stores)
    if (MB_NO_TWIDDLE_FILL) { for (unsigned e = threadIdx.x; e < M; e += G::NT) tw[e] = e; }
    else if (aux_tw) for (unsigned e = threadIdx.x; e < M; e += G::NT) tw[e] = aux_tw[e];
    else for (unsigned e = threadIdx.x; e < M; e += G::NT) tw[e] = w[(size_t)e << log_r];
    fp a[A];
    const unsigned in_lane = (l1 << LOGM) + s1;
    {
        const unsigned k10 = xcd_pair_tile(blockIdx.y * V5_ROWS_TILES) * L2;
#pragma unroll
        for (int c1 = 0; c1 < A; c1++) a[c1] = mb_load(src + ((size_t)k10 << LOGM) + c1 * T, in_lane, in_lane + c1);
    }
    const unsigned jh = t / Cn, c3 = t % Cn;
    const unsigned j1s = t / Bn, j2s = t % Bn;
#pragma unroll 1
    for (int it = 0; it < V5_ROWS_TILES; it++) {
        const unsigned k10 = xcd_pair_tile(blockIdx.y * V5_ROWS_TILES + it) * L2;
        reg_ntt_dif<LA, INV>(a); // step 1: columns c = c1 T + s
        __syncthreads(); // tw[] ready / the previous tile's last reads are done
        fp z[2][Bn];
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int p = 0; p < A; p++) {
                const unsigned j1 = cx_brev(p, LA);
                if ((int)(j1 / AH) == h) tile[((size_t)(j1 % AH) * L2 + l1) * TP + s1] = (j1 == 0) ? a[p] : fp_mul(a[p], tw[j1 * s1]);
            }
            __syncthreads();
#pragma unroll
            for (int c2 = 0; c2 < Bn; c2++) z[h][c2] = tile[((size_t)jh * L2 + l) * TP + c2 * Cn + c3];
            __syncthreads();
        }
        if (it + 1 < V5_ROWS_TILES) { // a[] is dead: the next tile's loads fly behind steps 2 and 3 of this one
            const unsigned k11 = xcd_pair_tile(blockIdx.y * V5_ROWS_TILES + it + 1) * L2;
#pragma unroll
            for (int c1 = 0; c1 < A; c1++) a[c1] = (src + ((size_t)k11 << LOGM) + c1 * T)[in_lane];
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            reg_ntt_dif<LB, INV>(z[h]);
#pragma unroll
            for (int p = 0; p < Bn; p++) {
                const unsigned j2 = cx_brev(p, LB);
                const fp v = (j2 == 0) ? z[h][p] : fp_mul(z[h][p], tw[(A * j2) * c3]);
                tile[((size_t)jh * L2 + l) * TP + j2 * Cn + (c3 ^ (j2 % Cn))] = v;
            }
            __syncthreads();
            fp x[Cn];
#pragma unroll
            for (int q = 0; q < Cn; q++) x[q] = tile[((size_t)j1s * L2 + l) * TP + j2s * Cn + (q ^ (j2s % Cn))];
            if (h == 0) __syncthreads();
            reg_ntt_dif<LC, INV>(x);
            const unsigned out_lane = (((j1s + AH * h) + A * j2s) << log_r) + l; // output frequency k2 = j1 + A j2 + A Bn j3
#pragma unroll
            for (int p = 0; p < Cn; p++) {
                fp val = x[p];
                if (do_scale) val = fp_mul(val, post_scale);
                mb_store(dst + ((size_t)(A * Bn * cx_brev(p, LC)) << log_r) + k10, out_lane, val);
            }
        }
    }
}

template <int RA, int RB, int RC, int CA, int CB, int CC, bool INV>
hipError_t launch_v5(const NttArgs &a, hipStream_t stream) {
    using GR = V4<RA, RB, RC>;
    using GC = V4<CA, CB, CC>;
    const size_t lds_a = ((size_t)GR::M / 2 * L2 + GR::M) * sizeof(fp);
    const size_t lds_b = ((size_t)GC::A / 2 * L2 * (GC::T + 4) + GC::M) * sizeof(fp);
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void *)k_ntt_rows_v5<CA, CB, CC, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void *)k_ntt_cols_v5<RA, RB, RC, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a)) != hipSuccess) return e;
    hipLaunchKernelGGL((k_ntt_cols_v5<RA, RB, RC, INV>), dim3((unsigned)GC::M / L2, a.batch, a.width), dim3(GR::NT), lds_a, stream, a.in, a.scratch, a.log_n,
                       a.w, a.prescale, a.in_batch_stride, a.scratch_batch_stride, a.prescale_batch_stride, a.aux, a.prescale ? a.aux_ps : nullptr,
                       a.aux_ps_batch_stride);
    static_assert((GR::M / L2) % V5_ROWS_TILES == 0, "tiles per workgroup");
    hipLaunchKernelGGL((k_ntt_rows_v5<CA, CB, CC, INV>), dim3(a.batch, (unsigned)GR::M / L2 / V5_ROWS_TILES, a.width), dim3(GC::NT), lds_b, stream, (const fp *)a.scratch,
                       a.out, a.log_n, a.w, a.post_scale, a.do_scale ? 1 : 0, a.scratch_batch_stride, a.out_batch_stride,
                       a.aux ? a.aux + GR::M : nullptr);
    return hipGetLastError();
}

template <int RA, int RB, int RC, int CA, int CB, int CC, bool INV>
hipError_t launch_v4(const NttArgs &a, hipStream_t stream) {
    using GR = V4<RA, RB, RC>;
    using GC = V4<CA, CB, CC>;
    const size_t lds_a = ((size_t)GR::M * L2 + GR::M) * sizeof(fp);
    const size_t lds_b = ((size_t)GC::A * L2 * (GC::T + 4) + GC::M) * sizeof(fp);
    static_assert((GC::M / L2) % V4_TILES == 0 && (GR::M / L2) % V4_TILES == 0, "tiles per workgroup");
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void *)k_ntt_rows_v4<CA, CB, CC, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void *)k_ntt_cols_v4<RA, RB, RC, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a)) != hipSuccess) return e;
    hipLaunchKernelGGL((k_ntt_cols_v4<RA, RB, RC, INV>), dim3((unsigned)GC::M / L2 / V4_TILES, a.batch, a.width), dim3(GR::NT), lds_a, stream, a.in,
                       a.scratch, a.log_n, a.w, a.prescale, a.in_batch_stride, a.scratch_batch_stride, a.prescale_batch_stride, a.aux,
                       a.prescale ? a.aux_ps : nullptr, a.aux_ps_batch_stride);
    hipLaunchKernelGGL((k_ntt_rows_v4<CA, CB, CC, INV>), dim3(a.batch, (unsigned)GR::M / L2 / V4_TILES, a.width), dim3(GC::NT), lds_b, stream,
                       (const fp *)a.scratch, a.out, a.log_n, a.w, a.post_scale, a.do_scale ? 1 : 0, a.scratch_batch_stride, a.out_batch_stride,
                       a.aux ? a.aux + GR::M : nullptr);
    return hipGetLastError();
}

template <int LRA, int LRB, int LCA, int LCB, bool INV>
hipError_t launch_v2(const NttArgs &a, hipStream_t stream) {
    constexpr int LOG_R = LRA + LRB, LOG_C = LCA + LCB;
    constexpr int TA = L2 << (LRA > LRB ? LRA : LRB), TB = L2 << (LCA > LCB ? LCA : LCB);
    const size_t lds_a = ((size_t)(1 << LRA) * ((1 << LRB) + 1) * L2 + (1 << LOG_R)) * sizeof(fp);
    const size_t lds_b = ((size_t)L2 * (1 << LOG_C) + (1 << LOG_C)) * sizeof(fp);
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void *)k_ntt_cols_v2<LRA, LRB, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void *)k_ntt_rows_v2<LCA, LCB, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b)) != hipSuccess) return e;
    hipLaunchKernelGGL((k_ntt_cols_v2<LRA, LRB, INV>), dim3(a.batch, (1u << LOG_C) / L2, a.width), dim3(TA), lds_a, stream, a.in, a.scratch, a.log_n,
                       a.w, a.prescale, a.in_batch_stride, a.scratch_batch_stride, a.prescale_batch_stride);
    hipLaunchKernelGGL((k_ntt_rows_v2<LCA, LCB, INV>), dim3(a.batch, (1u << LOG_R) / L2, a.width), dim3(TB), lds_b, stream,
                       (const fp *)a.scratch, a.out, a.log_n, a.w, a.post_scale, a.do_scale ? 1 : 0, a.scratch_batch_stride, a.out_batch_stride);
    return hipGetLastError();
}

// table[e] = base^e for e < n; each thread produces CHUNK consecutive powers
constexpr int CHUNK = 16;
__global__ void k_power_table(fp *table, size_t n, fp base) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t e0 = t * CHUNK;
    if (e0 >= n) return;
    fp x = fp_pow(base, e0);
    for (int i = 0; i < CHUNK && e0 + i < n; i++) { table[e0 + i] = x; x = fp_mul(x, base); }
}

template <int L>
hipError_t launch(const fp *in, fp *scratch, fp *out, unsigned width, unsigned batch, unsigned log_n, const fp *w, const fp *prescale,
                  size_t prescale_batch_stride, fp post_scale, bool do_scale, size_t in_batch_stride, size_t scratch_batch_stride,
                  size_t out_batch_stride, hipStream_t stream) {
    const unsigned log_r = (log_n + 1) / 2, log_c = log_n - log_r;
    const unsigned R = 1u << log_r, C = 1u << log_c;
    const size_t lds_a = ((size_t)R * L + R / 2) * sizeof(fp), lds_b = ((size_t)C * L + C / 2) * sizeof(fp);
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void *)k_ntt_cols<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void *)k_ntt_rows<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_ntt_cols<L>, dim3(C / L, width, batch), dim3(NT), lds_a, stream, in, scratch, log_n, log_r, w, prescale,
                       in_batch_stride, scratch_batch_stride, prescale_batch_stride);
    hipLaunchKernelGGL(k_ntt_rows<L>, dim3(R / L, width, batch), dim3(NT), lds_b, stream, (const fp *)scratch, out, log_n, log_r, w,
                       post_scale, do_scale ? 1 : 0, scratch_batch_stride, out_batch_stride);
    return hipGetLastError();
}

} // namespace

// Composition-polynomial column split: h holds the N = b*n coefficients of H(g y) in y; column i of out gets
// H_i[q] = h[b*q + i] * g^-(b*q + i)  (coefficients of H in x, H(x) = sum_i x^i H_i(x^b)).  One thread per q.
__global__ void k_split_columns(const fp *__restrict__ h, fp *__restrict__ out, size_t n, unsigned log_b, fp ginv) {
    const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (q >= n) return;
    const unsigned b = 1u << log_b;
    fp s = fp_pow(ginv, (uint64_t)q << log_b);
    for (unsigned i = 0; i < b; i++) {
        out[(size_t)i * n + q] = fp_mul(h[((size_t)q << log_b) + i], s);
        s = fp_mul(s, ginv);
    }
}
hipError_t split_columns(const fp *d_h, fp *d_out, unsigned log_n, unsigned log_b, fp ginv, hipStream_t stream) {
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(k_split_columns, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_h, d_out, n, log_b, ginv);
    return hipGetLastError();
}
// natural[b*j + k] = coset_major[k*n + j]
__global__ void k_interleave_cosets(const fp *__restrict__ in, fp *__restrict__ out, size_t n, unsigned log_b) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (n << log_b)) return;
    const size_t k = i & ((1u << log_b) - 1), j = i >> log_b;
    out[i] = in[k * n + j];
}
hipError_t interleave_cosets(const fp *d_in, fp *d_out, unsigned log_n, unsigned log_b, hipStream_t stream) {
    const size_t N = (size_t)1 << (log_n + log_b);
    hipLaunchKernelGGL(k_interleave_cosets, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, d_in, d_out, (size_t)1 << log_n, log_b);
    return hipGetLastError();
}

// one thread per q: the B values of row q across the cosets, twisted and put through a B-point inverse DFT (B <= 8: B^2 products)
template <int LOGB>
__global__ __launch_bounds__(256) void k_coset_combine(const fp *__restrict__ in, fp *__restrict__ out, size_t n, const fp *__restrict__ winv, fp b_inv) {
    constexpr int B = 1 << LOGB;
    const size_t q = blockIdx.x * (size_t)256 + threadIdx.x;
    if (q >= n) return;
    in += (size_t)blockIdx.y * B * n; // grid.y = table
    out += (size_t)blockIdx.y * B * n;
    fp v[B], wb[B];
#pragma unroll
    for (int k = 0; k < B; k++) {
        wb[k] = winv[(size_t)k * n];                                   // w_B^-k
        const fp x = in[(size_t)k * n + q];
        v[k] = k == 0 ? fp_mul(x, b_inv) : fp_mul(fp_mul(x, b_inv), winv[(size_t)k * q]); // (1 / B) w_N^(-k q) B_k[q]
    }
#pragma unroll
    for (int i = 0; i < B; i++) {
        Acc128 a = acc_zero();
#pragma unroll
        for (int k = 0; k < B; k++) {
            acc_mad(a, v[k], wb[(k * i) & (B - 1)]);
            if (k == 6) acc_fold(a);
        }
        acc_fold(a);
        out[(size_t)i * n + q] = acc_reduce(a);
    }
}
// (kc0, nkc): only even cosets [kc0, kc0 + nkc) of `in` are present, the others count as zero (one rank's share of a sharded proof:
// the map is linear, the ranks' outputs add up)
__global__ __launch_bounds__(256) void k_coset_even_to_odd(const fp *__restrict__ in, fp *__restrict__ out, size_t n, unsigned tables,
                                                           const fp *__restrict__ winv4n, const fp *__restrict__ w8n, fp quarter, unsigned kc0, unsigned nkc) {
    const size_t q = blockIdx.x * (size_t)256 + threadIdx.x;
    if (q >= n) return;
    const unsigned tb = blockIdx.y;
    fp v[4], w4[4], w8[8], a[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        w4[k] = winv4n[(size_t)k * n]; // w_4^-k
        const fp x = ((unsigned)k - kc0 < nkc) ? fp_mul(in[((size_t)tb * 4 + k) * n + q], quarter) : 0;
        v[k] = k == 0 ? x : fp_mul(x, winv4n[(size_t)k * q]);
    }
#pragma unroll
    for (int e = 0; e < 8; e++) w8[e] = w8n[(size_t)e * n]; // w_8^e
#pragma unroll
    for (int i = 0; i < 4; i++) { // coefficient a_{q + n i}
        fp s = v[0];
#pragma unroll
        for (int k = 1; k < 4; k++) s = fp_add(s, fp_mul(v[k], w4[(k * i) & 3]));
        a[i] = s;
    }
#pragma unroll
    for (int kc = 0; kc < 4; kc++) {
        const int k = 2 * kc + 1;
        fp s = a[0];
#pragma unroll
        for (int i = 1; i < 4; i++) s = fp_add(s, fp_mul(a[i], w8[(k * i) & 7]));
        out[((size_t)kc * tables + tb) * n + q] = s;
    }
}
hipError_t coset_even_to_odd(const fp *d_b, fp *d_out, unsigned log_n, unsigned tables, const fp *d_winv_4n, const fp *d_w_8n, fp quarter,
                             hipStream_t stream, unsigned kc0, unsigned nkc) {
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(k_coset_even_to_odd, dim3((unsigned)((n + 255) / 256), tables), dim3(256), 0, stream, d_b, d_out, n, tables, d_winv_4n, d_w_8n, quarter,
                       kc0, nkc);
    return hipGetLastError();
}
hipError_t coset_combine(const fp *d_b, fp *d_h, unsigned log_n, unsigned log_b, const fp *d_winv_N, fp b_inv, hipStream_t stream, unsigned tables) {
    const size_t n = (size_t)1 << log_n;
    const dim3 grid((unsigned)((n + 255) / 256), tables), block(256);
    if (log_b == 1) hipLaunchKernelGGL(k_coset_combine<1>, grid, block, 0, stream, d_b, d_h, n, d_winv_N, b_inv);
    else if (log_b == 2) hipLaunchKernelGGL(k_coset_combine<2>, grid, block, 0, stream, d_b, d_h, n, d_winv_N, b_inv);
    else if (log_b == 3) hipLaunchKernelGGL(k_coset_combine<3>, grid, block, 0, stream, d_b, d_h, n, d_winv_N, b_inv);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- compact tables of the v4 kernels --------------------------------------------------------------------------------
// The twiddles of a sub-transform, the row part of a coset's prescale and the output factors of the column pass are strided or
// scattered entries of the n-entry tables (w[e << log_c], s[r << log_c], w[kb c]): 8 useful bytes per 64-byte sector and, for the
// output factors, one sector per lane.  Gathered once per table into dense arrays, a workgroup reads them as a handful of
// contiguous lines: 2.3 k instead of 5 k sector requests per column-pass workgroup.
bool ntt_v4_shape(unsigned log_n, NttV4Shape *s) {
    static const bool v2_env = [] { const char *e = getenv("CSTARK_NTT_V2"); return e && atoi(e) != 0; }();
    if (v2_env) return false;
    if (log_n == 20) { *s = {10, 10, 7}; return true; }
    if (log_n == 18) { *s = {9, 9, 6}; return true; }
    if (log_n == 16) { *s = {8, 8, 6}; return true; }
    return false;
}
namespace {
// plan block: [R] w^(e C) | [C] w^(e R) | [C] w^(KB c) | [KB][C] w^(kb c)
__global__ void k_aux_plan(fp *aux, const fp *__restrict__ w, unsigned log_r, unsigned log_c, unsigned log_kb) {
    const size_t R = (size_t)1 << log_r, C = (size_t)1 << log_c, total = R + 2 * C + (C << log_kb);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        fp v;
        if (i < R) v = w[i << log_c];
        else if (i < R + C) v = w[(i - R) << log_r];
        else if (i < R + 2 * C) v = w[(i - R - C) << log_kb];
        else { const size_t q = i - R - 2 * C, kb = q >> log_c, c = q & (C - 1); v = w[kb * c]; }
        aux[i] = v;
    }
}
// coset block: [R] s^(r C) | [KB][C] w^(kb c) s^c
__global__ void k_aux_coset(fp *aux, const fp *__restrict__ w, const fp *__restrict__ s, unsigned log_r, unsigned log_c, unsigned log_kb) {
    const size_t R = (size_t)1 << log_r, C = (size_t)1 << log_c, total = R + (C << log_kb);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        fp v;
        if (i < R) v = s[i << log_c];
        else { const size_t q = i - R, kb = q >> log_c, c = q & (C - 1); v = fp_mul(w[kb * c], s[c]); }
        aux[i] = v;
    }
}
} // namespace
size_t ntt_aux_plan_words(const NttV4Shape &s) { return ((size_t)1 << s.log_r) + ((size_t)2 << s.log_c) + ((size_t)1 << (s.log_c + s.log_kb)); }
size_t ntt_aux_coset_words(const NttV4Shape &s) { return ((size_t)1 << s.log_r) + ((size_t)1 << (s.log_c + s.log_kb)); }
hipError_t ntt_build_aux_plan(fp *d_aux, const fp *d_w, const NttV4Shape &s, hipStream_t stream) {
    hipLaunchKernelGGL(k_aux_plan, dim3(256), dim3(256), 0, stream, d_aux, d_w, s.log_r, s.log_c, s.log_kb);
    return hipGetLastError();
}
hipError_t ntt_build_aux_coset(fp *d_aux, const fp *d_w, const fp *d_s, const NttV4Shape &s, hipStream_t stream) {
    hipLaunchKernelGGL(k_aux_coset, dim3(256), dim3(256), 0, stream, d_aux, d_w, d_s, s.log_r, s.log_c, s.log_kb);
    return hipGetLastError();
}

hipError_t ntt_power_table(fp *d_table, size_t n, fp base, hipStream_t stream) {
    const size_t threads = (n + CHUNK - 1) / CHUNK;
    hipLaunchKernelGGL(k_power_table, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, d_table, n, base);
    return hipGetLastError();
}

hipError_t ntt_columns(const NttArgs &a, hipStream_t stream) {
    if (a.log_n < NTT_MIN_LOG_N || a.log_n > NTT_MAX_LOG_N) return hipErrorInvalidValue;
    // register-tiled kernels for the production sizes; `inverse` selects the compile-time small twiddles
    static const bool v2_env = [] { const char *e = getenv("CSTARK_NTT_V2"); return e && atoi(e) != 0; }(); // tuning / debugging: two-step kernels
    static const bool v4_env = [] { const char *e = getenv("CSTARK_NTT_V4"); return e && atoi(e) != 0; }(); // tuning / debugging: whole-tile exchanges
    if (a.log_n == 20 && !v2_env && !v4_env) return a.inverse ? launch_v5<4, 3, 3, 4, 3, 3, true>(a, stream) : launch_v5<4, 3, 3, 4, 3, 3, false>(a, stream);
    if (a.log_n == 20 && !v2_env) return a.inverse ? launch_v4<4, 3, 3, 4, 3, 3, true>(a, stream) : launch_v4<4, 3, 3, 4, 3, 3, false>(a, stream);
    if (a.log_n == 20) return a.inverse ? launch_v2<5, 5, 5, 5, true>(a, stream) : launch_v2<5, 5, 5, 5, false>(a, stream);
    if (a.log_n == 18 && !v2_env) return a.inverse ? launch_v4<3, 3, 3, 3, 3, 3, true>(a, stream) : launch_v4<3, 3, 3, 3, 3, 3, false>(a, stream);
    if (a.log_n == 16 && !v2_env) return a.inverse ? launch_v4<3, 3, 2, 3, 3, 2, true>(a, stream) : launch_v4<3, 3, 2, 3, 3, 2, false>(a, stream);
    if (a.log_n == 18) return a.inverse ? launch_v2<5, 4, 5, 4, true>(a, stream) : launch_v2<5, 4, 5, 4, false>(a, stream);
    if (a.log_n == 16) return a.inverse ? launch_v2<4, 4, 4, 4, true>(a, stream) : launch_v2<4, 4, 4, 4, false>(a, stream);
    // L = 8 keeps global segments at 64 bytes; sub-transforms above 2^11 points need the narrower tile to fit LDS
    const unsigned log_r = (a.log_n + 1) / 2;
    if (log_r <= 11)
        return launch<8>(a.in, a.scratch, a.out, a.width, a.batch, a.log_n, a.w, a.prescale, a.prescale_batch_stride, a.post_scale,
                         a.do_scale, a.in_batch_stride, a.scratch_batch_stride, a.out_batch_stride, stream);
    return launch<4>(a.in, a.scratch, a.out, a.width, a.batch, a.log_n, a.w, a.prescale, a.prescale_batch_stride, a.post_scale, a.do_scale,
                     a.in_batch_stride, a.scratch_batch_stride, a.out_batch_stride, stream);
}

} // namespace cs
