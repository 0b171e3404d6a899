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

namespace cs {
namespace {

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
            tile[i1 * L + l] = fp_mul(fp_sub(u, v), tw[j << s]);
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

hipError_t ntt_power_table(fp *d_table, size_t n, fp base, hipStream_t stream) {
    const size_t threads = (n + CHUNK - 1) / CHUNK;
    hipLaunchKernelGGL(k_power_table, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, d_table, n, base);
    return hipGetLastError();
}

hipError_t ntt_columns(const NttArgs &a, hipStream_t stream) {
    if (a.log_n < NTT_MIN_LOG_N || a.log_n > NTT_MAX_LOG_N) return hipErrorInvalidValue;
    // L = 8 keeps global segments at 64 bytes; sub-transforms above 2^11 points need the narrower tile to fit LDS
    const unsigned log_r = (a.log_n + 1) / 2;
    if (log_r <= 11)
        return launch<8>(a.in, a.scratch, a.out, a.width, a.batch, a.log_n, a.w, a.prescale, a.prescale_batch_stride, a.post_scale,
                         a.do_scale, a.in_batch_stride, a.scratch_batch_stride, a.out_batch_stride, stream);
    return launch<4>(a.in, a.scratch, a.out, a.width, a.batch, a.log_n, a.w, a.prescale, a.prescale_batch_stride, a.post_scale, a.do_scale,
                     a.in_batch_stride, a.scratch_batch_stride, a.out_batch_stride, stream);
}

} // namespace cs
