// Out-of-domain evaluation and DEEP composition (engine steps after the constraint commitment; "next" rows of SURVEY 8(f)).
// [UPSTREAM-RECALL winterfell v0.3]: the OOD frame is the trace polynomials at z and z*w_n and the composition columns at
// z^b; the DEEP composition polynomial is
//   [ sum_c alpha_c (T_c(x) - T_c(z)) / (x - z) + beta_c (T_c(x) - T_c(z w)) / (x - z w)
//   + sum_i delta_i (H_i(x) - H_i(z^b)) / (x - z^b) ] * (deg_a + deg_b x)
// evaluated over the LDE domain.  One lane per LDE point; column loads are contiguous per wave (coset-major layout).
#include "deep.h"
#include "fp.cuh"

namespace cs {
namespace {

// Polynomial evaluation at a few points, two passes.  Pass 1, grid = (segments, width): a workgroup owns SEG consecutive
// coefficients of one column, every lane reads them with stride 256 (coalesced 2 KB per wave-row) and runs one Horner chain
// in z^256 per point; the column is read from HBM once for all points.  partial[(p * width + c) * segments + s] =
// sum over the segment of c_m z_p^m.  Pass 2 adds the segments.
#ifndef CS_PE_SEG
#define CS_PE_SEG 16384
#endif
constexpr int PE_SEG = CS_PE_SEG, PE_MAXPTS = 2;
template <int NP>
__global__ __launch_bounds__(256) void k_poly_eval_partial(const fp *__restrict__ coeffs, size_t n, const fp *__restrict__ points, fp *__restrict__ partial,
                                                           unsigned width, unsigned seg_len) {
    __shared__ fp part[NP][256];
    const unsigned segs = gridDim.x, seg = blockIdx.x, col = blockIdx.y, t = threadIdx.x;
    const fp *c = coeffs + (size_t)col * n + (size_t)seg * seg_len;
    fp z[NP], z256[NP], acc[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) { z[p] = points[p]; z256[p] = fp_pow(z[p], 256); acc[p] = 0; }
    const unsigned per = (seg_len + 255) / 256; // seg_len is a multiple of 256, or the whole (short) column
    for (unsigned k = per; k-- > 0;) {
        const fp v = k * 256 + t < seg_len ? c[(size_t)k * 256 + t] : 0;
#pragma unroll
        for (int p = 0; p < NP; p++) acc[p] = fp_add(fp_mul(acc[p], z256[p]), v);
    }
#pragma unroll
    for (int p = 0; p < NP; p++) part[p][t] = fp_mul(acc[p], fp_pow(z[p], (uint64_t)seg * seg_len + t));
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)t < s) {
#pragma unroll
            for (int p = 0; p < NP; p++) part[p][t] = fp_add(part[p][t], part[p][t + s]);
        }
        __syncthreads();
    }
    if (t < NP) partial[((size_t)t * width + col) * segs + seg] = part[t][0];
}
__global__ void k_poly_eval_sum(const fp *__restrict__ partial, fp *__restrict__ out, unsigned total, unsigned segs) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    fp a = 0;
    for (unsigned s = 0; s < segs; s++) a = fp_add(a, partial[(size_t)i * segs + s]);
    out[i] = a;
}

// grid = (n / 256, nk)
__global__ __launch_bounds__(256) void k_deep(DeepParams p) {
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)256 + threadIdx.x;
    if (j >= n) return;
    const unsigned kk = blockIdx.y;
    const fp x = fp_mul(p.shifts[p.k0 + kk], p.w[j]);
    const fp pz = p.scal ? p.scal[0] : p.z, pzw = p.scal ? p.scal[1] : p.zw, pzb = p.scal ? p.scal[2] : p.zb;
    const fp d1 = fp_sub(x, pz), d2 = fp_sub(x, pzw), d3 = fp_sub(x, pzb);
    const fp inv = fp_inv(fp_mul(fp_mul(d1, d2), d3)); // one inversion for the three divisors
    const fp i1 = fp_mul(inv, fp_mul(d2, d3)), i2 = fp_mul(inv, fp_mul(d1, d3)), i3 = fp_mul(inv, fp_mul(d1, d2));
    Acc128 s1 = acc_zero(), s2 = acc_zero(), s3 = acc_zero();
    const fp *t = p.trace_lde + (size_t)kk * p.width * n + j;
    // (batching 8 column loads per lane ahead of their use was measured slower, 1.94 vs 1.82 ms: the extra registers halve the
    // occupancy, and eight resident waves per SIMD with one request each already keep the memory system busy)
    for (unsigned c = 0; c < p.width; c++) {
        const fp v = t[(size_t)c * n];
        acc_mad(s1, p.coef[c], fp_sub(v, p.ood[c]));
        acc_mad(s2, p.coef[p.width + c], fp_sub(v, p.ood[p.width + c]));
        if ((c & 3) == 3) { acc_fold(s1); acc_fold(s2); }
    }
    acc_fold(s1); acc_fold(s2);
    const fp *h = p.comp_lde + (size_t)kk * p.nb * n + j;
    for (unsigned i = 0; i < p.nb; i++) {
        acc_mad(s3, p.coef[2 * p.width + i], fp_sub(h[(size_t)i * n], p.ood[2 * p.width + i]));
        if ((i & 3) == 3) acc_fold(s3);
    }
    acc_fold(s3);
    fp acc = fp_add(fp_add(fp_mul(acc_reduce(s1), i1), fp_mul(acc_reduce(s2), i2)), fp_mul(acc_reduce(s3), i3));
    const fp dga = p.scal ? p.scal[3] : p.deg_a, dgb = p.scal ? p.scal[4] : p.deg_b;
    p.out[(size_t)kk * n + j] = fp_mul(acc, fp_add(dga, fp_mul(dgb, x)));
}

// FRI layer folding, factor 4 [UPSTREAM-RECALL winterfell-fri apply_drp]: row i = { f(x_i zeta^t) } = evals[i + t N/4];
// the cubic through the four points evaluated at alpha:  (1/4) sum_k (alpha / x_i)^k sum_t v_t zeta^(-t k).
// winv = powers of w_N^-1 (x_i^-1 = offset^-1 * winv[i]); zeta^-1 = winv[N/4] is a primitive 4th root: zeta^-2 = -1.
// alpha_dev != null: the folding point is read from device memory (drawn there by the device-side coin, blake3.hip k_fri_coin)
__global__ __launch_bounds__(256) void k_fri_fold4(const fp *__restrict__ evals, fp *__restrict__ out, size_t q, const fp *__restrict__ winv,
                                                   fp offset_inv, fp alpha, fp inv4, const fp *__restrict__ alpha_dev) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i >= q) return;
    if (alpha_dev) alpha = *alpha_dev;
    const fp v0 = evals[i], v1 = evals[i + q], v2 = evals[i + 2 * q], v3 = evals[i + 3 * q];
    const fp zi = winv[q]; // zeta^-1
    // size-4 inverse DFT: s_k = sum_t v_t zeta^(-t k)
    const fp a = fp_add(v0, v2), b = fp_sub(v0, v2), c = fp_add(v1, v3), d = fp_mul(fp_sub(v1, v3), zi);
    const fp s0 = fp_add(a, c), s2 = fp_sub(a, c), s1 = fp_add(b, d), s3 = fp_sub(b, d);
    const fp r = fp_mul(alpha, fp_mul(offset_inv, winv[i]));
    const fp r2 = fp_sqr(r), r3 = fp_mul(r2, r);
    fp acc = fp_add(fp_add(s0, fp_mul(r, s1)), fp_add(fp_mul(r2, s2), fp_mul(r3, s3)));
    out[i] = fp_mul(acc, inv4);
}

// Folding factor F = 2^LOG_F = 8, 16 (FriOptions::folding_factor, examples/state-transition.rs:46-47): row i = { f(x_i zeta^t) } =
// evals[i + t N/F], zeta = w_N^(N/F); the polynomial of degree F - 1 through the row evaluated at alpha:
// (1/F) sum_k (alpha / x_i)^k s_k,  s_k = sum_t v_t zeta^(-t k): an F-point inverse DFT by radix-2 steps in registers (exact
// arithmetic: the same values as the direct sums of the CPU restatement), then Horner in alpha / x_i.
template <int LOG_F>
__global__ __launch_bounds__(256) void k_fri_fold(const fp *__restrict__ evals, fp *__restrict__ out, size_t q, const fp *__restrict__ winv,
                                                  fp offset_inv, fp alpha, fp inv_f, const fp *__restrict__ alpha_dev) {
    constexpr int F = 1 << LOG_F;
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i >= q) return;
    if (alpha_dev) alpha = *alpha_dev;
    fp v[F];
#pragma unroll
    for (int t = 0; t < F; t++) v[t] = evals[i + (size_t)t * q];
    // decimation in frequency with the inverse root zeta^-1 = winv[q] (zeta^-e = winv[e q]): outputs in bit-reversed order
#pragma unroll
    for (int len = F; len >= 2; len >>= 1) {
        const int half = len >> 1, step = F / len;
#pragma unroll
        for (int base = 0; base < F; base += len) {
#pragma unroll
            for (int t = 0; t < half; t++) {
                const fp a = v[base + t], b = v[base + t + half];
                v[base + t] = fp_add(a, b);
                const fp d = fp_sub(a, b);
                v[base + t + half] = (t == 0) ? d : fp_mul(d, winv[(size_t)(t * step) * q]);
            }
        }
    }
    const fp r = fp_mul(alpha, fp_mul(offset_inv, winv[i]));
    fp acc = 0; // Horner from s_(F-1) down: s_k sits at the bit-reversed index
#pragma unroll
    for (int k = F - 1; k >= 0; k--) {
        int br = 0;
#pragma unroll
        for (int bit = 0; bit < LOG_F; bit++) br |= ((k >> bit) & 1) << (LOG_F - 1 - bit);
        acc = fp_add(fp_mul(acc, r), v[br]);
    }
    out[i] = fp_mul(acc, inv_f);
}

} // namespace

hipError_t fri_fold4(const uint64_t *d_evals, uint64_t *d_out, unsigned log_n, const uint64_t *d_winv, uint64_t offset_inv, uint64_t alpha,
                     uint64_t inv4, hipStream_t stream, const uint64_t *d_alpha) {
    const size_t q = ((size_t)1 << log_n) / 4;
    hipLaunchKernelGGL(k_fri_fold4, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, stream, d_evals, d_out, q, d_winv, offset_inv, alpha, inv4, d_alpha);
    return hipGetLastError();
}
hipError_t fri_fold(const uint64_t *d_evals, uint64_t *d_out, unsigned log_n, unsigned log_f, const uint64_t *d_winv, uint64_t offset_inv, uint64_t alpha,
                    uint64_t inv_f, hipStream_t stream, const uint64_t *d_alpha) {
    if (log_f == 2) return fri_fold4(d_evals, d_out, log_n, d_winv, offset_inv, alpha, inv_f, stream, d_alpha);
    if ((log_f != 3 && log_f != 4) || log_n < log_f) return hipErrorInvalidValue;
    const size_t q = ((size_t)1 << log_n) >> log_f;
    const dim3 grid((unsigned)((q + 255) / 256));
    if (log_f == 3) hipLaunchKernelGGL(k_fri_fold<3>, grid, dim3(256), 0, stream, d_evals, d_out, q, d_winv, offset_inv, alpha, inv_f, d_alpha);
    else hipLaunchKernelGGL(k_fri_fold<4>, grid, dim3(256), 0, stream, d_evals, d_out, q, d_winv, offset_inv, alpha, inv_f, d_alpha);
    return hipGetLastError();
}

size_t poly_eval_scratch_words(unsigned width, unsigned log_n, unsigned npts) {
    const size_t n = (size_t)1 << log_n, seg = n < (size_t)PE_SEG ? n : (size_t)PE_SEG;
    return (size_t)npts * width * (n / seg);
}
hipError_t poly_eval(const uint64_t *d_coeffs, unsigned width, unsigned log_n, const uint64_t *d_points, unsigned npts, uint64_t *d_out,
                     uint64_t *d_scratch, hipStream_t stream) {
    const size_t n = (size_t)1 << log_n;
    const unsigned seg_len = (unsigned)(n < (size_t)PE_SEG ? n : (size_t)PE_SEG), segs = (unsigned)(n / seg_len);
    for (unsigned p0 = 0; p0 < npts; p0 += PE_MAXPTS) { // points in groups of two: one pass over the coefficients per group
        const unsigned np = npts - p0 >= 2 ? 2 : 1;
        fp *part = d_scratch + (size_t)p0 * width * segs;
        if (np == 2) hipLaunchKernelGGL(k_poly_eval_partial<2>, dim3(segs, width), dim3(256), 0, stream, d_coeffs, n, d_points + p0, part, width, seg_len);
        else hipLaunchKernelGGL(k_poly_eval_partial<1>, dim3(segs, width), dim3(256), 0, stream, d_coeffs, n, d_points + p0, part, width, seg_len);
    }
    const unsigned total = npts * width;
    hipLaunchKernelGGL(k_poly_eval_sum, dim3((total + 255) / 256), dim3(256), 0, stream, d_scratch, d_out, total, segs);
    return hipGetLastError();
}
hipError_t deep_composition(const DeepParams &p, unsigned nk, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    hipLaunchKernelGGL(k_deep, dim3((unsigned)((n + 255) / 256), nk), dim3(256), 0, stream, p);
    return hipGetLastError();
}

} // namespace cs
