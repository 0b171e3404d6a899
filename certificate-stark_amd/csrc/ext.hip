// FieldExtension::Quadratic / Cubic: the stages after the constraint evaluation over an extension of f63.
// [UPSTREAM-RECALL winterfell v0.3; the extensions the fork defines for f63 are not in the reference tree -- parity unpinned.
// Assumed: the two polynomials of the reference's own curve tower (src/utils/ecc.rs:407-648): E2 = F_p[u]/(u^2 - 2u - 2) and
// E3 = F_p[v]/(v^3 + v + 1) -- CSTARK_CONV_E2_* / E3_* of include/cstark_conventions.h; the arithmetic below is generic in those
// coefficients.]  The trace is base-field; coefficients, the out-of-domain point, the DEEP composition and FRI are in the extension.
#include "ext.h"
#include "fp.cuh"
#include "../../include/cstark_conventions.h"

namespace cs {
namespace {

template <int M> struct Ext { fp c[M]; };
template <int M> __device__ __forceinline__ Ext<M> x_zero() { Ext<M> r; for (int i = 0; i < M; i++) r.c[i] = 0; return r; }
template <int M> __device__ __forceinline__ Ext<M> x_one() { Ext<M> r = x_zero<M>(); r.c[0] = FP_ONE; return r; }
template <int M> __device__ __forceinline__ Ext<M> x_load(const uint64_t *p) { Ext<M> r; for (int i = 0; i < M; i++) r.c[i] = p[i]; return r; }
template <int M> __device__ __forceinline__ Ext<M> x_add(Ext<M> x, Ext<M> y) { Ext<M> r; for (int i = 0; i < M; i++) r.c[i] = fp_add(x.c[i], y.c[i]); return r; }
template <int M> __device__ __forceinline__ Ext<M> x_sub(Ext<M> x, Ext<M> y) { Ext<M> r; for (int i = 0; i < M; i++) r.c[i] = fp_sub(x.c[i], y.c[i]); return r; }
template <int M> __device__ __forceinline__ Ext<M> x_scale(Ext<M> x, fp s) { Ext<M> r; for (int i = 0; i < M; i++) r.c[i] = fp_mul(x.c[i], s); return r; }
// x times a small signed integer constant (reduction coefficients of the extension polynomials)
template <int C> __device__ __forceinline__ fp x_small(fp x) {
    static_assert(C >= -4 && C <= 4, "extension polynomial coefficients are small integers");
    return fp_mul_small(x, C);
}
__device__ __forceinline__ Ext<2> x_mul(Ext<2> x, Ext<2> y) { // u^2 = C1 u + C0
    const fp bd = fp_mul(x.c[1], y.c[1]);
    return {{fp_add(fp_mul(x.c[0], y.c[0]), x_small<CSTARK_CONV_E2_C0>(bd)),
             fp_add(fp_add(fp_mul(x.c[0], y.c[1]), fp_mul(x.c[1], y.c[0])), x_small<CSTARK_CONV_E2_C1>(bd))}};
}
__device__ __forceinline__ Ext<3> x_mul(Ext<3> x, Ext<3> y) { // v^3 = C2 v^2 + C1 v + C0, v^4 = (C2^2 + C1) v^2 + (C2 C1 + C0) v + C2 C0
    constexpr int C0 = CSTARK_CONV_E3_C0, C1 = CSTARK_CONV_E3_C1, C2 = CSTARK_CONV_E3_C2;
    const fp d0 = fp_mul(x.c[0], y.c[0]);
    const fp d1 = fp_add(fp_mul(x.c[0], y.c[1]), fp_mul(x.c[1], y.c[0]));
    const fp d2 = fp_add(fp_add(fp_mul(x.c[0], y.c[2]), fp_mul(x.c[1], y.c[1])), fp_mul(x.c[2], y.c[0]));
    const fp d3 = fp_add(fp_mul(x.c[1], y.c[2]), fp_mul(x.c[2], y.c[1]));
    const fp d4 = fp_mul(x.c[2], y.c[2]);
    return {{fp_add(fp_add(d0, x_small<C0>(d3)), x_small<C2 * C0>(d4)),
             fp_add(fp_add(d1, x_small<C1>(d3)), x_small<C2 * C1 + C0>(d4)),
             fp_add(fp_add(d2, x_small<C2>(d3)), x_small<C2 * C2 + C1>(d4))}};
}
template <int M> __device__ inline Ext<M> x_pow(Ext<M> x, uint64_t e) {
    Ext<M> r = x_one<M>();
    while (e) {
        if (e & 1) r = x_mul(r, x);
        x = x_mul(x, x);
        e >>= 1;
    }
    return r;
}
// 1 / (x - z) for base x as adjugate / norm (the norm is a base-field element, so several inverses share one inversion)
template <int M> struct XInv { Ext<M> adj; fp norm; };
__device__ __forceinline__ XInv<2> x_inv_parts(fp x, const uint64_t *zc, Ext<2> *) {
    // (a + b u)^-1 = (a + C1 b - b u) / (a (a + C1 b) - C0 b^2)   for u^2 = C1 u + C0
    const fp a = fp_sub(x, zc[0]), b = fp_neg(zc[1]);
    const fp t = fp_add(a, x_small<CSTARK_CONV_E2_C1>(b));
    return {{{t, fp_neg(b)}}, fp_sub(fp_mul(a, t), x_small<CSTARK_CONV_E2_C0>(fp_sqr(b)))};
}
__device__ __forceinline__ XInv<3> x_inv_parts(fp x, const uint64_t *zc, Ext<3> *) {
    // Multiplication by e = a + b v + c v^2 in the basis (1, v, v^2) is the matrix with columns e, e v, e v^2; e^-1 is the first
    // column of its adjugate over its determinant: the cofactors of the first row (for v^3 + v + 1: the cubic layer of ecc.rs:551-591).
    constexpr int C0 = CSTARK_CONV_E3_C0, C1 = CSTARK_CONV_E3_C1, C2 = CSTARK_CONV_E3_C2;
    const fp a = fp_sub(x, zc[0]), b = fp_neg(zc[1]), c = fp_neg(zc[2]);
    const fp y0 = x_small<C0>(c), y1 = fp_add(a, x_small<C1>(c)), y2 = fp_add(b, x_small<C2>(c));          // e v
    const fp w0 = x_small<C0>(y2), w1 = fp_add(y0, x_small<C1>(y2)), w2 = fp_add(y1, x_small<C2>(y2));     // e v^2
    const fp r0 = fp_sub(fp_mul(y1, w2), fp_mul(y2, w1));
    const fp r1 = fp_sub(fp_mul(c, w1), fp_mul(b, w2));
    const fp r2 = fp_sub(fp_mul(b, y2), fp_mul(c, y1));
    return {{{r0, r1, r2}}, fp_add(fp_add(fp_mul(a, r0), fp_mul(y0, r1)), fp_mul(w0, r2))};
}

constexpr int PE_SEG = 16384;
// grid = (segments, width): one segment of one column at an extension point.  Lane t owns the coefficients t, t + 256, ... of the
// segment: sum_k c[256 k + t] (z^256)^k is a dot product of BASE coefficients with the components of the powers (z^256)^k, which
// are the same for every lane -- tabulated once per workgroup in LDS, M lazily accumulated base products per coefficient instead
// of an extension multiplication (9 base products for M = 3).  The lane's sum is then moved into place by z^t and z^(segment start).
template <int M>
__global__ __launch_bounds__(256) void k_poly_eval_ext_partial(const fp *__restrict__ coeffs, size_t n, Ext<M> z, fp *__restrict__ partial, unsigned seg_len) {
    __shared__ Ext<M> part[256];
    __shared__ fp pk[M][PE_SEG / 256];
    __shared__ Ext<M> zseg;
    const unsigned segs = gridDim.x, seg = blockIdx.x, col = blockIdx.y, t = threadIdx.x;
    const fp *c = coeffs + (size_t)col * n + (size_t)seg * seg_len;
    const unsigned per = (seg_len + 255) / 256;
    {
        const Ext<M> z256 = x_pow(z, 256);
        if (t < per) {
            const Ext<M> pw = x_pow(z256, t);
#pragma unroll
            for (int q = 0; q < M; q++) pk[q][t] = pw.c[q];
        }
        if (t == 255) zseg = x_pow(z, (uint64_t)seg * seg_len);
    }
    __syncthreads();
    Acc128 a[M];
#pragma unroll
    for (int q = 0; q < M; q++) a[q] = acc_zero();
    for (unsigned k = 0; k < per; k++) {
        const fp v = k * 256 + t < seg_len ? c[(size_t)k * 256 + t] : 0;
#pragma unroll
        for (int q = 0; q < M; q++) {
            acc_mad(a[q], v, pk[q][k]);
            if (k % 7 == 6) acc_fold(a[q]);
        }
    }
    Ext<M> acc;
#pragma unroll
    for (int q = 0; q < M; q++) { acc_fold(a[q]); acc.c[q] = acc_reduce(a[q]); }
    part[t] = x_mul(x_mul(acc, x_pow(z, t)), zseg);
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)t < s) part[t] = x_add(part[t], part[t + s]);
        __syncthreads();
    }
    if (t == 0)
        for (int q = 0; q < M; q++) partial[M * ((size_t)col * segs + seg) + q] = part[0].c[q];
}
__global__ void k_poly_eval_ext_sum(const fp *__restrict__ partial, fp *__restrict__ out, unsigned width, unsigned segs, unsigned m) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x; // (column, component)
    if (i >= m * width) return;
    const unsigned col = i / m, k = i % m;
    fp a = 0;
    for (unsigned s = 0; s < segs; s++) a = fp_add(a, partial[m * ((size_t)col * segs + s) + k]);
    out[i] = a;
}

// grid = (ceil(n / 256), b)
template <int M>
__global__ __launch_bounds__(256) void k_deep_ext(DeepExtParams p) {
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)256 + threadIdx.x;
    if (j >= n) return;
    const unsigned k = blockIdx.y, b = p.nk ? p.nk : 1u << p.log_b; // b: cosets in the output
    const fp x = fp_mul(p.shifts[k], p.w[j]);
    // the three divisors with one base-field inversion
    const XInv<M> q1 = x_inv_parts(x, p.z, (Ext<M> *)nullptr), q2 = x_inv_parts(x, p.zw, (Ext<M> *)nullptr), q3 = x_inv_parts(x, p.zb, (Ext<M> *)nullptr);
    const fp n12 = fp_mul(q1.norm, q2.norm);
    const fp inv = fp_inv(fp_mul(n12, q3.norm));
    const Ext<M> i1 = x_scale(q1.adj, fp_mul(inv, fp_mul(q2.norm, q3.norm))), i2 = x_scale(q2.adj, fp_mul(inv, fp_mul(q1.norm, q3.norm))),
                 i3 = x_scale(q3.adj, fp_mul(inv, n12));
    // sum_c alpha_c T_c(x), sum_c beta_c T_c(x): base values against extension coefficients = component-wise dot products
    Acc128 s1[M], s2[M];
#pragma unroll
    for (int q = 0; q < M; q++) { s1[q] = acc_zero(); s2[q] = acc_zero(); }
    const fp *t = p.trace_lde + (size_t)k * p.width * n + j;
    const fp *al = p.coef, *be = p.coef + (size_t)M * p.width, *de = p.coef + (size_t)2 * M * p.width;
    for (unsigned c = 0; c < p.width; c++) {
        const fp v = t[(size_t)c * n];
#pragma unroll
        for (int q = 0; q < M; q++) { acc_mad(s1[q], al[M * c + q], v); acc_mad(s2[q], be[M * c + q], v); }
        if ((c & 3) == 3) {
#pragma unroll
            for (int q = 0; q < M; q++) { acc_fold(s1[q]); acc_fold(s2[q]); }
        }
    }
    Ext<M> e1, e2;
#pragma unroll
    for (int q = 0; q < M; q++) {
        acc_fold(s1[q]); acc_fold(s2[q]);
        e1.c[q] = fp_sub(acc_reduce(s1[q]), p.k1[q]);
        e2.c[q] = fp_sub(acc_reduce(s2[q]), p.k2[q]);
    }
    Ext<M> e3 = x_zero<M>();
    const fp *h = p.comp_lde + (size_t)k * M * p.nb * n + j;
    for (unsigned i = 0; i < p.nb; i++) {
        Ext<M> hv;
#pragma unroll
        for (int q = 0; q < M; q++) hv.c[q] = h[(size_t)(M * i + q) * n];
        e3 = x_add(e3, x_mul(x_load<M>(de + M * i), hv));
    }
    e3 = x_sub(e3, x_load<M>(p.k3));
    Ext<M> acc = x_add(x_add(x_mul(e1, i1), x_mul(e2, i2)), x_mul(e3, i3));
    acc = x_mul(acc, x_add(x_load<M>(p.deg_a), x_scale(x_load<M>(p.deg_b), x)));
#pragma unroll
    for (int q = 0; q < M; q++) p.out[((size_t)q * b + k) * n + j] = acc.c[q];
}

template <int M>
__global__ __launch_bounds__(256) void k_fri_fold4_ext(const fp *__restrict__ evals, fp *__restrict__ out, size_t q, const fp *__restrict__ winv,
                                                       fp offset_inv, Ext<M> alpha, fp inv4) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i >= q) return;
    const size_t N = 4 * q;
    const fp zi = winv[q]; // zeta^-1
    Ext<M> s[4];
#pragma unroll
    for (int comp = 0; comp < M; comp++) { // the size-4 inverse DFT is linear over the base field: per component
        const fp *e = evals + comp * N;
        const fp v0 = e[i], v1 = e[i + q], v2 = e[i + 2 * q], v3 = e[i + 3 * q];
        const fp a = fp_add(v0, v2), b = fp_sub(v0, v2), c = fp_add(v1, v3), d = fp_mul(fp_sub(v1, v3), zi);
        s[0].c[comp] = fp_add(a, c); s[2].c[comp] = fp_sub(a, c); s[1].c[comp] = fp_add(b, d); s[3].c[comp] = fp_sub(b, d);
    }
    const Ext<M> r = x_scale(alpha, fp_mul(offset_inv, winv[i]));
    const Ext<M> r2 = x_mul(r, r), r3 = x_mul(r2, r);
    Ext<M> acc = x_add(x_add(s[0], x_mul(r, s[1])), x_add(x_mul(r2, s[2]), x_mul(r3, s[3])));
    acc = x_scale(acc, inv4);
#pragma unroll
    for (int comp = 0; comp < M; comp++) out[comp * q + i] = acc.c[comp];
}

// folding factor F = 2^LOG_F = 8, 16 over the extension (deep.hip, k_fri_fold: the F-point inverse DFT is linear over the base field --
// per component -- and the Horner steps in alpha / x_i are extension products)
template <int M, int LOG_F>
__global__ __launch_bounds__(256) void k_fri_fold_ext(const fp *__restrict__ evals, fp *__restrict__ out, size_t q, const fp *__restrict__ winv,
                                                      fp offset_inv, Ext<M> alpha, fp inv_f) {
    constexpr int F = 1 << LOG_F;
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i >= q) return;
    const size_t N = (size_t)F * q;
    Ext<M> s[F];
#pragma unroll
    for (int comp = 0; comp < M; comp++) {
        fp v[F];
#pragma unroll
        for (int t = 0; t < F; t++) v[t] = evals[comp * N + i + (size_t)t * q];
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
#pragma unroll
        for (int t = 0; t < F; t++) s[t].c[comp] = v[t]; // bit-reversed order
    }
    const Ext<M> r = x_scale(alpha, fp_mul(offset_inv, winv[i]));
    Ext<M> acc = x_zero<M>();
#pragma unroll
    for (int k = F - 1; k >= 0; k--) {
        int br = 0;
#pragma unroll
        for (int bit = 0; bit < LOG_F; bit++) br |= ((k >> bit) & 1) << (LOG_F - 1 - bit);
        acc = x_add(x_mul(acc, r), s[br]);
    }
    acc = x_scale(acc, inv_f);
#pragma unroll
    for (int comp = 0; comp < M; comp++) out[comp * q + i] = acc.c[comp];
}

} // namespace

size_t poly_eval_ext_scratch_words(unsigned width, unsigned log_n, unsigned m) {
    const size_t n = (size_t)1 << log_n, seg = n < (size_t)PE_SEG ? n : (size_t)PE_SEG;
    return (size_t)m * width * (n / seg);
}
hipError_t poly_eval_ext(const uint64_t *d_coeffs, unsigned width, unsigned log_n, const uint64_t *z, unsigned m, uint64_t *d_out, uint64_t *d_scratch,
                         hipStream_t stream) {
    const size_t n = (size_t)1 << log_n;
    const unsigned seg_len = (unsigned)(n < (size_t)PE_SEG ? n : (size_t)PE_SEG), segs = (unsigned)(n / seg_len);
    if (m == 2) hipLaunchKernelGGL(k_poly_eval_ext_partial<2>, dim3(segs, width), dim3(256), 0, stream, d_coeffs, n, Ext<2>{{z[0], z[1]}}, d_scratch, seg_len);
    else if (m == 3) hipLaunchKernelGGL(k_poly_eval_ext_partial<3>, dim3(segs, width), dim3(256), 0, stream, d_coeffs, n, Ext<3>{{z[0], z[1], z[2]}}, d_scratch, seg_len);
    else return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_poly_eval_ext_sum, dim3((m * width + 255) / 256), dim3(256), 0, stream, d_scratch, d_out, width, segs, m);
    return hipGetLastError();
}
hipError_t deep_composition_ext(const DeepExtParams &p, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    const dim3 grid((unsigned)((n + 255) / 256), p.nk ? p.nk : 1u << p.log_b);
    if (p.m == 2) hipLaunchKernelGGL(k_deep_ext<2>, grid, dim3(256), 0, stream, p);
    else if (p.m == 3) hipLaunchKernelGGL(k_deep_ext<3>, grid, dim3(256), 0, stream, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t fri_fold4_ext(const uint64_t *d_evals, uint64_t *d_out, unsigned log_n, const uint64_t *d_winv, uint64_t offset_inv, const uint64_t *alpha,
                         unsigned m, uint64_t inv4, hipStream_t stream) {
    const size_t q = ((size_t)1 << log_n) / 4;
    const dim3 grid((unsigned)((q + 255) / 256));
    if (m == 2) hipLaunchKernelGGL(k_fri_fold4_ext<2>, grid, dim3(256), 0, stream, d_evals, d_out, q, d_winv, offset_inv, Ext<2>{{alpha[0], alpha[1]}}, inv4);
    else if (m == 3) hipLaunchKernelGGL(k_fri_fold4_ext<3>, grid, dim3(256), 0, stream, d_evals, d_out, q, d_winv, offset_inv, Ext<3>{{alpha[0], alpha[1], alpha[2]}}, inv4);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t fri_fold_ext(const uint64_t *d_evals, uint64_t *d_out, unsigned log_n, unsigned log_f, const uint64_t *d_winv, uint64_t offset_inv,
                        const uint64_t *alpha, unsigned m, uint64_t inv_f, hipStream_t stream) {
    if (log_f == 2) return fri_fold4_ext(d_evals, d_out, log_n, d_winv, offset_inv, alpha, m, inv_f, stream);
    if ((log_f != 3 && log_f != 4) || log_n < log_f || (m != 2 && m != 3)) return hipErrorInvalidValue;
    const size_t q = ((size_t)1 << log_n) >> log_f;
    const dim3 grid((unsigned)((q + 255) / 256));
    if (m == 2 && log_f == 3) hipLaunchKernelGGL((k_fri_fold_ext<2, 3>), grid, dim3(256), 0, stream, d_evals, d_out, q, d_winv, offset_inv, Ext<2>{{alpha[0], alpha[1]}}, inv_f);
    else if (m == 2) hipLaunchKernelGGL((k_fri_fold_ext<2, 4>), grid, dim3(256), 0, stream, d_evals, d_out, q, d_winv, offset_inv, Ext<2>{{alpha[0], alpha[1]}}, inv_f);
    else if (log_f == 3) hipLaunchKernelGGL((k_fri_fold_ext<3, 3>), grid, dim3(256), 0, stream, d_evals, d_out, q, d_winv, offset_inv, Ext<3>{{alpha[0], alpha[1], alpha[2]}}, inv_f);
    else hipLaunchKernelGGL((k_fri_fold_ext<3, 4>), grid, dim3(256), 0, stream, d_evals, d_out, q, d_winv, offset_inv, Ext<3>{{alpha[0], alpha[1], alpha[2]}}, inv_f);
    return hipGetLastError();
}

namespace {
struct SetPtrs { const uint64_t *p[3]; };
// grid = (ceil(n / 256), cols * m)
__global__ void k_interleave_set_columns(uint64_t *__restrict__ dst, SetPtrs src, unsigned m, size_t n) {
    const size_t j = blockIdx.x * (size_t)256 + threadIdx.x;
    if (j >= n) return;
    const unsigned col = blockIdx.y, i = col / m, q = col % m;
    dst[(size_t)col * n + j] = src.p[q][(size_t)i * n + j];
}
} // namespace
hipError_t interleave_set_columns(uint64_t *d_dst, const uint64_t *const src[3], unsigned m, unsigned cols, size_t n, hipStream_t stream) {
    if (m == 0 || m > 3) return hipErrorInvalidValue;
    SetPtrs sp{{src[0], m > 1 ? src[1] : nullptr, m > 2 ? src[2] : nullptr}};
    hipLaunchKernelGGL(k_interleave_set_columns, dim3((unsigned)((n + 255) / 256), cols * m), dim3(256), 0, stream, d_dst, sp, m, n);
    return hipGetLastError();
}
} // namespace cs
