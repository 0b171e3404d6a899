// FieldExtension::Quadratic: the stages after the constraint evaluation over E = F_p[u]/(u^2 - 2u - 2).
// [UPSTREAM-RECALL winterfell v0.3; the extension the fork defines for f63 is not in the reference tree -- parity unpinned.
// Assumed: the quadratic extension the reference uses as the base of its curve tower, src/utils/ecc.rs:407-466 = Fp2 of tower.cuh.]
// The trace is base-field; coefficients, the out-of-domain point, the DEEP composition and FRI are in E.
#include "ext.h"
#include "tower.cuh"

namespace cs {
namespace {

__device__ __forceinline__ Fp2 e_scale(Fp2 x, fp s) { return {fp_mul(x.a, s), fp_mul(x.b, s)}; }
__device__ inline Fp2 e_pow(Fp2 x, uint64_t e) {
    Fp2 r = {FP_ONE, 0};
    while (e) {
        if (e & 1) r = fp2_mul(r, x);
        x = fp2_sqr(x);
        e >>= 1;
    }
    return r;
}

constexpr int PE_SEG = 16384;
// grid = (segments, width): Horner in z^256 over one segment of one column, E accumulator, base coefficients
__global__ __launch_bounds__(256) void k_poly_eval_ext_partial(const fp *__restrict__ coeffs, size_t n, Fp2 z, fp *__restrict__ partial, unsigned seg_len) {
    __shared__ Fp2 part[256];
    const unsigned segs = gridDim.x, seg = blockIdx.x, col = blockIdx.y, t = threadIdx.x;
    const fp *c = coeffs + (size_t)col * n + (size_t)seg * seg_len;
    const Fp2 z256 = e_pow(z, 256);
    Fp2 acc = {0, 0};
    const unsigned per = (seg_len + 255) / 256;
    for (unsigned k = per; k-- > 0;) {
        const fp v = k * 256 + t < seg_len ? c[(size_t)k * 256 + t] : 0;
        acc = fp2_mul(acc, z256);
        acc.a = fp_add(acc.a, v);
    }
    part[t] = fp2_mul(acc, e_pow(z, (uint64_t)seg * seg_len + t));
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)t < s) part[t] = fp2_add(part[t], part[t + s]);
        __syncthreads();
    }
    if (t == 0) { partial[2 * ((size_t)col * segs + seg)] = part[0].a; partial[2 * ((size_t)col * segs + seg) + 1] = part[0].b; }
}
__global__ void k_poly_eval_ext_sum(const fp *__restrict__ partial, fp *__restrict__ out, unsigned width, unsigned segs) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x; // (column, component)
    if (i >= 2 * width) return;
    const unsigned col = i >> 1, k = i & 1;
    fp a = 0;
    for (unsigned s = 0; s < segs; s++) a = fp_add(a, partial[2 * ((size_t)col * segs + s) + k]);
    out[i] = a;
}

// 1 / (x - z) for base x: (a, b) = (x - z.a, -z.b); inverse = (a + 2b, -b) / (a^2 + 2ab - 2b^2)
struct EInv { Fp2 num; fp norm; };
__device__ __forceinline__ EInv e_inv_parts(fp x, const uint64_t zc[2]) {
    const fp a = fp_sub(x, zc[0]), b = fp_neg(zc[1]);
    const fp norm = fp_sub(fp_add(fp_sqr(a), fp_mul(fp_dbl(a), b)), fp_dbl(fp_sqr(b)));
    return {{fp_add(a, fp_dbl(b)), fp_neg(b)}, norm};
}

// grid = (ceil(n / 256), b)
__global__ __launch_bounds__(256) void k_deep_ext(DeepExtParams p) {
    const size_t n = (size_t)1 << p.log_n;
    const size_t j = blockIdx.x * (size_t)256 + threadIdx.x;
    if (j >= n) return;
    const unsigned k = blockIdx.y, b = 1u << p.log_b;
    const fp x = fp_mul(p.shifts[k], p.w[j]);
    // the three divisors with one base-field inversion
    const EInv q1 = e_inv_parts(x, p.z), q2 = e_inv_parts(x, p.zw), q3 = e_inv_parts(x, p.zb);
    const fp n12 = fp_mul(q1.norm, q2.norm);
    const fp inv = fp_inv(fp_mul(n12, q3.norm));
    const Fp2 i1 = e_scale(q1.num, fp_mul(inv, fp_mul(q2.norm, q3.norm))), i2 = e_scale(q2.num, fp_mul(inv, fp_mul(q1.norm, q3.norm))),
              i3 = e_scale(q3.num, fp_mul(inv, n12));
    // sum_c alpha_c T_c(x) and sum_c beta_c T_c(x): base values against E coefficients = component-wise dot products
    Acc128 s1a = acc_zero(), s1b = acc_zero(), s2a = acc_zero(), s2b = acc_zero();
    const fp *t = p.trace_lde + (size_t)k * p.width * n + j;
    const fp *al = p.coef, *be = p.coef + 2 * p.width, *de = p.coef + 4 * p.width;
    for (unsigned c = 0; c < p.width; c++) {
        const fp v = t[(size_t)c * n];
        acc_mad(s1a, al[2 * c], v); acc_mad(s1b, al[2 * c + 1], v);
        acc_mad(s2a, be[2 * c], v); acc_mad(s2b, be[2 * c + 1], v);
        if ((c & 3) == 3) { acc_fold(s1a); acc_fold(s1b); acc_fold(s2a); acc_fold(s2b); }
    }
    acc_fold(s1a); acc_fold(s1b); acc_fold(s2a); acc_fold(s2b);
    const Fp2 s1 = fp2_sub({acc_reduce(s1a), acc_reduce(s1b)}, {p.k1[0], p.k1[1]});
    const Fp2 s2 = fp2_sub({acc_reduce(s2a), acc_reduce(s2b)}, {p.k2[0], p.k2[1]});
    Fp2 s3 = {0, 0};
    const fp *h = p.comp_lde + (size_t)k * 2 * p.nb * n + j;
    for (unsigned i = 0; i < p.nb; i++) s3 = fp2_add(s3, fp2_mul({de[2 * i], de[2 * i + 1]}, {h[(size_t)(2 * i) * n], h[(size_t)(2 * i + 1) * n]}));
    s3 = fp2_sub(s3, {p.k3[0], p.k3[1]});
    Fp2 acc = fp2_add(fp2_add(fp2_mul(s1, i1), fp2_mul(s2, i2)), fp2_mul(s3, i3));
    acc = fp2_mul(acc, fp2_add({p.deg_a[0], p.deg_a[1]}, e_scale({p.deg_b[0], p.deg_b[1]}, x)));
    p.out[(size_t)k * n + j] = acc.a;
    p.out[((size_t)b + k) * n + j] = acc.b;
}

__global__ __launch_bounds__(256) void k_fri_fold4_ext(const fp *__restrict__ evals, fp *__restrict__ out, size_t q, const fp *__restrict__ winv,
                                                       fp offset_inv, Fp2 alpha, fp inv4) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i >= q) return;
    const size_t N = 4 * q;
    const fp zi = winv[q]; // zeta^-1
    Fp2 s[4];
#pragma unroll
    for (int comp = 0; comp < 2; comp++) { // the size-4 inverse DFT is linear over the base field: per component
        const fp *e = evals + comp * N;
        const fp v0 = e[i], v1 = e[i + q], v2 = e[i + 2 * q], v3 = e[i + 3 * q];
        const fp a = fp_add(v0, v2), b = fp_sub(v0, v2), c = fp_add(v1, v3), d = fp_mul(fp_sub(v1, v3), zi);
        const fp r0 = fp_add(a, c), r2 = fp_sub(a, c), r1 = fp_add(b, d), r3 = fp_sub(b, d);
        if (comp == 0) { s[0].a = r0; s[1].a = r1; s[2].a = r2; s[3].a = r3; }
        else { s[0].b = r0; s[1].b = r1; s[2].b = r2; s[3].b = r3; }
    }
    const Fp2 r = e_scale(alpha, fp_mul(offset_inv, winv[i]));
    const Fp2 r2 = fp2_sqr(r), r3 = fp2_mul(r2, r);
    Fp2 acc = fp2_add(fp2_add(s[0], fp2_mul(r, s[1])), fp2_add(fp2_mul(r2, s[2]), fp2_mul(r3, s[3])));
    acc = e_scale(acc, inv4);
    out[i] = acc.a;
    out[q + i] = acc.b;
}

} // namespace

size_t poly_eval_ext_scratch_words(unsigned width, unsigned log_n) {
    const size_t n = (size_t)1 << log_n, seg = n < (size_t)PE_SEG ? n : (size_t)PE_SEG;
    return 2 * (size_t)width * (n / seg);
}
hipError_t poly_eval_ext(const uint64_t *d_coeffs, unsigned width, unsigned log_n, uint64_t za, uint64_t zb, uint64_t *d_out, uint64_t *d_scratch,
                         hipStream_t stream) {
    const size_t n = (size_t)1 << log_n;
    const unsigned seg_len = (unsigned)(n < (size_t)PE_SEG ? n : (size_t)PE_SEG), segs = (unsigned)(n / seg_len);
    hipLaunchKernelGGL(k_poly_eval_ext_partial, dim3(segs, width), dim3(256), 0, stream, d_coeffs, n, Fp2{za, zb}, d_scratch, seg_len);
    hipLaunchKernelGGL(k_poly_eval_ext_sum, dim3((2 * width + 255) / 256), dim3(256), 0, stream, d_scratch, d_out, width, segs);
    return hipGetLastError();
}
hipError_t deep_composition_ext(const DeepExtParams &p, hipStream_t stream) {
    const size_t n = (size_t)1 << p.log_n;
    hipLaunchKernelGGL(k_deep_ext, dim3((unsigned)((n + 255) / 256), 1u << p.log_b), dim3(256), 0, stream, p);
    return hipGetLastError();
}
hipError_t fri_fold4_ext(const uint64_t *d_evals, uint64_t *d_out, unsigned log_n, const uint64_t *d_winv, uint64_t offset_inv, uint64_t alpha_a,
                         uint64_t alpha_b, uint64_t inv4, hipStream_t stream) {
    const size_t q = ((size_t)1 << log_n) / 4;
    hipLaunchKernelGGL(k_fri_fold4_ext, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, stream, d_evals, d_out, q, d_winv, offset_inv,
                       Fp2{alpha_a, alpha_b}, inv4);
    return hipGetLastError();
}

} // namespace cs
