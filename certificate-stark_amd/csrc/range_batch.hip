// Device stages of the BATCHED range prover: B reference-shaped range proofs (RangeProver / RangeProofAir, /root/reference/
// src/range/prover.rs:24-43, src/range/air.rs:60-105: 64 rows x 2 registers each) per launch.  One 64-row proof is ~150 launches and host
// round trips of a few microseconds of GPU work each (0.33 ms per proof, host-API bound); here every stage is ONE launch over the batch
// and the host walks B Fiat-Shamir channels between the stages.  Exact field arithmetic: every proof equals cstark_air_prove's bytes.
// Interpolation / extension of the 2 B columns go through the generic transform kernels (ntt.hip); the stages below are the per-proof
// parts that have no batched form elsewhere.
#include "range_batch.h"
#include "fp.cuh"

namespace cs {
namespace {

// RangeProver::build_trace for proof t (k_trace_range, trace_gen.hip): row q holds bit (62 - (q - 1)) and the top q bits of the value
__global__ void k_rb_trace(const uint64_t *__restrict__ numbers, fp *__restrict__ trace, unsigned batch) {
    const unsigned t = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, q = threadIdx.x % 64;
    if (t >= batch) return;
    const uint64_t v63 = numbers[t] & 0x7FFFFFFFFFFFFFFFULL;
    const uint64_t acc = q == 0 ? 0 : v63 >> (63 - q);
    trace[(size_t)(2 * t) * 64 + q] = (q >= 1 && (acc & 1)) ? FP_ONE : 0;
    trace[(size_t)(2 * t + 1) * 64 + q] = fp_from_u64(acc);
}

// RangeProofAir::evaluate_transition (src/range/air.rs:60-98) merged with its two assertions (:79-86) as the engine's evaluator does
// (k_eval_transitions_range + k_air_combine, constraints.hip): thread = (proof t, coset kk of the evaluation domain, row j)
__global__ __launch_bounds__(128) void k_rb_combine(RangeBatchConsts c, const fp *__restrict__ lde, const fp *__restrict__ coefs, const fp *__restrict__ numbers,
                                                    fp *__restrict__ out, unsigned batch) {
    const unsigned t = blockIdx.x, kk = threadIdx.x / 64, j = threadIdx.x % 64, k = 4 * kk;
    const size_t W = 2 * (size_t)batch;
    const fp *bitc = lde + ((size_t)k * W + 2 * t) * 64, *accc = bitc + 64;
    const unsigned jn = (j + 1) & 63;
    const fp nb = bitc[jn];
    const fp c0 = fp_sub(fp_sqr(nb), nb);                                  // result[0]: the bit is binary
    const fp c1 = fp_sub(accc[jn], fp_add(fp_dbl(accc[j]), nb));           // result[1]: acc' = 2 acc + bit
    const fp *cf = coefs + (size_t)t * 8;
    const fp x = fp_mul(c.shift[k], c.w64[j]);
    fp acc = fp_add(fp_mul(c0, fp_add(cf[0], fp_mul(cf[2], fp_pow(x, c.adj[0])))), fp_mul(c1, fp_add(cf[1], fp_mul(cf[3], fp_pow(x, c.adj[1])))));
    acc = fp_mul(acc, fp_mul(fp_sub(x, c.w_last), c.zinv[k]));
    const fp xb = fp_pow(x, c.badj);
    const fp tv = accc[j];
    acc = fp_add(acc, fp_mul(fp_mul(tv, fp_add(cf[4], fp_mul(cf[6], xb))), fp_inv(fp_sub(x, FP_ONE))));                       // acc[0] = 0
    acc = fp_add(acc, fp_mul(fp_mul(fp_sub(tv, numbers[t]), fp_add(cf[5], fp_mul(cf[7], xb))), fp_inv(fp_sub(x, c.w_last)))); // acc[63] = number
    out[((size_t)t * 2 + kk) * 64 + j] = acc;
}

// in-place radix-2 transform of M = 2^log_m points in LDS (bit-reversal, then log_m decimation-in-time stages); tw[e] = root^e for
// e < M / 2.  M / 2 threads.
__device__ __forceinline__ void lds_fft(fp *a, unsigned log_m, const fp *__restrict__ tw, unsigned tid) {
    const unsigned M = 1u << log_m;
    for (unsigned i = tid; i < M; i += M / 2) {
        const unsigned r = __brev(i) >> (32 - log_m);
        if (i < r) { const fp x = a[i]; a[i] = a[r]; a[r] = x; }
    }
    __syncthreads();
    for (unsigned s = 0; s < log_m; s++) {
        const unsigned half = 1u << s, jj = tid & (half - 1), i0 = ((tid >> s) << (s + 1)) + jj, i1 = i0 + half;
        const fp u = a[i0], v = fp_mul(a[i1], tw[jj << (log_m - 1 - s)]);
        a[i0] = fp_add(u, v);
        a[i1] = fp_sub(u, v);
        __syncthreads();
    }
}
// composition polynomial: the 128 merged evaluations of proof t over g <w_128> (point 2 j + kk = row j of coset 4 kk) -> coefficients
// of H(g y) in y -> H in x (times g^-m) -> columns H_i[q] = h[2 q + i]   (cstark_composition_columns for one proof)
__global__ __launch_bounds__(64) void k_rb_composition(RangeBatchConsts c, const fp *__restrict__ combined, fp *__restrict__ ccoef) {
    __shared__ fp a[128];
    const unsigned t = blockIdx.x, tid = threadIdx.x;
    for (unsigned i = tid; i < 128; i += 64) a[i] = combined[((size_t)t * 2 + (i & 1)) * 64 + (i >> 1)];
    __syncthreads();
    lds_fft(a, 7, c.winv128, tid);
    for (unsigned m = tid; m < 128; m += 64) {
        const fp h = fp_mul(fp_mul(a[m], c.inv128), fp_pow(c.ginv, m));
        ccoef[((size_t)2 * t + (m & 1)) * 64 + (m >> 1)] = h;
    }
}

// out-of-domain frame of proof t (cstark_evaluate_polys_at for its own z): lane v of 6 runs one Horner chain over 64 coefficients
__global__ void k_rb_ood(RangeBatchConsts c, const fp *__restrict__ coeffs, const fp *__restrict__ ccoef, const fp *__restrict__ z, fp *__restrict__ out, unsigned batch) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x, t = idx / 6, v = idx % 6;
    if (t >= batch) return;
    const fp zt = z[t];
    const fp pt = v < 2 ? zt : v < 4 ? fp_mul(zt, c.w64[1]) : fp_sqr(zt);
    const fp *col = (v < 4 ? coeffs : ccoef) + ((size_t)2 * t + (v & 1)) * 64;
    fp acc = 0;
    for (int m = 63; m >= 0; m--) acc = fp_add(fp_mul(acc, pt), col[m]);
    out[(size_t)t * 6 + v] = acc;
}

// DEEP composition (k_deep, deep.hip) at LDE point i = 8 j + k of proof t, written in natural order
__global__ __launch_bounds__(256) void k_rb_deep(RangeBatchConsts c, const fp *__restrict__ lde, const fp *__restrict__ clde, const fp *__restrict__ z,
                                                 const fp *__restrict__ ood, const fp *__restrict__ dcoef, fp *__restrict__ layer, unsigned batch) {
    const unsigned t = blockIdx.x, W = 2 * batch;
    for (unsigned i = threadIdx.x; i < 512; i += 256) {
        const unsigned k = i & 7, j = i >> 3;
        const fp x = fp_mul(c.shift[k], c.w64[j]);
        const fp zt = z[t], zw = fp_mul(zt, c.w64[1]), zb = fp_sqr(zt);
        const fp d1 = fp_sub(x, zt), d2 = fp_sub(x, zw), d3 = fp_sub(x, zb);
        const fp inv = fp_inv(fp_mul(fp_mul(d1, d2), d3));
        const fp i1 = fp_mul(inv, fp_mul(d2, d3)), i2 = fp_mul(inv, fp_mul(d1, d3)), i3 = fp_mul(inv, fp_mul(d1, d2));
        const fp *o = ood + (size_t)t * 6, *cf = dcoef + (size_t)t * 8;
        const fp t0 = lde[((size_t)k * W + 2 * t) * 64 + j], t1 = lde[((size_t)k * W + 2 * t + 1) * 64 + j];
        const fp h0 = clde[((size_t)k * W + 2 * t) * 64 + j], h1 = clde[((size_t)k * W + 2 * t + 1) * 64 + j];
        const fp s1 = fp_add(fp_mul(cf[0], fp_sub(t0, o[0])), fp_mul(cf[1], fp_sub(t1, o[1])));
        const fp s2 = fp_add(fp_mul(cf[2], fp_sub(t0, o[2])), fp_mul(cf[3], fp_sub(t1, o[3])));
        const fp s3 = fp_add(fp_mul(cf[4], fp_sub(h0, o[4])), fp_mul(cf[5], fp_sub(h1, o[5])));
        const fp acc = fp_add(fp_add(fp_mul(s1, i1), fp_mul(s2, i2)), fp_mul(s3, i3));
        layer[(size_t)t * 512 + i] = fp_mul(acc, fp_add(cf[6], fp_mul(cf[7], x)));
    }
}

// FRI folding by 4 (k_fri_fold4, deep.hip) of proof t's 512 evaluations over g <w_512> with its own alpha
__global__ __launch_bounds__(128) void k_rb_fold(RangeBatchConsts c, const fp *__restrict__ layer, const fp *__restrict__ alpha, fp *__restrict__ out) {
    const unsigned t = blockIdx.x, i = threadIdx.x, q = 128;
    const fp *e = layer + (size_t)t * 512;
    const fp v0 = e[i], v1 = e[i + q], v2 = e[i + 2 * q], v3 = e[i + 3 * q];
    const fp zi = c.winv512[q];
    const fp a = fp_add(v0, v2), b = fp_sub(v0, v2), cc = fp_add(v1, v3), d = fp_mul(fp_sub(v1, v3), zi);
    const fp s0 = fp_add(a, cc), s2 = fp_sub(a, cc), s1 = fp_add(b, d), s3 = fp_sub(b, d);
    const fp r = fp_mul(alpha[t], fp_mul(c.offset_inv, c.winv512[i]));
    const fp r2 = fp_sqr(r), r3 = fp_mul(r2, r);
    const fp acc = fp_add(fp_add(s0, fp_mul(r, s1)), fp_add(fp_mul(r2, s2), fp_mul(r3, s3)));
    out[(size_t)t * 128 + i] = fp_mul(acc, c.inv4);
}

// slot layout (bytes): trace rows [nq][16] | trace paths [nq][9][32] | composition rows [nq][16] | paths [nq][9][32] | layer rows
// [nq][32] | layer paths [nq][7][32]   (the last two hold lcount[t] entries)
__global__ __launch_bounds__(64) void k_rb_open(RangeBatchOpen o) {
    const unsigned t = blockIdx.y, qi = blockIdx.x, W = 2 * o.batch, nq = o.nq;
    uint8_t *slot = o.out + (size_t)t * o.slot;
    const size_t o_trows = 0, o_tpath = o_trows + (size_t)nq * 16, o_crows = o_tpath + (size_t)nq * 288, o_cpath = o_crows + (size_t)nq * 16,
                 o_lrows = o_cpath + (size_t)nq * 288, o_lpath = o_lrows + (size_t)nq * 32;
    const unsigned pos = o.pos[(size_t)t * nq + qi], k = pos & 7, j = pos >> 3, lane = threadIdx.x;
    if (lane < 2) {
        reinterpret_cast<uint64_t *>(slot + o_trows)[qi * 2 + lane] = o.lde[((size_t)k * W + 2 * t + lane) * 64 + j];
        reinterpret_cast<uint64_t *>(slot + o_crows)[qi * 2 + lane] = o.clde[((size_t)k * W + 2 * t + lane) * 64 + j];
    }
    for (unsigned w = lane; w < 2 * 9 * 2; w += 64) { // two trees x 9 levels x two 16-byte halves
        const unsigned tree = w / 18, r = w % 18, lvl = r >> 1, half = r & 1;
        const uint4 *nodes = reinterpret_cast<const uint4 *>((tree ? o.cnodes : o.tnodes) + (size_t)t * 1024 * 32);
        const size_t node = ((512u + pos) >> lvl) ^ 1;
        reinterpret_cast<uint4 *>(slot + (tree ? o_cpath : o_tpath))[((size_t)qi * 9 + lvl) * 2 + half] = nodes[2 * node + half];
    }
    if (o.n_layers && qi < o.lcount[t]) {
        const unsigned lp = o.lpos[(size_t)t * nq + qi];
        if (lane < 4) reinterpret_cast<uint64_t *>(slot + o_lrows)[qi * 4 + lane] = o.layer[(size_t)t * 512 + lp + 128 * lane];
        const uint4 *nodes = reinterpret_cast<const uint4 *>(o.lnodes + (size_t)t * 256 * 32);
        for (unsigned w = lane; w < 7 * 2; w += 64) {
            const unsigned lvl = w >> 1, half = w & 1;
            const size_t node = ((128u + lp) >> lvl) ^ 1;
            reinterpret_cast<uint4 *>(slot + o_lpath)[((size_t)qi * 7 + lvl) * 2 + half] = nodes[2 * node + half];
        }
    }
}

} // namespace

hipError_t rb_trace(const uint64_t *d_numbers_canonical, uint64_t *d_trace, unsigned batch, hipStream_t stream) {
    hipLaunchKernelGGL(k_rb_trace, dim3((batch + 3) / 4), dim3(256), 0, stream, d_numbers_canonical, d_trace, batch);
    return hipGetLastError();
}
hipError_t rb_combine(const RangeBatchConsts &c, const uint64_t *d_lde, const uint64_t *d_coefs, const uint64_t *d_numbers, uint64_t *d_out, unsigned batch,
                      hipStream_t stream) {
    hipLaunchKernelGGL(k_rb_combine, dim3(batch), dim3(128), 0, stream, c, d_lde, d_coefs, d_numbers, d_out, batch);
    return hipGetLastError();
}
hipError_t rb_composition(const RangeBatchConsts &c, const uint64_t *d_combined, uint64_t *d_ccoef, unsigned batch, hipStream_t stream) {
    hipLaunchKernelGGL(k_rb_composition, dim3(batch), dim3(64), 0, stream, c, d_combined, d_ccoef);
    return hipGetLastError();
}
hipError_t rb_ood(const RangeBatchConsts &c, const uint64_t *d_coeffs, const uint64_t *d_ccoef, const uint64_t *d_z, uint64_t *d_out, unsigned batch, hipStream_t stream) {
    hipLaunchKernelGGL(k_rb_ood, dim3((6 * batch + 255) / 256), dim3(256), 0, stream, c, d_coeffs, d_ccoef, d_z, d_out, batch);
    return hipGetLastError();
}
hipError_t rb_deep(const RangeBatchConsts &c, const uint64_t *d_lde, const uint64_t *d_clde, const uint64_t *d_z, const uint64_t *d_ood, const uint64_t *d_dcoef,
                   uint64_t *d_layer, unsigned batch, hipStream_t stream) {
    hipLaunchKernelGGL(k_rb_deep, dim3(batch), dim3(256), 0, stream, c, d_lde, d_clde, d_z, d_ood, d_dcoef, d_layer, batch);
    return hipGetLastError();
}
hipError_t rb_fold(const RangeBatchConsts &c, const uint64_t *d_layer, const uint64_t *d_alpha, uint64_t *d_out, unsigned batch, hipStream_t stream) {
    hipLaunchKernelGGL(k_rb_fold, dim3(batch), dim3(128), 0, stream, c, d_layer, d_alpha, d_out);
    return hipGetLastError();
}
hipError_t rb_open(const RangeBatchOpen &o, hipStream_t stream) {
    hipLaunchKernelGGL(k_rb_open, dim3(o.nq, o.batch), dim3(64), 0, stream, o);
    return hipGetLastError();
}

} // namespace cs
