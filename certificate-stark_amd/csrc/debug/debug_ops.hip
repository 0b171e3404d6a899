// Element-wise test/benchmark kernels for the device field arithmetic (fp.cuh, tower.cuh), exposed through
// cstark_debug_* so the GPU parity tests can pin every primitive against the oracle, and so the modular
// multiplication rate of the chip can be measured in isolation.
#include <hip/hip_runtime.h>
#include "../../../include/cstark.h"
#include "../tower.cuh"

namespace cs {
namespace {

__global__ void k_fp_op(const fp *a, const fp *b, fp *out, size_t n, int op) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    fp x = a[i], y = b ? b[i] : 0, r = 0;
    switch (op) {
        case 0: r = fp_mul(x, y); break;
        case 1: r = fp_add(x, y); break;
        case 2: r = fp_sub(x, y); break;
        case 3: r = fp_inv(x); break;
        case 4: r = fp_inv_sbox(x); break;
        case 5: r = fp_from_u64(x); break;
        case 6: r = fp_to_u64(x); break;
        case 7: r = fp_neg(x); break;
        case 8: r = fp_dbl(x); break;
        case 9: r = wave_next(x, y); break; // lane l + 1's x, lane 63: its own y
        case 10: r = fp_mul(fp_sub_lazy(x, y), y); break; // the product takes an unreduced first factor in (0, 2p): (x - y) * y
        case 11: { Acc128 acc; acc.lo = x; acc.hi = y; r = acc_reduce_below_p(acc); break; } // (y 2^64 + x) / 2^64 mod p, y <= p - 2^32
        case 12: { Acc128 acc; acc.lo = x; acc.hi = y; r = acc_reduce(acc); break; }         // the same for y < 2p
    }
    out[i] = r;
}

// n independent F_p6 operations, operands stored [i][6]
__global__ void k_fp6_op(const fp *a, const fp *b, fp *out, size_t n, int op) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp6 x = fp6_load(a + 6 * i), r;
    if (op == 0) r = fp6_mul(x, fp6_load(b + 6 * i));
    else if (op == 1) r = fp6_sqr(x);
    else r = fp6_inv(x);
    fp6_store(out + 6 * i, r);
}

// ILP independent Montgomery-product chains per lane; result written so nothing is optimised away
template <int ILP>
__global__ __launch_bounds__(256) void k_modmul_bench(fp *out, int iters, fp seed) {
    fp x[ILP], y = seed | 1;
#pragma unroll
    for (int k = 0; k < ILP; k++) x[k] = seed + threadIdx.x + 977 * k + blockIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) x[k] = fp_mul(x[k], y);
    }
    fp acc = 0;
#pragma unroll
    for (int k = 0; k < ILP; k++) acc ^= x[k];
    out[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = acc;
}

} // namespace
} // namespace cs

extern "C" {

int cstark_debug_fp_op(void *stream, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t n, int op) {
    if (n == 0) return CSTARK_OK;
    hipLaunchKernelGGL(cs::k_fp_op, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_a, d_b, d_out, n, op);
    return hipGetLastError() == hipSuccess ? CSTARK_OK : CSTARK_ERR_HIP;
}
int cstark_debug_fp6_op(void *stream, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t n, int op) {
    if (n == 0) return CSTARK_OK;
    hipLaunchKernelGGL(cs::k_fp6_op, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_a, d_b, d_out, n, op);
    return hipGetLastError() == hipSuccess ? CSTARK_OK : CSTARK_ERR_HIP;
}
// Runs blocks x 256 lanes x ilp chains x iters Montgomery products; returns the elapsed milliseconds.
int cstark_debug_modmul_bench(void *stream, uint64_t *d_out, int blocks, int iters, int ilp, float *ms) {
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return CSTARK_ERR_HIP;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0, s);
        if (ilp == 1) hipLaunchKernelGGL(cs::k_modmul_bench<1>, dim3(blocks), dim3(256), 0, s, d_out, iters, (uint64_t)12345);
        else if (ilp == 2) hipLaunchKernelGGL(cs::k_modmul_bench<2>, dim3(blocks), dim3(256), 0, s, d_out, iters, (uint64_t)12345);
        else if (ilp == 4) hipLaunchKernelGGL(cs::k_modmul_bench<4>, dim3(blocks), dim3(256), 0, s, d_out, iters, (uint64_t)12345);
        else hipLaunchKernelGGL(cs::k_modmul_bench<8>, dim3(blocks), dim3(256), 0, s, d_out, iters, (uint64_t)12345);
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess) return CSTARK_ERR_HIP;
    }
    (void)hipEventElapsedTime(ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return CSTARK_OK;
}
}

// ---- experiment: INV_MDS * d on the matrix cores vs the limb dot products -------------------------------------------------------
#include "../mds_mfma.cuh"
#include "../rescue.cuh"
namespace cs {
namespace {
__global__ void k_mds_table(uint8_t *tab) { mdsmfma::build_table_entry(tab, c_inv_mds, 14, blockIdx.x, threadIdx.x); }

// in / out: [14][npts] column-major; one wave per 64 points; blockDim = 256
__global__ __launch_bounds__(256) void k_mds_mfma(const uint8_t *__restrict__ gtab, const fp *__restrict__ in, fp *__restrict__ out, size_t npts, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr size_t TB = mdsmfma::table_bytes(14);
    uint8_t *tab = smem, *stage = smem + TB + (size_t)(threadIdx.x >> 6) * 64 * mdsmfma::ROW_BYTES;
    for (size_t i = threadIdx.x * 16; i < TB; i += blockDim.x * 16) *(uint4 *)(tab + i) = *(const uint4 *)(gtab + i);
    const int lane = threadIdx.x & 63;
    const size_t base = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) & ~(size_t)63, p = base + lane;
    fp x[14];
#pragma unroll
    for (int j = 0; j < 14; j++) x[j] = in[(size_t)j * npts + p];
    __syncthreads();
    fp keep[7][2];
    for (int it = 0; it < iters; it++) {
        mdsmfma::stage_vector(stage, lane, x);
        __syncthreads();
        mdsmfma::BFrags b;
        mdsmfma::load_bfrags(b, stage, lane);
#pragma unroll
        for (int T = 0; T < 7; T++) mdsmfma::tile_product(tab, 7, T, b, lane, keep[T]);
        __syncthreads();
        if (it + 1 < iters) { // feed something back so the loop is not collapsed
#pragma unroll
            for (int T = 0; T < 7; T++) x[T] = keep[T][0];
        }
    }
    const int n = lane & 31, g = lane >> 5;
#pragma unroll
    for (int T = 0; T < 7; T++) {
        out[(size_t)(2 * T + g) * npts + base + n] = keep[T][0];
        out[(size_t)(2 * T + g) * npts + base + 32 + n] = keep[T][1];
    }
}
// the same product with plain field arithmetic (reference and VALU timing)
__global__ __launch_bounds__(256) void k_mds_valu(const fp *__restrict__ in, fp *__restrict__ out, size_t npts, int iters) {
    const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    fp x[14], y[14];
#pragma unroll
    for (int j = 0; j < 14; j++) x[j] = in[(size_t)j * npts + p];
    for (int it = 0; it < iters; it++) {
#pragma unroll 1
        for (int i = 0; i < 14; i++) {
            Acc128 a = acc_zero();
#pragma unroll
            for (int j = 0; j < 7; j++) acc_mad(a, c_inv_mds[i * 14 + j], x[j]);
            acc_fold(a);
#pragma unroll
            for (int j = 7; j < 14; j++) acc_mad(a, c_inv_mds[i * 14 + j], x[j]);
            acc_fold(a);
            y[i] = acc_reduce(a);
        }
        if (it + 1 < iters)
#pragma unroll
            for (int T = 0; T < 7; T++) x[T] = y[2 * T + ((threadIdx.x >> 5) & 1)];
    }
#pragma unroll
    for (int i = 0; i < 14; i++) out[(size_t)i * npts + p] = y[i];
}
} // namespace
} // namespace cs

extern "C" int cstark_debug_mds(void *stream, const uint64_t *d_in, uint64_t *d_out, size_t npts, int use_mfma, int iters, float *ms) {
    hipStream_t s = (hipStream_t)stream;
    if (npts % 256) return CSTARK_ERR_INVALID_ARG;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return CSTARK_ERR_HIP;
    uint8_t *tab = nullptr;
    const size_t TB = cs::mdsmfma::table_bytes(14);
    if (hipMalloc((void **)&tab, TB) != hipSuccess) return CSTARK_ERR_OOM;
    (void)hipMemsetAsync(tab, 0, TB, s);
    hipLaunchKernelGGL(cs::k_mds_table, dim3(14), dim3(16), 0, s, tab);
    const size_t lds = TB + 4 * 64 * cs::mdsmfma::ROW_BYTES;
    (void)hipFuncSetAttribute((const void *)cs::k_mds_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0, s);
        if (use_mfma) hipLaunchKernelGGL(cs::k_mds_mfma, dim3((unsigned)(npts / 256)), dim3(256), lds, s, tab, d_in, d_out, npts, iters);
        else hipLaunchKernelGGL(cs::k_mds_valu, dim3((unsigned)(npts / 256)), dim3(256), 0, s, d_in, d_out, npts, iters);
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess) { (void)hipFree(tab); return CSTARK_ERR_HIP; }
    }
    (void)hipEventElapsedTime(ms, e0, e1);
    const hipError_t err = hipGetLastError();
    (void)hipFree(tab);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return err == hipSuccess ? CSTARK_OK : CSTARK_ERR_HIP;
}
