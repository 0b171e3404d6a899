// Element-wise test/benchmark kernels for the device field arithmetic (fp.cuh, tower.cuh), exposed through
// cstark_debug_* so the GPU parity tests can pin every primitive against the oracle, and so the modular
// multiplication rate of the chip can be measured in isolation.
#include <hip/hip_runtime.h>
#include "../../include/cstark.h"
#include "tower.cuh"

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
