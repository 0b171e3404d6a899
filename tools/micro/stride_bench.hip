// Micro-benchmark: lane-per-row reads of W column streams whose bases are `stride` elements apart (the access shape of k_hash_rows and
// of the lane-per-point constraint kernels).  Does a power-of-two column stride (2^20 elements = 8 MB) cost HBM bandwidth?
//   hipcc --offload-arch=gfx950 -O3 -o stride_bench tools/micro/stride_bench.hip && ./stride_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void k_read_cols(const uint64_t *__restrict__ base, uint64_t *__restrict__ out, size_t n, size_t stride, int w) {
    const size_t j = blockIdx.x * (size_t)256 + threadIdx.x;
    if (j >= n) return;
    uint64_t acc = 0;
    for (int c0 = 0; c0 < w; c0 += 8) {
        uint64_t v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = c0 + i < w ? base[(size_t)(c0 + i) * stride + j] : 0;
#pragma unroll
        for (int i = 0; i < 8; i++) acc ^= v[i] + (acc << 1);
    }
    out[j] = acc;
}

int main() {
    const size_t n = (size_t)1 << 20;
    const int w = 94, cosets = 8;
    const size_t max_stride = n + 8192;
    uint64_t *buf, *out;
    hipMalloc(&buf, (size_t)cosets * w * max_stride * 8);
    hipMalloc(&out, (size_t)cosets * n * 8);
    hipMemset(buf, 1, (size_t)cosets * w * max_stride * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t pads[] = {0, 32, 64, 128, 512, 1024, 2048, 4096 + 32};
    for (size_t pad : pads) {
        const size_t stride = n + pad;
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            for (int k = 0; k < cosets; k++)
                hipLaunchKernelGGL(k_read_cols, dim3((unsigned)(n / 256)), dim3(256), 0, 0, buf + (size_t)k * w * stride, out + (size_t)k * n, n, stride, w);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("column stride 2^20 + %5zu elements: %.3f ms for %.2f GB -> %.0f GB/s\n", pad, best, cosets * w * n * 8 / 1e9, cosets * w * n * 8 / 1e6 / best);
    }
    return 0;
}
