// Micro-benchmark: what bandwidth does the lane-per-row read of a 94-column table reach, and is it the column-major layout (94 streams
// 8 MB apart per workgroup) that limits it?  Three layouts of the same 4 x 94 x 2^20 table, the same arithmetic on what is read:
//   columns   [coset][column][n]                      what the transforms write today
//   tiles     [coset][n / T][column][T]               a workgroup's rows are one contiguous 94 T-element block
//   columns, padded stride (the 2^20 + 32 of stride_bench)
// and for each, 8 or 16 loads in flight per lane.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/rowread_bench tools/micro/rowread_bench.hip && /tmp/rowread_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int B, int T>
__global__ __launch_bounds__(256) void k_read(const uint64_t *__restrict__ base, uint64_t *__restrict__ out, size_t n, size_t stride, int w) {
    const size_t j = blockIdx.x * (size_t)256 + threadIdx.x;
    const uint64_t *p = T == 0 ? base + j : base + (j / T) * (size_t)w * T + (j % T);
    const size_t cs = T == 0 ? stride : (size_t)T;
    uint64_t acc = 0;
    for (int c0 = 0; c0 < w; c0 += B) {
        uint64_t v[B];
#pragma unroll
        for (int i = 0; i < B; i++) v[i] = c0 + i < w ? p[(size_t)(c0 + i) * cs] : 0;
#pragma unroll
        for (int i = 0; i < B; i++) acc ^= v[i] + (acc << 1);
    }
    out[j] = acc;
}

template <int B, int T>
static void run(const char *name, const uint64_t *buf, uint64_t *out, size_t n, size_t stride, int w, int cosets) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(e0);
        for (int k = 0; k < cosets; k++)
            hipLaunchKernelGGL((k_read<B, T>), dim3((unsigned)(n / 256)), dim3(256), 0, 0, buf + (size_t)k * w * stride, out + (size_t)k * n, n, stride, w);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-34s %2d loads in flight: %.3f ms for %.2f GB -> %.0f GB/s\n", name, B, best, cosets * (double)w * n * 8 / 1e9, cosets * (double)w * n * 8 / 1e6 / best);
}

int main() {
    const size_t n = (size_t)1 << 20;
    const int w = 94, cosets = 4;
    const size_t max_stride = n + 8192;
    uint64_t *buf, *out;
    hipMalloc(&buf, (size_t)cosets * w * max_stride * 8);
    hipMalloc(&out, (size_t)cosets * n * 8);
    hipMemset(buf, 1, (size_t)cosets * w * max_stride * 8);
    run<8, 0>("columns, stride 2^20", buf, out, n, n, w, cosets);
    run<16, 0>("columns, stride 2^20", buf, out, n, n, w, cosets);
    run<24, 0>("columns, stride 2^20", buf, out, n, n, w, cosets);
    run<8, 0>("columns, stride 2^20 + 32", buf, out, n, n + 32, w, cosets);
    run<16, 0>("columns, stride 2^20 + 32", buf, out, n, n + 32, w, cosets);
    run<8, 0>("columns, stride 2^20 + 4128", buf, out, n, n + 4128, w, cosets);
    run<8, 256>("tiles of 256 rows", buf, out, n, n, w, cosets);
    run<16, 256>("tiles of 256 rows", buf, out, n, n, w, cosets);
    run<8, 64>("tiles of 64 rows", buf, out, n, n, w, cosets);
    run<16, 64>("tiles of 64 rows", buf, out, n, n, w, cosets);
    run<8, 1024>("tiles of 1024 rows", buf, out, n, n, w, cosets);
    run<16, 1024>("tiles of 1024 rows", buf, out, n, n, w, cosets);
    return 0;
}
