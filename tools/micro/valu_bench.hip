// Micro-benchmark: issue cost of the vector instructions the field arithmetic is made of, on gfx950.  Every lane runs ILP independent
// dependent chains of ONE instruction (inline asm, so the compiler neither fuses nor removes them); blocks of 256 threads,
// `wps` waves per SIMD.  Prints SIMD cycles per wave-instruction at the clock the kernel sustained (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_bench tools/micro/valu_bench.hip && /tmp/valu_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int ILP = 8, UNROLL = 16;

template <int OP>
__global__ __launch_bounds__(256) void k_chain(uint64_t *out, int iters, uint32_t seed, unsigned long long *clk) {
    uint32_t a[ILP], b[ILP];
    uint64_t w[ILP];
    for (int i = 0; i < ILP; i++) { a[i] = seed + threadIdx.x * 7 + i; b[i] = seed * 3 + i + blockIdx.x; w[i] = ((uint64_t)a[i] << 32) | b[i]; }
    const uint32_t c = seed | 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(a[i]));
                if (OP == 3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
                if (OP == 4) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(c) : "vcc");
                if (OP == 5) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"((uint64_t)c));
                if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : );
                if (OP == 7) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 8) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 9) asm volatile("v_not_b32 %0, %0" : "+v"(a[i]));
                if (OP == 10) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(c) : "vcc");
                if (OP == 11) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(w[i]), "v"((uint64_t)c) : "vcc");
                if (OP == 12) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 13) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 14) asm volatile("v_cmp_gt_i32 vcc, 0, %0" : : "v"(a[i]) : "vcc");
                if (OP == 15) asm volatile("v_cmp_gt_i32_e64 s[20:21], 0, %0" : : "v"(a[i]) : "s20", "s21");
                if (OP == 16) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 17) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(c));
                if (OP == 18) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[i]));
                if (OP == 19) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
                if (OP == 20) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
                if (OP == 21) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(w[i]));
                if (OP == 22) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : "vcc");
                if (OP == 23) asm volatile("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(w[i]) : "v"(a[i]) : "vcc");
                if (OP == 24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
                if (OP == 25) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 26) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(c));
                if (OP == 27) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 28) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(w[i]) : "v"((uint64_t)c));
                if (OP == 29) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint64_t acc = 0;
    for (int i = 0; i < ILP; i++) acc ^= a[i] + w[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int OP>
void run(const char *name, uint64_t *out, unsigned long long *clk) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_chain<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u, clk);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        unsigned long long h[2];
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9; // s_memrealtime ticks at 100 MHz
        const double instr_per_simd = (double)blocks * 4 / 1024.0 * iters * UNROLL * ILP; // waves per SIMD x instructions per wave
        printf("%-16s %d waves/SIMD: %7.3f ms  clock %.2f GHz  %.2f SIMD cycles per wave-instruction\n", name, wps, ms, ghz,
               ms * 1e-3 * ghz * 1e9 / instr_per_simd);
    }
}

int main() {
    uint64_t *out; unsigned long long *clk;
    hipMalloc(&out, 256 * 8 * 256 * 8); hipMalloc(&clk, 16);
    run<0>("v_add_u32", out, clk);
    run<1>("v_xor_b32", out, clk);
    run<2>("v_alignbit_b32", out, clk);
    run<3>("v_add3_u32", out, clk);
    run<9>("v_not_b32", out, clk);
    run<12>("v_mov_b32", out, clk);
    run<6>("v_cndmask_b32", out, clk);
    run<10>("v_sub_co_u32", out, clk);
    run<5>("v_lshl_add_u64", out, clk);
    run<11>("v_cmp_lt_u64", out, clk);
    run<7>("v_mul_lo_u32", out, clk);
    run<8>("v_mul_hi_u32", out, clk);
    run<4>("v_mad_u64_u32", out, clk);
    run<13>("v_pk_add_u16", out, clk);
    run<14>("v_cmp_gt_i32 vcc", out, clk);
    run<15>("v_cmp_gt_i32 sgpr", out, clk);
    run<16>("v_min_u32", out, clk);
    run<17>("v_cndmask e64", out, clk);
    run<18>("v_ashrrev_i32", out, clk);
    run<19>("v_and_or_b32", out, clk);
    run<20>("v_bfi_b32", out, clk);
    run<21>("v_lshlrev_b64", out, clk);
    run<22>("v_addc_co_u32", out, clk);
    run<23>("v_mad_u64 x*1+c", out, clk);
    run<24>("v_mad_u32_u24", out, clk);
    run<25>("v_mov_b32 dpp", out, clk);
    run<26>("v_add_u32 sdwa", out, clk);
    run<27>("v_pk_mul_lo_u16", out, clk);
    run<28>("v_fma_f64", out, clk);
    run<29>("v_sub_u32", out, clk);
    return 0;
}
