// Micro-benchmark + check: 128-bit multiply-accumulate acc += a * b (a, b < 2^62.1) in two forms.
//   plain    fp.cuh acc_mad as the compiler emits it: 4 v_mad_u64_u32, 5 moves that zero-extend a word into a 64-bit addend, one 32-bit add,
//            a four-word carry chain (44 issue cycles by the cost table of tools/isa_mix.py)
//   carries  the four products are added where they stand -- a0 b0 into a low 64-bit lane, a1 b0 and a0 b1 into a middle lane (weight
//            2^32), a1 b1 into a high lane (weight 2^64) -- with the addend of every v_mad_u64_u32 the lane itself (no moves), and the
//            carry-outs of the low and middle lanes, which v_mad_u64_u32 delivers in a scalar pair, counted into two words: 4 v_mad_u64_u32
//            + 3 v_addc_co_u32 (28 cycles), eight registers of state instead of four
// Every lane accumulates TERMS products into ACCS accumulators and reduces them to one 128-bit value; both forms must agree.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/accmad_bench tools/micro/accmad_bench.hip && /tmp/accmad_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

struct Acc128 { uint64_t lo, hi; };
__device__ __forceinline__ uint64_t mad_u64_u32(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }
__device__ __forceinline__ void acc_mad(Acc128 &acc, uint64_t a, uint64_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t t0 = (uint64_t)a0 * b0;
    const uint64_t t1 = mad_u64_u32(a1, b0, t0 >> 32);
    const uint64_t t2 = mad_u64_u32(a0, b1, (uint32_t)t1);
    const uint64_t t3 = mad_u64_u32(a1, b1, (uint32_t)((uint32_t)(t1 >> 32) + (uint32_t)(t2 >> 32)));
    unsigned cy;
    const uint32_t l0 = __builtin_addc((uint32_t)acc.lo, (uint32_t)t0, 0u, &cy);
    const uint32_t l1 = __builtin_addc((uint32_t)(acc.lo >> 32), (uint32_t)t2, cy, &cy);
    const uint32_t h0 = __builtin_addc((uint32_t)acc.hi, (uint32_t)t3, cy, &cy);
    const uint32_t h1 = __builtin_addc((uint32_t)(acc.hi >> 32), (uint32_t)(t3 >> 32), cy, &cy);
    acc.lo = ((uint64_t)l1 << 32) | l0;
    acc.hi = ((uint64_t)h1 << 32) | h0;
}

struct AccLanes { uint64_t lo, mid, hi; uint32_t c_lo, c_mid; }; // value = lo + 2^32 mid + 2^64 (hi + c_lo) + 2^96 c_mid
// b in vector registers
__device__ __forceinline__ void acc_mad_lanes(AccLanes &A, uint64_t a, uint64_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    uint64_t k0, k1, k2;
    asm("v_mad_u64_u32 %[lo], %[k0], %[a0], %[b0], %[lo]\n\t"
        "v_mad_u64_u32 %[mid], %[k1], %[a1], %[b0], %[mid]\n\t"
        "v_mad_u64_u32 %[hi], vcc, %[a1], %[b1], %[hi]\n\t"
        "v_mad_u64_u32 %[mid], %[k2], %[a0], %[b1], %[mid]\n\t"
        "v_addc_co_u32_e64 %[cl], vcc, %[cl], 0, %[k0]\n\t"
        "v_addc_co_u32_e64 %[cm], vcc, %[cm], 0, %[k1]\n\t"
        "v_addc_co_u32_e64 %[cm], vcc, %[cm], 0, %[k2]"
        : [lo] "+v"(A.lo), [mid] "+v"(A.mid), [hi] "+v"(A.hi), [cl] "+v"(A.c_lo), [cm] "+v"(A.c_mid), [k0] "=&s"(k0), [k1] "=&s"(k1), [k2] "=&s"(k2)
        : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1)
        : "vcc");
}
// b uniform: its words are scalar operands
__device__ __forceinline__ void acc_mad_lanes_s(AccLanes &A, uint64_t a, uint64_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    uint64_t k0, k1, k2;
    asm("v_mad_u64_u32 %[lo], %[k0], %[b0], %[a0], %[lo]\n\t"
        "v_mad_u64_u32 %[mid], %[k1], %[b0], %[a1], %[mid]\n\t"
        "v_mad_u64_u32 %[hi], vcc, %[b1], %[a1], %[hi]\n\t"
        "v_mad_u64_u32 %[mid], %[k2], %[b1], %[a0], %[mid]\n\t"
        "v_addc_co_u32_e64 %[cl], vcc, %[cl], 0, %[k0]\n\t"
        "v_addc_co_u32_e64 %[cm], vcc, %[cm], 0, %[k1]\n\t"
        "v_addc_co_u32_e64 %[cm], vcc, %[cm], 0, %[k2]"
        : [lo] "+v"(A.lo), [mid] "+v"(A.mid), [hi] "+v"(A.hi), [cl] "+v"(A.c_lo), [cm] "+v"(A.c_mid), [k0] "=&s"(k0), [k1] "=&s"(k1), [k2] "=&s"(k2)
        : [a0] "v"(a0), [a1] "v"(a1), [b0] "s"(b0), [b1] "s"(b1)
        : "vcc");
}
__device__ __forceinline__ Acc128 lanes_value(const AccLanes &A) {
    const unsigned __int128 v = (unsigned __int128)A.lo + ((unsigned __int128)A.mid << 32) + ((unsigned __int128)(A.hi + A.c_lo) << 64) + ((unsigned __int128)A.c_mid << 96);
    return {(uint64_t)v, (uint64_t)(v >> 64)};
}

constexpr int ACCS = 4, TERMS = 7;
constexpr uint64_t P = 0x4180000000000001ull;
__device__ __forceinline__ uint64_t next_val(uint64_t &st) { st = st * 6364136223846793005ull + 1442695040888963407ull; return (st >> 2) % P; }

template <int FORM>
__global__ __launch_bounds__(256) void k_bench(const uint64_t *__restrict__ coef, uint64_t *__restrict__ out, int iters, unsigned long long *clk) {
    uint64_t st = blockIdx.x * 256 + threadIdx.x + 1;
    uint64_t x[TERMS];
    for (int i = 0; i < TERMS; i++) x[i] = next_val(st);
    Acc128 total = {0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        Acc128 a[ACCS];
        AccLanes l[ACCS];
        for (int q = 0; q < ACCS; q++) { a[q] = {0, 0}; l[q] = {0, 0, 0, 0, 0}; }
#pragma unroll
        for (int i = 0; i < TERMS; i++)
#pragma unroll
            for (int q = 0; q < ACCS; q++) {
                const uint64_t c = FORM == 2 ? __builtin_nontemporal_load(&coef[(it & 3) * 32 + q * TERMS + i]) : coef[(it & 3) * 32 + q * TERMS + i] ^ 0;
                if (FORM == 0) acc_mad(a[q], x[i], c);
                if (FORM == 1) acc_mad_lanes(l[q], x[i], c);
                if (FORM == 2) acc_mad_lanes_s(l[q], x[i], c);
            }
        for (int q = 0; q < ACCS; q++) {
            const Acc128 v = FORM == 0 ? a[q] : lanes_value(l[q]);
            total.lo ^= v.lo; total.hi += v.hi;
        }
        x[it % TERMS] ^= total.lo & 0xffff; // keep the loop from being hoisted; stays below p
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[2 * (blockIdx.x * 256 + threadIdx.x)] = total.lo;
    out[2 * (blockIdx.x * 256 + threadIdx.x) + 1] = total.hi;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[FORM] = t1 - t0;
}

int main() {
    const int blocks = 256 * 4, iters = 2000;
    uint64_t *coef, *out[3], h_coef[128];
    unsigned long long *clk, h_clk[3];
    uint64_t st = 12345;
    for (int i = 0; i < 128; i++) { st = st * 6364136223846793005ull + 1442695040888963407ull; h_coef[i] = (st >> 2) % P; }
    h_coef[0] = P - 1; h_coef[1] = P - 1; h_coef[7] = 0xffffffffull; h_coef[8] = P - 1;
    (void)hipMalloc(&coef, sizeof h_coef); (void)hipMemcpy(coef, h_coef, sizeof h_coef, hipMemcpyHostToDevice);
    (void)hipMalloc(&clk, 24);
    for (int f = 0; f < 3; f++) (void)hipMalloc(&out[f], (size_t)blocks * 256 * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms[3];
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k_bench<0>, dim3(blocks), dim3(256), 0, 0, coef, out[0], iters, clk); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[0], e0, e1);
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k_bench<1>, dim3(blocks), dim3(256), 0, 0, coef, out[1], iters, clk); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[1], e0, e1);
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k_bench<2>, dim3(blocks), dim3(256), 0, 0, coef, out[2], iters, clk); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[2], e0, e1);
    }
    (void)hipMemcpy(h_clk, clk, 24, hipMemcpyDeviceToHost);
    const size_t words = (size_t)blocks * 256 * 2;
    uint64_t *h[3];
    for (int f = 0; f < 3; f++) { h[f] = (uint64_t *)malloc(words * 8); (void)hipMemcpy(h[f], out[f], words * 8, hipMemcpyDeviceToHost); }
    size_t bad1 = 0, bad2 = 0;
    for (size_t i = 0; i < words; i++) { bad1 += h[0][i] != h[1][i]; bad2 += h[0][i] != h[2][i]; }
    const double mads = (double)blocks * 4 /*waves*/ * iters * ACCS * TERMS;
    const char *name[3] = {"plain (compiler)", "lanes + carry-outs, vector b", "lanes + carry-outs, scalar b"};
    for (int f = 0; f < 3; f++)
        printf("%-30s %.3f ms  %.1f ns per wave-level multiply-accumulate per SIMD  (%llu memtime ticks in wave 0)\n", name[f], ms[f], ms[f] * 1e6 / (mads / 1024), h_clk[f]);
    printf("mismatches against the plain form: vector b %zu, scalar b %zu of %zu words\n", bad1, bad2, words);
    return bad1 || bad2;
}
