// Micro-benchmark: issue cost of the integer instructions the field arithmetic is made of (gfx950).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/imad.hip -o /tmp/imad && /tmp/imad > profiles/rNN_imad_ubench.txt
// (bench.py reads the v_mad_u64_u32 line of the 4-waves-per-SIMD section from the newest such file as the integer-issue peak)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 4096
template <int OP>
__global__ void k(uint32_t *out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    uint64_t acc[8];
    uint32_t x[8];
    for (int i = 0; i < 8; i++) { acc[i] = a + i; x[i] = a * (i + 1); }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x[i]), "v"(b) : "vcc");
            if (OP == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            if (OP == 2) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
            if (OP == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            if (OP == 4) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            if (OP == 5) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
            if (OP == 6) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(x[i]), "+v"(x[(i+1)&7]) : "v"(b), "v"(a) : "vcc");
            if (OP == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(x[(i + 1) & 7]));
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 8; i++) s += acc[i] + x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
template <int OP>
void run(const char *name, int blocks, int threads) {
    uint32_t *d;
    hipMalloc(&d, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(d, 12345);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(d, 12345);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double waves = (double)blocks * threads / 64;
    double wave_instr = waves * ITERS * 8;
    // cycles per wave-instruction per SIMD at 2.4 GHz with 1024 SIMDs
    double waves_per_simd = waves / 1024.0;
    double cyc = ms * 1e-3 * 2.4e9 / (ITERS * 8.0 * (waves_per_simd < 1 ? 1 : waves_per_simd));
    printf("%-14s blocks=%5d threads=%4d  %.3f ms  %.2f Ginstr/s(wave)  ~%.2f cycles/wave-instr/SIMD\n", name, blocks, threads, ms, wave_instr / ms / 1e6, cyc);
    hipFree(d);
}
int main() {
    for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = 1024 * wps, threads = 64;
        printf("--- %d wave(s) per SIMD\n", wps);
        run<0>("mad_u64_u32", blocks, threads);
        run<1>("add_u32", blocks, threads);
        run<2>("add_u64", blocks, threads);
        run<3>("mul_lo_u32", blocks, threads);
        run<4>("mul_hi_u32", blocks, threads);
        run<5>("mad_u32_u24", blocks, threads);
        run<6>("add_co+addc(2)", blocks, threads);
        run<7>("mov_b32", blocks, threads);
    }
    return 0;
}
