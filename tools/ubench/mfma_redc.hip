// Micro-benchmark (time-boxed): the constant-operand half of the 381-bit Montgomery multiplication -
// m = T_low * p' mod 2^392 and (T + m p) >> 392 - on the integer matrix cores against the 196 v_mad_u64_u32 the kernels
// use today (csrc/h2v_field.hpp: fp_mont28).
//
// The matrix-core form: 7-bit limbs (a 28-bit limb of the kernels' radix is exactly four of them, so slicing needs no
// carries), the two constant multiplications as Toeplitz matrices times a 56 x 32 operand - 32 field elements per wave,
// element j in lanes j and j + 32 - on v_mfma_i32_32x32x16_i8 (i32 accumulate: 56 products of 2^14 stay below 2^20).
// With the zero blocks of the band matrices skipped: 6 tiles for the low product (56 rows) + 6 for rows 56..119 of m p =
// 12 MFMAs per 32 reductions.  Around them, per lane: slicing 14 limbs into bytes (6 ops per limb), moving half of every
// operand to the partner lane (v_permlane32_swap), and carrying the 20-bit column sums back to 28-bit limbs twice (after
// the low product: m must be bytes again; after the second: the result), each as two data-parallel passes (a serial carry
// chain would hop between the lane halves seven times).
//
// What is measured (cycles per REDUCTION per SIMD, one and two waves per SIMD):
//   mad        the reduction as the kernels do it: 196 v_mad_u64_u32 + 14 v_mul_lo on one lane per element (64 per wave)
//   mfma_only  the 12 MFMAs per 32 elements back to back (the matrix pipe's own cost)
//   mfma_full  MFMAs + slicing + exchanges + the four normalisation passes.  The shell has the instruction count of the
//              real thing but is NOT a validated reduction (the two-pass carry leaves a limb of 2^28 in rare cases, which
//              a real kernel must still handle): its time is a LOWER bound for the matrix-core form.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_redc.hip -o /tmp/mfma_redc && /tmp/mfma_redc
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITERS 2048
typedef int v16i __attribute__((ext_vector_type(16)));
__constant__ uint32_t P28[14] = {0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u, 0xf38512bu,
                                 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u};
#define N0 0xffcfffdu
#define MASK 0x0fffffffu

// ---- the reduction as the kernels do it (product columns given): one element per lane
__global__ void __launch_bounds__(64) k_mad(uint32_t *out, uint32_t seed) {
    uint64_t col[27];
    for (int k = 0; k < 27; k++) col[k] = (uint64_t)(seed + threadIdx.x * 977u + k * 131u) * 0x9e3779b97f4a7c15ull >> 8;
    uint32_t t[14];
    for (int it = 0; it < ITERS; it++) {
        uint32_t m[14];
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 14; k++) {
            acc += col[k];
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P28[k - i];
            m[k] = ((uint32_t)acc * N0) & MASK;
            acc += (uint64_t)m[k] * P28[0];
            acc >>= 28;
        }
#pragma unroll
        for (int k = 14; k < 27; k++) {
            acc += col[k];
#pragma unroll
            for (int i = k - 13; i < 14; i++) acc += (uint64_t)m[i] * P28[k - i];
            t[k - 14] = (uint32_t)acc & MASK;
            acc >>= 28;
        }
        t[13] = (uint32_t)acc;
#pragma unroll
        for (int k = 0; k < 14; k++) { col[k] += t[k]; col[13 + k] ^= t[13 - k]; }   // next iteration depends on this one
    }
    uint32_t s = 0;
    for (int k = 0; k < 14; k++) s ^= t[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

// ---- matrix-core form
__device__ __forceinline__ uint32_t slice7(uint32_t x) {   // 28 bits -> four 7-bit bytes (6 ops)
    const uint32_t y = (x & 0x3fffu) | ((x & 0xfffc000u) << 2);
    return (y & 0x007f007fu) | ((y & 0x3f803f80u) << 1);
}
__device__ __forceinline__ void swap32(uint32_t &a, uint32_t &b) {   // upper half of a <-> lower half of b
    const auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = sw[0]; b = sw[1];
}
// 16 column sums of one M-tile (this lane's four groups of four consecutive rows) -> four 28-bit limbs + their carries
__device__ __forceinline__ void columns_to_limbs(const v16i &z, uint32_t (&lo)[4], uint32_t (&hi)[4]) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const uint64_t L = (uint64_t)(uint32_t)z[4 * g] + ((uint64_t)(uint32_t)z[4 * g + 1] << 7) + ((uint64_t)(uint32_t)z[4 * g + 2] << 14) +
                           ((uint64_t)(uint32_t)z[4 * g + 3] << 21);
        lo[g] = (uint32_t)L & MASK;
        hi[g] = (uint32_t)(L >> 28);
    }
}
template <bool FULL>
__global__ void __launch_bounds__(64) k_mfma(uint32_t *out, const uint64_t *__restrict__ toep, uint32_t seed) {
    const int lane = threadIdx.x;
    // the 12 constant tiles (row-permuted band matrices of p' and p), 8 bytes per lane each
    long A[12];
#pragma unroll
    for (int q = 0; q < 12; q++) A[q] = (long)toep[q * 64 + lane];
    uint32_t t28[14], thi[14];
    for (int k = 0; k < 14; k++) { t28[k] = (seed * (k + 3) + lane * 2654435761u) & MASK; thi[k] = (seed * (k + 11) + lane * 40503u) & MASK; }
    for (int it = 0; it < ITERS; it++) {
        uint32_t b[16];
        if (FULL) {
            // slicing + the partner's half of every operand
#pragma unroll
            for (int k = 0; k < 14; k++) b[k] = slice7(t28[k]);
            b[14] = 0; b[15] = 0;
#pragma unroll
            for (int s = 0; s < 4; s++) { swap32(b[4 * s + 2], b[4 * s]); swap32(b[4 * s + 3], b[4 * s + 1]); }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) b[k] = t28[k % 14];
        }
        // low product: 56 rows, 6 non-zero tiles
        v16i z0 = {0}, z1 = {0};
#define B64(s) ((long)(((uint64_t)b[4 * (s) + 1] << 32) | b[4 * (s)]))
        z0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[0], B64(0), z0, 0, 0, 0);
        z0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[1], B64(1), z0, 0, 0, 0);
        z1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[2], B64(0), z1, 0, 0, 0);
        z1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[3], B64(1), z1, 0, 0, 0);
        z1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[4], B64(2), z1, 0, 0, 0);
        z1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[5], B64(3), z1, 0, 0, 0);
        uint32_t mb[16];
        if (FULL) {
            // m as bytes again: columns -> limbs, two data-parallel carry passes (the neighbour limb lives in the partner half
            // for every second limb), then slicing
            uint32_t lo[8], hi[8];
            columns_to_limbs(z0, *(uint32_t(*)[4])&lo[0], *(uint32_t(*)[4])&hi[0]);
            columns_to_limbs(z1, *(uint32_t(*)[4])&lo[4], *(uint32_t(*)[4])&hi[4]);
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                uint32_t up[8];
#pragma unroll
                for (int g = 0; g < 8; g++) up[g] = hi[g];
#pragma unroll
                for (int g = 1; g < 8; g += 2) swap32(up[g], up[g - 1]);       // carries that cross the lane halves
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    const uint32_t v = lo[g] + up[(g + 7) & 7];
                    lo[g] = v & MASK;
                    hi[g] = v >> 28;
                }
            }
#pragma unroll
            for (int g = 0; g < 8; g++) { mb[2 * g] = slice7(lo[g]); mb[2 * g + 1] = slice7(lo[g] ^ hi[g]); }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) mb[k] = (uint32_t)z0[k] ^ (uint32_t)z1[k];
        }
#define M64(s) ((long)(((uint64_t)mb[4 * (s) + 1] << 32) | mb[4 * (s)]))
        // rows 56 .. 119 of m p: 6 non-zero tiles
        v16i y0 = {0}, y1 = {0};
        y0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[6], M64(0), y0, 0, 0, 0);
        y0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[7], M64(1), y0, 0, 0, 0);
        y0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[8], M64(2), y0, 0, 0, 0);
        y0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[9], M64(3), y0, 0, 0, 0);
        y1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[10], M64(2), y1, 0, 0, 0);
        y1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[11], M64(3), y1, 0, 0, 0);
        if (FULL) {
            uint32_t lo[8], hi[8];
            columns_to_limbs(y0, *(uint32_t(*)[4])&lo[0], *(uint32_t(*)[4])&hi[0]);
            columns_to_limbs(y1, *(uint32_t(*)[4])&lo[4], *(uint32_t(*)[4])&hi[4]);
#pragma unroll
            for (int g = 0; g < 7; g++) lo[g] += thi[2 * g];                   // + T_high (this half's limbs)
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                uint32_t up[8];
#pragma unroll
                for (int g = 0; g < 8; g++) up[g] = hi[g];
#pragma unroll
                for (int g = 1; g < 8; g += 2) swap32(up[g], up[g - 1]);
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    const uint32_t v = lo[g] + up[(g + 7) & 7];
                    lo[g] = v & MASK;
                    hi[g] = v >> 28;
                }
            }
            // back to one element per lane: the partner's seven limbs
            uint32_t other[8];
#pragma unroll
            for (int g = 0; g < 8; g++) other[g] = lo[g];
#pragma unroll
            for (int g = 0; g < 8; g += 2) swap32(other[g], other[g + 1]);
#pragma unroll
            for (int k = 0; k < 7; k++) { t28[2 * k] = lo[k]; t28[2 * k + 1] = other[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 14; k++) t28[k] = ((uint32_t)y0[k] + (uint32_t)y1[k]) & MASK;
        }
    }
    uint32_t s = 0;
    for (int k = 0; k < 14; k++) s ^= t28[k];
    out[blockIdx.x * 64 + lane] = s;
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    uint32_t *d;
    uint64_t *toep;
    hipMalloc(&d, (size_t)4096 * 64 * 4);
    hipMalloc(&toep, 12 * 64 * 8);
    // band-matrix tiles of p' and p as bytes (the values only matter for a correctness test; the timing needs non-trivial operands)
    uint64_t h[12 * 64];
    for (int q = 0; q < 12 * 64; q++) {
        uint64_t v = 0;
        for (int k = 0; k < 8; k++) v |= (uint64_t)(((q * 37 + k * 11) * 2654435761u >> 9) & 0x7f) << (8 * k);
        h[q] = v;
    }
    hipMemcpy(toep, h, sizeof h, hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 2; wps++) {
        const int blocks = 1024 * wps;
        printf("--- %d wave(s) per SIMD (1024 SIMDs, 2.4 GHz)\n", wps);
        const double t_mad = time_ms([&] { k_mad<<<blocks, 64>>>(d, 7); });
        const double t_only = time_ms([&] { k_mfma<false><<<blocks, 64>>>(d, toep, 7); });
        const double t_full = time_ms([&] { k_mfma<true><<<blocks, 64>>>(d, toep, 7); });
        // cycles per reduction per SIMD: a wave does 64 (mad) / 32 (mfma) reductions per iteration
        auto cyc = [&](double ms, int per_wave) { return ms * 1e-3 * 2.4e9 / ((double)ITERS * wps * per_wave); };
        printf("mad        %.3f ms  %.2f cycles per reduction per SIMD  (196 v_mad_u64_u32 + 14 v_mul_lo per lane, 64 reductions per wave)\n", t_mad, cyc(t_mad, 64));
        printf("mfma_only  %.3f ms  %.2f cycles per reduction per SIMD  (12 v_mfma_i32_32x32x16_i8 per 32 reductions)\n", t_only, cyc(t_only, 32));
        printf("mfma_full  %.3f ms  %.2f cycles per reduction per SIMD  (+ slicing, lane exchanges, four carry passes: LOWER bound, see header)\n", t_full, cyc(t_full, 32));
    }
    return 0;
}
