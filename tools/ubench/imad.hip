// Micro-benchmark: issue cost of the integer (and FP64) instructions the field arithmetic is made of, gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/imad.hip -o /tmp/imad && /tmp/imad > profiles/rNN_imad_ubench.txt
// bench.py reads the `mad_u64_u32` line of the 4-waves-per-SIMD section of the newest such file as the integer-issue peak.
//
// Round 4: every figure is the MEDIAN of 5 timed launches of >= 5 ms each (the iteration count is
// calibrated per instruction), after a warm-up launch; sections for 1 / 2 / 4 / 8 waves per SIMD (256 workgroups of 4 W
// waves; W = 8: 512 of 16).  Three readings per row:
//   * event time x 2.4 GHz / (instructions per wave x W): what a launch of that shape delivers (it assumes that the
//     dispatcher gave every SIMD exactly W waves - it does not: see the histogram);
//   * PER SIMD, the honest one: every wave stamps s_memtime at both ends of its loop and reads where it ran (HW_ID / XCC_ID);
//     the host groups the waves by physical SIMD and takes  (last end - first start) / instructions issued there, median over
//     the SIMDs that really held W waves - with the histogram of waves per SIMD beside it;
//   * the clock the chip held (s_memtime against the 100 MHz s_memrealtime).
// The loop body is 64 instructions (8 independent chains x 8), so the loop's own scalar instructions and branch are < 5 %.
// Mixed rows (a multiply-add stream with a second instruction stream interleaved) show whether costs ADD or OVERLAP.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

enum { MAD64 = 0, ADD32, ADD64, MULLO, MULHI, MAD24, ADDC2, MOV, MADI64, FMA64, ADDF64, MAD_ADD32, MAD_ADD64, MAD_MOV, FMA64_ADD64, MAD64_VCC, MAD64_CHAIN, MAD64_SGPR, MAD_ADD32_FREE, N_OPS };
static const char *NAMES[N_OPS] = {"mad_u64_u32", "add_u32", "add_u64", "mul_lo_u32", "mul_hi_u32", "mad_u32_u24", "add_co+addc(2)", "mov_b32",
                                   "mad_i64_i32", "fma_f64", "add_f64", "mad64+add32(2)", "mad64+add64(2)", "mad64+mov(2)", "fma64+add64(2)",
                                   "mad64 cout=vcc", "mad64 one-acc", "mad64 x sgpr", "mad64+add32 free"};
static const int PER_ITER[N_OPS] = {64, 64, 64, 64, 64, 64, 128, 64, 64, 64, 64, 128, 128, 128, 128, 64, 64, 64, 128};   // instructions per loop iteration

template <int OP>
__global__ void __launch_bounds__(1024) k(uint32_t *out, uint64_t *stamps, uint32_t seed, uint32_t iters) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    uint64_t acc[8];
    uint32_t x[8], y[8];
    double f[8], g = 1.0000001, h = 0.5;
    for (int i = 0; i < 8; i++) { acc[i] = a + i; x[i] = a * (i + 1); y[i] = a ^ i; f[i] = 1.0 + i + a; }
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int i8 = 0; i8 < 64; i8++) {
            const int i = i8 & 7;
            // (carry-out to an SGPR pair, as compiled code has it; the rows `cout=vcc`, `one-acc`, `x sgpr` vary that form)
            if (OP == MAD64 || OP == MAD_ADD32 || OP == MAD_ADD64 || OP == MAD_MOV || OP == MAD_ADD32_FREE) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[i]) : "v"(x[i]), "v"(b) : "s20", "s21");
            if (OP == MAD64_VCC) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x[i]), "v"(b) : "vcc");
            if (OP == MAD64_CHAIN) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[0]) : "v"(x[i]), "v"(b) : "s20", "s21");
            if (OP == MAD64_SGPR) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[i]) : "v"(x[i]), "s"(seed) : "s20", "s21");
            if (OP == MAD_ADD32_FREE) asm volatile("v_add_u32 %0, %0, %1" : "+v"(y[i]) : "v"(b));
            if (OP == ADD32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            if (OP == MAD_ADD32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            if (OP == ADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
            if (OP == MAD_ADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));   // (f: a second register set)
            if (OP == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            if (OP == MULHI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            if (OP == MAD24) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
            if (OP == ADDC2) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(x[i]), "+v"(x[(i + 1) & 7]) : "v"(b), "v"(a) : "vcc");
            if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(x[(i + 1) & 7]));
            if (OP == MAD_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(x[(i + 1) & 7]));
            if (OP == MADI64) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x[i]), "v"(b) : "vcc");
            if (OP == FMA64 || OP == FMA64_ADD64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[i]) : "v"(g), "v"(h));
            if (OP == ADDF64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[i]) : "v"(g));
            if (OP == FMA64_ADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint64_t s = 0;
    for (int i = 0; i < 8; i++) s += acc[i] + x[i] + y[i] + (uint64_t)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
    if ((threadIdx.x & 63) == 0) {
        const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64;
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_HW_ID, HW_REG_XCC_ID
        stamps[5 * w] = t1 - t0;
        stamps[5 * w + 1] = r1 - r0;
        stamps[5 * w + 2] = ((uint64_t)(xcc & 0xf) << 16) | (hw & 0xff30);   // se_id, sh_id, cu_id, simd_id (wave slot and pipe left out)
        stamps[5 * w + 3] = t0;
        stamps[5 * w + 4] = t1;
    }
}

template <int OP>
static void run(int wps) {
    const int threads = wps >= 4 ? 1024 : 256 * wps, blocks = 256 * (wps >= 4 ? wps / 4 : 1);
    const size_t waves = (size_t)blocks * threads / 64;
    uint32_t *d;
    uint64_t *st;
    hipMalloc(&d, (size_t)blocks * threads * 4);
    hipMalloc(&st, waves * 40);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto once = [&](uint32_t iters) {
        hipEventRecord(e0);
        k<OP><<<blocks, threads>>>(d, st, 12345, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        return ms;
    };
    uint32_t iters = 512;
    float ms = once(iters);                                    // warm-up + calibration: at least 5 ms per timed launch
    while (ms < 5.0f && iters < (1u << 26)) { iters = (uint32_t)(iters * (ms > 0.05f ? 6.0f / ms : 16.0f)) + 1; ms = once(iters); }
    std::vector<float> t;
    std::vector<double> clk, simd_cyc;
    int hist[17] = {};
    const double n_instr = (double)iters * PER_ITER[OP];
    for (int rep = 0; rep < 5; rep++) {
        t.push_back(once(iters));
        std::vector<uint64_t> h(5 * waves);
        hipMemcpy(h.data(), st, waves * 40, hipMemcpyDeviceToHost);
        std::vector<double> r(waves);
        for (size_t w = 0; w < waves; w++) r[w] = (double)h[5 * w] / ((double)h[5 * w + 1] * 10.0);   // s_memrealtime: 100 MHz
        std::sort(r.begin(), r.end());
        clk.push_back(r[waves / 2]);
        // group by physical SIMD
        std::vector<std::pair<uint64_t, size_t>> key(waves);
        for (size_t w = 0; w < waves; w++) key[w] = {h[5 * w + 2], w};
        std::sort(key.begin(), key.end());
        std::vector<double> per;
        for (size_t a = 0; a < waves;) {
            size_t b = a;
            uint64_t lo = ~0ull, hi = 0;
            while (b < waves && key[b].first == key[a].first) { lo = std::min(lo, h[5 * key[b].second + 3]); hi = std::max(hi, h[5 * key[b].second + 4]); b++; }
            const int cnt = (int)(b - a);
            if (rep == 0) hist[cnt > 16 ? 16 : cnt]++;
            if (cnt == wps) per.push_back((double)(hi - lo) / (n_instr * cnt));
            a = b;
        }
        std::sort(per.begin(), per.end());
        simd_cyc.push_back(per.empty() ? 0.0 : per[per.size() / 2]);
    }
    std::sort(t.begin(), t.end());
    std::sort(clk.begin(), clk.end());
    std::sort(simd_cyc.begin(), simd_cyc.end());
    const double ev_cyc = t[2] * 1e-3 * 2.4e9 / (n_instr * wps);          // event time, 2.4 GHz and W waves on every SIMD assumed
    printf("%-15s waves/SIMD=%d iters=%7u  median %.3f ms (min %.3f max %.3f)  ~%.2f cycles/wave-instr/SIMD at 2.4 GHz | per SIMD that held %d waves: %.2f cycles/wave-instr, clock %.2f GHz | SIMDs by waves held:",
           NAMES[OP], wps, iters, t[2], t[0], t[4], ev_cyc, wps, simd_cyc[2], clk[2]);
    for (int c = 1; c <= 16; c++) if (hist[c]) printf(" %dx%d", hist[c], c);
    printf("\n");
    hipFree(d);
    hipFree(st);
}
int main() {
    printf("# imad ubench v2: median of 5 launches of >= 5 ms; 256 workgroups x (4 x waves/SIMD) waves; cycles = time x clock / (instructions per wave x waves per SIMD)\n");
    for (int wps = 1; wps <= 8; wps *= 2) {
        printf("--- %d wave(s) per SIMD\n", wps);
        run<MAD64>(wps); run<ADD32>(wps); run<ADD64>(wps); run<MULLO>(wps); run<MULHI>(wps); run<MAD24>(wps); run<ADDC2>(wps); run<MOV>(wps);
        run<MADI64>(wps); run<FMA64>(wps); run<ADDF64>(wps); run<MAD_ADD32>(wps); run<MAD_ADD64>(wps); run<MAD_MOV>(wps); run<FMA64_ADD64>(wps);
        run<MAD64_VCC>(wps); run<MAD64_CHAIN>(wps); run<MAD64_SGPR>(wps); run<MAD_ADD32_FREE>(wps);
    }
    return 0;
}
