// Micro-benchmark (time-boxed): a 381-bit Montgomery product on the vector FP64 pipe against the one on
// v_mad_u64_u32 that the kernels use.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dfma_mont.hip -o /tmp/dfma && /tmp/dfma > profiles/rNN_dfma_ubench.txt
//
// FP64 form (Emmart, Zheng, Weems: "Faster modular exponentiation using double precision floating point arithmetic on the
// GPU", ARITH 2018): 8 limbs of 52 bits held as doubles, R = 2^416.  A limb product a b < 2^104 comes as an exact pair
//     hi = fma_rz(a, b, 2^104)                 = 2^104 + 2^52 floor(a b / 2^52)          (ulp at 2^104 is 2^52; round toward zero)
//     lo = fma_rz(a, b, (2^104 + 2^52) - hi)   = 2^52 + (a b mod 2^52)
// whose BIT PATTERNS are (exponent << 52) | payload, so a column is summed with 64-bit INTEGER additions of the raw patterns;
// the exponent constants are known per column and sit in the accumulators' initial values.  Per limb product: 2 v_fma_f64 +
// 1 v_add_f64 + 2 64-bit integer additions; 64 products for a b, 64 for the reduction, 8 quotient digits the same way.
// Integer form: 14 limbs of 28 bits, R = 2^392, product scanning: one v_mad_u64_u32 per limb product (196 + 196) - the
// multiplier of csrc/h2v_fp28.hpp, written out here so that the file stands alone.
// Both are checked BIT-EXACTLY against a host big-integer a b R^-1 mod p on 10 240 operand pairs (random + edge values).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef unsigned __int128 u128;
static const uint64_t P64[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const uint64_t RINV52_64[6] = {0x563bfca80d3357daull, 0x49ad31af38b4d1ffull, 0x1570c7a87408ddull, 0xbfcacba62ee6ff12ull, 0x60405e9a0c4ba44full, 0xcf3587083336882ull};   // 2^-416 mod p
static const uint64_t RINV28_64[6] = {0xd03433825937d5f3ull, 0x5bab6111a3ad18faull, 0x89b2c24e13432f44ull, 0x29e226de2c8bd445ull, 0x58dea736114b9b5aull, 0x1055a9f965d8eb2dull};  // 2^-392 mod p

__constant__ double P52D[8] = {(double)0xeffffffffaaabull, (double)0xfeb153ffffb9full, (double)0x6b0f6241eabffull, (double)0x12bf6730d2a0full,
                               (double)0x764774b84f385ull, (double)0x1ba7b6434bacdull, (double)0x1ea397fe69a4bull, (double)0x1a011ull};
#define N0_52D ((double)0x3fffcfffcfffdull)
__constant__ uint32_t P28[14] = {0xfffaaab, 0xfefffff, 0x3ffffb9, 0xfffeb15, 0x6241eab, 0xa0f6b0f, 0xf6730d2, 0xf38512b, 0x4774b84, 0x4bacd76, 0xba7b643, 0xe69a4b1, 0x1ea397f, 0x1a011};
#define N0_28 0xffcfffdu
#define M52 ((1ull << 52) - 1)
#define C_LO (1075ull << 52)   // bit pattern of 2^52
#define C_HI (1127ull << 52)   // bit pattern of 2^104

__device__ __forceinline__ double fma_rz(double a, double b, double c) {   // the wave runs with MODE.fp_round(f64) = toward zero
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));   // (not volatile: free to schedule; every operand descends from the token below)
    return r;
}
__device__ __forceinline__ double sub_exact(double a, double b) {
    double r;
    asm("v_add_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void prod52(double a, double b, uint64_t &lo_bits, uint64_t &hi_bits) {
    const double hi = fma_rz(a, b, 0x1p104);
    const double sub = sub_exact(0x1p104 + 0x1p52, hi);
    const double lo = fma_rz(a, b, sub);
    lo_bits = (uint64_t)__double_as_longlong(lo);
    hi_bits = (uint64_t)__double_as_longlong(hi);
}
__device__ __forceinline__ double u52_to_double(uint64_t v) { return sub_exact(__longlong_as_double((long long)(v | C_LO)), 0x1p52); }
__host__ __device__ constexpr uint64_t n_diag(int c) { return c < 0 || c > 14 ? 0 : (uint64_t)((c < 14 - c ? c : 14 - c) + 1); }

// t = a b 2^-416 mod p (+ a multiple of p below 2p): limbs in, limbs out (52 bits each, as integers)
__device__ __forceinline__ void montmul_dfma(uint64_t (&t)[8], const uint64_t (&a)[8], const uint64_t (&b)[8]) {
    double ad[8], bd[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { ad[i] = u52_to_double(a[i]); bd[i] = u52_to_double(b[i]); }
    uint64_t col[17];
#pragma unroll
    for (int c = 0; c < 17; c++) col[c] = 0ull - (2 * n_diag(c) * C_LO + 2 * n_diag(c - 1) * C_HI);   // every pattern that will land here
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t lo, hi;
            prod52(ad[i], bd[j], lo, hi);
            col[i + j] += lo;
            col[i + j + 1] += hi;
        }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint64_t low = (col[k] + C_LO) & M52;                      // (row k's own lo pattern has not arrived yet)
        uint64_t mlo, mhi;
        prod52(u52_to_double(low), N0_52D, mlo, mhi);
        const double m = u52_to_double(mlo & M52);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t lo, hi;
            prod52(m, P52D[j], lo, hi);
            col[k + j] += lo;
            col[k + j + 1] += hi;
        }
        col[k + 1] += col[k] >> 52;                                      // low 52 bits are zero now
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        t[k] = col[8 + k] & M52;
        col[9 + k] += col[8 + k] >> 52;
    }
}
// 14 x 28-bit limbs, R = 2^392 (csrc/h2v_fp28.hpp: fp28 product scanning, reduced below 2p, carried)
__device__ __forceinline__ void montmul_imad(uint32_t (&t)[14], const uint32_t (&a)[14], const uint32_t (&b)[14]) {
    uint64_t acc[28];
#pragma unroll
    for (int i = 0; i < 28; i++) acc[i] = 0;
#pragma unroll
    for (int i = 0; i < 14; i++)
#pragma unroll
        for (int j = 0; j < 14; j++) acc[i + j] += (uint64_t)a[i] * b[j];
#pragma unroll
    for (int k = 0; k < 14; k++) {
        const uint32_t m = ((uint32_t)acc[k] * N0_28) & 0xfffffffu;
#pragma unroll
        for (int j = 0; j < 14; j++) acc[k + j] += (uint64_t)m * P28[j];
        acc[k + 1] += acc[k] >> 28;
    }
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 13; k++) {
        carry += acc[14 + k];
        t[k] = (uint32_t)carry & 0xfffffffu;
        carry >>= 28;
    }
    t[13] = (uint32_t)(carry + acc[27]);
}

// mode 0: one product per lane (parity); mode 1: a dependent chain x <- x * y of `iters` products (throughput)
template <int MAXT> __global__ void __launch_bounds__(MAXT) k_dfma(const uint64_t *in, uint64_t *out, uint64_t *stamps, uint32_t n, uint32_t iters) {
    uint32_t tok;   // f64 / f16 rounding: toward zero; the token (0) enters the load index, so every FP64 operation below depends on this statement
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3\n\ts_mov_b32 %0, 0" : "=s"(tok));
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, idx = (gid < n ? gid : gid % n) + tok;
    uint64_t x[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = in[(size_t)idx * 16 + i]; y[i] = in[(size_t)idx * 16 + 8 + i]; }
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < iters; it++) {
        uint64_t t[8];
        montmul_dfma(t, x, y);
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = t[i];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (gid < n || iters > 1)
#pragma unroll
        for (int i = 0; i < 8; i++) out[(size_t)gid * 8 + i] = x[i];
    if (stamps && (threadIdx.x & 63) == 0) stamps[gid / 64] = t1 - t0;
}
template <int MAXT> __global__ void __launch_bounds__(MAXT) k_imad(const uint32_t *in, uint32_t *out, uint64_t *stamps, uint32_t n, uint32_t iters) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, idx = gid < n ? gid : gid % n;
    uint32_t x[14], y[14];
#pragma unroll
    for (int i = 0; i < 14; i++) { x[i] = in[(size_t)idx * 28 + i]; y[i] = in[(size_t)idx * 28 + 14 + i]; }
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < iters; it++) {
        uint32_t t[14];
        montmul_imad(t, x, y);
#pragma unroll
        for (int i = 0; i < 14; i++) x[i] = t[i];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (gid < n || iters > 1)
#pragma unroll
        for (int i = 0; i < 14; i++) out[(size_t)gid * 14 + i] = x[i];
    if (stamps && (threadIdx.x & 63) == 0) stamps[gid / 64] = t1 - t0;
}

// ---- host big integers (6 x 64 bits), slow and plain: the checker
struct B384 { uint64_t w[6]; };
static int cmp(const B384 &a, const B384 &b) { for (int i = 5; i >= 0; i--) if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1; return 0; }
static B384 P() { B384 p; memcpy(p.w, P64, 48); return p; }
static void sub_in(B384 &a, const B384 &b) { u128 br = 0; for (int i = 0; i < 6; i++) { u128 d = (u128)a.w[i] - b.w[i] - br; a.w[i] = (uint64_t)d; br = (d >> 64) & 1; } }
static void add_mod(B384 &a, const B384 &b) {   // a, b < p
    u128 c = 0;
    for (int i = 0; i < 6; i++) { c += (u128)a.w[i] + b.w[i]; a.w[i] = (uint64_t)c; c >>= 64; }
    const B384 p = P();
    if (c || cmp(a, p) >= 0) sub_in(a, p);
}
static B384 mul_mod(const B384 &a, const B384 &b) {   // double-and-add over the bits of b
    B384 r = {};
    for (int bit = 383; bit >= 0; bit--) {
        add_mod(r, r);
        if ((b.w[bit / 64] >> (bit % 64)) & 1) add_mod(r, a);
    }
    return r;
}
static B384 reduce_mod(B384 a) { const B384 p = P(); while (cmp(a, p) >= 0) sub_in(a, p); return a; }
static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static B384 rand_below_p() { B384 a; for (int i = 0; i < 6; i++) a.w[i] = rnd(); a.w[5] &= (1ull << 61) - 1; return reduce_mod(reduce_mod(a)); }
static void to_limbs(const B384 &a, int bits, int n, uint64_t *out) {
    for (int i = 0; i < n; i++) {
        const int lo = bits * i;
        u128 v = lo / 64 < 6 ? (u128)a.w[lo / 64] : 0;
        if (lo / 64 + 1 < 6) v |= (u128)a.w[lo / 64 + 1] << 64;
        out[i] = (uint64_t)(v >> (lo % 64)) & ((1ull << bits) - 1);
    }
}
static B384 from_limbs(const uint64_t *l, int bits, int n) {   // value may exceed 384 bits only by what reduce handles (< 2p here)
    B384 r = {};
    for (int i = 0; i < n; i++) {
        const int lo = bits * i;
        if (lo / 64 >= 6) continue;
        u128 v = (u128)l[i] << (lo % 64);
        u128 c = (u128)r.w[lo / 64] + (uint64_t)v;
        r.w[lo / 64] = (uint64_t)c;
        u128 carry = (c >> 64) + (v >> 64);
        for (int q = lo / 64 + 1; q < 6 && carry; q++) { c = (u128)r.w[q] + (uint64_t)carry; r.w[q] = (uint64_t)c; carry = (c >> 64) + (carry >> 64); }
    }
    return r;
}

template <class F> static void timed(const char *name, int wps, uint32_t n_in, double per_product_instr, F launch, uint64_t *d_st) {
    const int threads = wps >= 4 ? 1024 : 256 * wps, blocks = 256 * (wps >= 4 ? wps / 4 : 1);
    const size_t waves = (size_t)blocks * threads / 64;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto once = [&](uint32_t iters) { (void)hipEventRecord(e0); launch(blocks, threads, iters); (void)hipEventRecord(e1); (void)hipDeviceSynchronize(); float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms; };
    uint32_t iters = 256;
    float ms = once(iters);
    while (ms < 5.0f && iters < (1u << 24)) { iters = (uint32_t)(iters * (ms > 0.05f ? 6.0f / ms : 16.0f)) + 1; ms = once(iters); }
    std::vector<float> t;
    std::vector<uint64_t> cyc;
    for (int rep = 0; rep < 5; rep++) {
        t.push_back(once(iters));
        std::vector<uint64_t> h(waves);
        (void)hipMemcpy(h.data(), d_st, waves * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        cyc.push_back(h[waves / 2]);
    }
    std::sort(t.begin(), t.end()); std::sort(cyc.begin(), cyc.end());
    printf("%-10s waves/SIMD=%d  %u products per lane  median %.3f ms  | %.0f cycles per reduced product per SIMD at 2.4 GHz (event time), %.0f by s_memtime  (%.0f products/us/chip)\n",
           name, wps, iters, t[2], t[2] * 1e-3 * 2.4e9 / ((double)iters * wps), (double)cyc[2] / ((double)iters * wps), (double)iters * waves * 64 / (t[2] * 1e3));
    (void)n_in; (void)per_product_instr;
}

int main() {
    const uint32_t N = 10240;
    std::vector<B384> A(N), Bv(N);
    const B384 p = P();
    B384 pm1 = p; { B384 one = {{1, 0, 0, 0, 0, 0}}; sub_in(pm1, one); }
    std::vector<B384> edge;
    edge.push_back(B384{}); edge.push_back(B384{{1, 0, 0, 0, 0, 0}}); edge.push_back(pm1);
    for (int k = 0; k < 381; k += 13) { B384 e = {}; e.w[k / 64] = 1ull << (k % 64); edge.push_back(reduce_mod(e)); B384 f = e; sub_in(f, B384{{1, 0, 0, 0, 0, 0}}); if (k) edge.push_back(reduce_mod(f)); }
    { B384 e; for (int i = 0; i < 6; i++) e.w[i] = ~0ull; e.w[5] = (1ull << 60) - 1; edge.push_back(reduce_mod(e)); }   // all-ones limbs
    for (uint32_t i = 0; i < N; i++) {
        if (i < edge.size() * edge.size() && i < 4096) { A[i] = edge[i / edge.size()]; Bv[i] = edge[i % edge.size()]; }
        else { A[i] = rand_below_p(); Bv[i] = rand_below_p(); }
    }
    B384 rinv52, rinv28;
    memcpy(rinv52.w, RINV52_64, 48); memcpy(rinv28.w, RINV28_64, 48);
    std::vector<uint64_t> in52((size_t)N * 16);
    std::vector<uint32_t> in28((size_t)N * 28);
    for (uint32_t i = 0; i < N; i++) {
        to_limbs(A[i], 52, 8, &in52[(size_t)i * 16]); to_limbs(Bv[i], 52, 8, &in52[(size_t)i * 16 + 8]);
        uint64_t t[14];
        to_limbs(A[i], 28, 14, t); for (int q = 0; q < 14; q++) in28[(size_t)i * 28 + q] = (uint32_t)t[q];
        to_limbs(Bv[i], 28, 14, t); for (int q = 0; q < 14; q++) in28[(size_t)i * 28 + 14 + q] = (uint32_t)t[q];
    }
    const size_t max_threads = 256 * 2 * 1024;
    uint64_t *d_in52, *d_out52, *d_st; uint32_t *d_in28, *d_out28;
    (void)hipMalloc(&d_in52, in52.size() * 8); (void)hipMalloc(&d_out52, max_threads * 8 * 8); (void)hipMalloc(&d_st, max_threads / 64 * 8);
    (void)hipMalloc(&d_in28, in28.size() * 4); (void)hipMalloc(&d_out28, max_threads * 14 * 4);
    (void)hipMemcpy(d_in52, in52.data(), in52.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_in28, in28.data(), in28.size() * 4, hipMemcpyHostToDevice);
    // ---- parity: one product per lane, both forms, against the host's a b R^-1 mod p
    k_dfma<256><<<(N + 255) / 256, 256>>>(d_in52, d_out52, nullptr, N, 1);
    k_imad<256><<<(N + 255) / 256, 256>>>(d_in28, d_out28, nullptr, N, 1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<uint64_t> o52((size_t)N * 8); std::vector<uint32_t> o28((size_t)N * 14);
    (void)hipMemcpy(o52.data(), d_out52, o52.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(o28.data(), d_out28, o28.size() * 4, hipMemcpyDeviceToHost);
    uint32_t bad52 = 0, bad28 = 0, limb_overflow = 0;
    for (uint32_t i = 0; i < N; i++) {
        const B384 ab = mul_mod(A[i], Bv[i]);
        const B384 want52 = mul_mod(ab, rinv52), want28 = mul_mod(ab, rinv28);
        for (int q = 0; q < 8; q++) if (o52[(size_t)i * 8 + q] >> 52) limb_overflow++;
        const B384 got52 = reduce_mod(from_limbs(&o52[(size_t)i * 8], 52, 8));
        uint64_t t[14]; for (int q = 0; q < 14; q++) t[q] = o28[(size_t)i * 14 + q];
        const B384 got28 = reduce_mod(from_limbs(t, 28, 14));
        if (cmp(got52, want52) != 0) bad52++;
        if (cmp(got28, want28) != 0) bad28++;
    }
    printf("# parity on %u operand pairs (%zu x %zu edge pairs, the rest random): FP64 form %u mismatches (limbs above 52 bits: %u), integer form %u mismatches\n",
           N, edge.size(), edge.size(), bad52, limb_overflow, bad28);
    if (bad52 || bad28 || limb_overflow) { printf("PARITY FAILED\n"); return 2; }
    // ---- throughput: dependent chains, 1 / 2 / 4 waves per SIMD (256 workgroups of 4 W waves)
    for (int wps = 1; wps <= 4; wps *= 2) {
        printf("--- %d wave(s) per SIMD\n", wps);
        // (register budget of the launch bound: 512 / 256 / 128 per lane for 1 / 2 / 4 waves per SIMD)
        timed("fp64-fma", wps, N, 0, [&](int b, int t, uint32_t it) {
            if (t == 256) k_dfma<256><<<b, t>>>(d_in52, d_out52, d_st, N, it); else if (t == 512) k_dfma<512><<<b, t>>>(d_in52, d_out52, d_st, N, it); else k_dfma<1024><<<b, t>>>(d_in52, d_out52, d_st, N, it); }, d_st);
        timed("int-mad", wps, N, 0, [&](int b, int t, uint32_t it) {
            if (t == 256) k_imad<256><<<b, t>>>(d_in28, d_out28, d_st, N, it); else if (t == 512) k_imad<512><<<b, t>>>(d_in28, d_out28, d_st, N, it); else k_imad<1024><<<b, t>>>(d_in28, d_out28, d_st, N, it); }, d_st);
    }
    return 0;
}
