// Random-linear-combination (RLC) batch mode: the kernels around the bucket MSM of h2v_pippenger.hpp.
// Included at the end of h2v_kernels.hip (uses its transcript hash and field helpers).
//
// For a batch of B proofs of one plan, after the unchanged per-proof phase 1 (transcript + combiner -> the T MSM scalars
// s_{i,t}; decompression + subgroup test of every G1 element):
//   k_rlc_prepare   per proof: good_i = "nothing rejected so far"; r_i = low 128 bits of blake2b-256(seed || i) (0 when
//                   !good_i: the proof is rejected by itself and takes no part in the combination); the scaled scalars
//                   r_i s_{i,t} mod r of its per-proof terms, its share of the VK-base sums, and (r_i, pi_i) for the left side
//   k_rlc_vk_sum    sum_i r_i s_{i,f} per VK base f (the bases are the same for every proof: one term each)
//   bucket MSMs     R = sum_i r_i er_i (255-bit scalars, GLV)  and  L = sum_i r_i pi_i (128-bit scalars)
//   k_pairing_rlc   ONE check e(L, s_g2) == e(R, G2) (wide engine) with the epilogue fused in:
//                   accepted: accept[i] = good_i.  Rejected (some good-looking proof fails its own pairing equation, caught
//                   with probability >= 1 - 2^-128 over the seed): the per-proof MSM + pairing kernels that follow on the
//                   stream are NOT skipped and produce accept[] exactly as the per-proof mode does.
#pragma once
#include "h2v_pippenger.hpp"

struct RlcArgs {
    uint32_t n, n_var, n_fix, slots, pi_point, scal_stride;
    const uint32_t *terms;      // the plan's (kind, index) table: [0, n_var) per-proof slots, [n_var, n_var + n_fix) VK bases
    const uint32_t *scalars;    // n x scal_stride x 8, canonical
    uint32_t *status;           // n (H2V_ST_BAD_POINT is folded in here)
    const uint8_t *valid, *valid_sub;
    uint32_t seed[8];
    uint32_t *r_scal, *r_idx;   // n * n_var + n_fix terms of the right-hand MSM
    uint32_t *l_scal, *l_idx;   // n terms of the left-hand MSM
    uint32_t *vk_part;          // ceil(n / 64) x n_fix x 8 (Montgomery): per-block sums of r_i s_{i,f}
    uint8_t *good;              // n
};

extern "C" __global__ void __launch_bounds__(64)
k_rlc_prepare(RlcArgs a) {
    __shared__ uint32_t sbuf[32 * 64];
    const int lane = threadIdx.x;
    const uint32_t i = blockIdx.x * 64 + lane;
    const bool live = i < a.n;
    const uint32_t ii = live ? i : a.n - 1;
    uint32_t st = a.status[ii];
    for (uint32_t j = 0; j < a.slots; j++)
        if (!a.valid[(size_t)ii * a.slots + j] || (a.valid_sub && !a.valid_sub[(size_t)ii * a.slots + j])) st |= H2V_ST_BAD_POINT;
    const bool good = live && st == 0;
    // r_i: blake2b-256(seed || LE32(i)), low 128 bits (never 0 for a proof that takes part)
    Transcript tr;
    tr_init(tr);
#pragma unroll 1
    for (int k = 0; k < 32; k++) tr_put(tr, sbuf, lane, (a.seed[k >> 2] >> (8 * (k & 3))) & 0xffu);
#pragma unroll 1
    for (int k = 0; k < 4; k++) tr_put(tr, sbuf, lane, (ii >> (8 * k)) & 0xffu);
    uint64_t h[4];
    tr_digest(tr, sbuf, lane, h);
    Fr r, rm;
#pragma unroll
    for (int l = 0; l < 8; l++) r.v[l] = 0;
    r.v[0] = (uint32_t)h[0]; r.v[1] = (uint32_t)(h[0] >> 32); r.v[2] = (uint32_t)h[1]; r.v[3] = (uint32_t)(h[1] >> 32);
    if ((r.v[0] | r.v[1] | r.v[2] | r.v[3]) == 0) r.v[0] = 1;
    if (!good) { r.v[0] = 0; r.v[1] = 0; r.v[2] = 0; r.v[3] = 0; }
    fr_to_mont(rm, r);
    if (live) {
        a.status[i] = st;
        a.good[i] = good ? 1 : 0;
#pragma unroll
        for (int l = 0; l < 8; l++) a.l_scal[(size_t)i * 8 + l] = r.v[l];
        a.l_idx[i] = i * a.slots + a.pi_point;
    }
    const uint32_t *sp = a.scalars + (size_t)ii * a.scal_stride * 8;
#pragma unroll 1
    for (uint32_t t = 0; t < a.n_var; t++) {
        Fr s, sm, p, pc;
#pragma unroll
        for (int l = 0; l < 8; l++) s.v[l] = sp[t * 8 + l];
        fr_to_mont(sm, s);
        fr_mul(p, sm, rm);
        fr_from_mont(pc, p);
        if (live) {
            const size_t e = (size_t)i * a.n_var + t;
#pragma unroll
            for (int l = 0; l < 8; l++) a.r_scal[e * 8 + l] = pc.v[l];
            a.r_idx[e] = i * a.slots + a.terms[2 * t + 1];
        }
    }
#pragma unroll 1
    for (uint32_t f = 0; f < a.n_fix; f++) {
        Fr s, sm, p;
#pragma unroll
        for (int l = 0; l < 8; l++) s.v[l] = sp[(a.n_var + f) * 8 + l];
        fr_to_mont(sm, s);
        fr_mul(p, sm, rm);               // 0 for lanes that take no part (rm = 0)
#pragma unroll 1
        for (int d = 32; d >= 1; d >>= 1) {
            Fr o;
#pragma unroll
            for (int l = 0; l < 8; l++) o.v[l] = __shfl_down(p.v[l], d);
            fr_add(p, p, o);
        }
        if (lane == 0) {
#pragma unroll
            for (int l = 0; l < 8; l++) a.vk_part[((size_t)blockIdx.x * a.n_fix + f) * 8 + l] = p.v[l];
        }
    }
}

extern "C" __global__ void __launch_bounds__(64)
k_rlc_vk_sum(RlcArgs a, uint32_t n_blocks) {
    const uint32_t f = blockIdx.x * 64 + threadIdx.x;
    if (f >= a.n_fix) return;
    Fr acc, c;
    FrF::set_zero(acc);
#pragma unroll 1
    for (uint32_t b = 0; b < n_blocks; b++) {
        Fr p;
#pragma unroll
        for (int l = 0; l < 8; l++) p.v[l] = a.vk_part[((size_t)b * a.n_fix + f) * 8 + l];
        fr_add(acc, acc, p);
    }
    fr_from_mont(c, acc);
    const size_t e = (size_t)a.n * a.n_var + f;
#pragma unroll
    for (int l = 0; l < 8; l++) a.r_scal[e * 8 + l] = c.v[l];
    a.r_idx[e] = a.n * a.slots + a.terms[2 * (a.n_var + f) + 1];   // pool 1 (the plan's VK bases) starts at n * slots
}

// fall-back path only: the MSM window tables of every per-proof point, which the RLC mode's decompression launch skips
extern "C" __global__ void __launch_bounds__(64)
k_build_tables(uint32_t n_points, const uint32_t *__restrict__ pts, const uint8_t *__restrict__ valid, uint32_t *__restrict__ pt_tab,
               const uint32_t *__restrict__ skip, uint32_t slots) {
    if (skip && skip[0]) return;
    for (uint32_t g = blockIdx.x * 64 + threadIdx.x; g < n_points; g += gridDim.x * 64) {   // (small grid: see k_g1_msm_cond)
        if (skip && skip[1 + ((g / slots) >> 6)]) continue;   // the point's group of 64 proofs passed its own check
        if (!valid[g]) continue;
        G1A p;
#pragma unroll
        for (int k = 0; k < 12; k++) { p.x.v[k] = pts[(size_t)g * 24 + k]; p.y.v[k] = pts[(size_t)g * 24 + 12 + k]; }
        if (g1a_is_inf(p)) continue;
        g1_build_window_tables_glv(pt_tab + (size_t)g * 448, p);
    }
}

// A call that was routed to the per-proof kernels (h2v_capi.hip: rlc_route) still reports what the RLC mode would have met:
// groups of 64 proofs seen, and groups holding a proof that only the pairing rejects (everything rejected earlier takes no
// part in a combination).  One lane per group.
extern "C" __global__ void __launch_bounds__(64)
k_rlc_count_groups(uint32_t n, const uint32_t *__restrict__ status, uint32_t *__restrict__ stats) {
    const uint32_t g = blockIdx.x * 64 + threadIdx.x, n_groups = (n + 63) / 64;
    uint32_t failed = 0;
    if (g < n_groups)
        for (uint32_t i = g * 64; i < n && i < g * 64 + 64; i++) failed |= (status[i] == H2V_ST_PAIRING) ? 1u : 0u;
    const unsigned long long bal = __ballot(failed != 0), live = __ballot(g < n_groups);
    if (threadIdx.x == 0) {
        atomicAdd(stats, (uint32_t)__popcll(live));
        atomicAdd(stats + 1, (uint32_t)__popcll(bal));
    }
}

// Fall-back, stage 1 (after a failed batch check): the term list of every GROUP of 64 proofs - group g is block g of
// k_rlc_prepare - for its own right-hand bucket MSM: the group's slice of the batch's per-proof terms, then the VK bases
// with the group's own scalar sums (vk_part[g], still in Montgomery form).  stride = 64 n_var + n_fix terms per group.
struct RlcGroupArgs {
    uint32_t n, n_var, n_fix, slots, stride;
    const uint32_t *terms, *r_scal, *r_idx, *vk_part;
    uint32_t *g_scal, *g_idx;
};
extern "C" __global__ void __launch_bounds__(64)
k_rlc_group_terms(RlcGroupArgs a, const uint32_t *__restrict__ skip) {
    if (skip[0]) return;
    const uint32_t g = blockIdx.x, lo = g * 64, ng = a.n - lo < 64 ? a.n - lo : 64, nt = ng * a.n_var;
    const size_t src = (size_t)lo * a.n_var, dst = (size_t)g * a.stride;
    for (uint32_t j = threadIdx.x; j < nt; j += 64) {
#pragma unroll
        for (int l = 0; l < 8; l++) a.g_scal[(dst + j) * 8 + l] = a.r_scal[(src + j) * 8 + l];
        a.g_idx[dst + j] = a.r_idx[src + j];
    }
    for (uint32_t f = threadIdx.x; f < a.n_fix; f += 64) {
        Fr p, c;
#pragma unroll
        for (int l = 0; l < 8; l++) p.v[l] = a.vk_part[((size_t)g * a.n_fix + f) * 8 + l];
        fr_from_mont(c, p);
#pragma unroll
        for (int l = 0; l < 8; l++) a.g_scal[(dst + nt + f) * 8 + l] = c.v[l];
        a.g_idx[dst + nt + f] = a.n * a.slots + a.terms[2 * (a.n_var + f) + 1];
    }
}
