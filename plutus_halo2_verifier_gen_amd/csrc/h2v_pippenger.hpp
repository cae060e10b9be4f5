// Bucketed (Pippenger) G1 multi-scalar multiplication for gfx950, and the kernels of the random-linear-combination
// (RLC) batch mode that is its caller.
//
// What it computes: sum_n s_n * P_n over N (scalar, affine base) pairs.  The per-proof MSM of the parity path has T <= 64
// terms and cannot fill buckets; the RLC mode folds a whole batch into ONE check
//     e( sum_i r_i pi_i , s_g2 ) == e( sum_i r_i er_i , G2 )            (r_i: 128-bit, unpredictable to the prover)
// i.e. into one MSM over every per-proof point of the batch (B * n_var terms with scalars r_i * s_{i,t} mod r, plus the
// VK bases with the summed scalars) and one over the B points pi_i - the algebra of
// /root/reference/aiken-verifier/aiken_halo2/lib/halo2_kzg.ak:37-43 (er = final_com + v (-G1) + x3 pi, el = pi) and the
// single pairing equation of aiken-verifier/templates/verification_h2.hbs:125-128, summed over the batch.
//
// Scheme (signed c-bit windows, GLV halves, buckets sorted by size):
//   k_pip_digits      one lane per term: re-cut the base to the 14 x 28-bit lazy form once (x, y, beta x), split the
//                     scalar k = k1 + k2 lambda (GLV: both halves < 2^128, phi(P) = (beta x, y)), recode both halves into
//                     W signed digits in [-2^(c-1), 2^(c-1)], count bucket sizes (one counter per (window, |digit|))
//   k_pip_scan        one block: exclusive prefix sum of the counters, the bucket order by descending size, and the size
//                     CLASSES: a bucket with count in (T 2^(k-1), T 2^k] gets 2^k lanes (k = 0..8), so that every lane
//                     of the accumulation sums at most T entries whatever the digit distribution is (the top window of a
//                     128-bit half holds 128 mod c bits: a few buckets with thousands of entries; without classes those
//                     lanes ran 3 ms chains beside 0.3 ms ones); empty buckets get no lane
//   k_pip_scatter     one lane per term: (term, half, sign) into its bucket's slice of the entry list
//   k_pip_accumulate  THE bulk kernel: buckets in descending size, 2^k neighbouring lanes per bucket of class k (a block
//                     holds one class); each lane sums its slice with mixed additions (8M + 3S, affine base read once per
//                     use as 2 x 56 bytes), the 2^k partial sums meet in an LDS tree.  Exceptional additions (equal /
//                     opposite points, which adversarial or merely repeated proof points can produce) zero the running Z:
//                     detected ONCE at the end of the chain, and that chain is then redone with the complete addition
//   k_pip_reduce      one block per window: bucket partials -> sum_j j B_j by a suffix scan + tree sum in LDS (complete
//                     additions, log depth), then the window's 2^(c w) by doublings
//   k_pip_combine     sum of the W window values -> canonical Jacobian point
// Algorithmic bytes: 128 per term (32 scalar + 96 affine base), as for the per-proof MSM.
#pragma once
#include "h2v_curve28.hpp"
#include "h2v_plan.h"

#define PIP_PT_DW 42          // x, y, beta*x: 3 x 14 limbs per term
#define PIP_PART_DW 44        // Jacobian partial sum: 42 limbs + infinity flag + pad (16-byte multiple)
#define PIP_MAX_W 20
#define PIP_MAX_C 10          // NB = 2^(c-1) <= 512 buckets per window: one block's LDS holds a window's bucket sums
#define PIP_N_CLASSES 9       // lanes per bucket: 2^0 .. 2^8
#define PIP_CLS_DW (3 * PIP_N_CLASSES + 1)   // per class: first rank, end rank, first lane; then the lane total

struct PipArgs {
    uint32_t n;               // terms
    uint32_t c, W, NB;        // window bits, windows per half, buckets per window (2^(c-1))
    uint32_t halves;          // 2: 255-bit scalars split by GLV; 1: scalars below 2^128 (k2 = 0)
    uint32_t chain;           // T: most entries a lane of k_pip_accumulate sums (bucket classes above)
    const uint32_t *scal;     // n x 8 canonical little-endian limbs (< r)
    const uint32_t *pidx;     // n point indices, or NULL for the identity map
    const uint32_t *pool0;    // affine Montgomery points (24 dwords; all-zero = infinity): indices [0, n_pool0)
    uint32_t n_pool0;
    const uint32_t *pool1;    // indices >= n_pool0 (the plan's VK bases), or NULL
    uint32_t *pts28;          // n x PIP_PT_DW
    int16_t *dig;             // n x halves x W signed digits
    uint32_t *cnt;            // W*NB counters, zeroed before k_pip_digits; reused as scatter cursors
    uint32_t *off;            // W*NB + 1 exclusive offsets
    uint32_t *order;          // W*NB bucket ids by descending size
    uint32_t *cls;            // PIP_CLS_DW dwords written by k_pip_scan: the size classes
    uint32_t *list;           // entries: (term * 2 + half) | sign << 31
    uint32_t *partial;        // W*NB x PIP_PART_DW: bucket sums
    uint32_t *wsum;           // W x PIP_PART_DW: window values, already multiplied by 2^(c w)
    uint32_t *out;            // 36 dwords: canonical Jacobian result (Z = 0: infinity)
};
// Up to two independent MSMs share every launch (blockIdx.y picks one): the RLC mode's right- and left-hand sums run their
// phases - and above all their latency-bound tails - side by side without a second stream.
struct PipArgs2 { PipArgs p[2]; };

H2V_DI const uint32_t *pip_point_ptr(const PipArgs &a, uint32_t n) {
    const uint32_t idx = a.pidx ? a.pidx[n] : n;
    return idx < a.n_pool0 ? a.pool0 + (size_t)idx * 24 : a.pool1 + (size_t)(idx - a.n_pool0) * 24;
}

static __device__ __forceinline__ void pip_digits_impl(const PipArgs &a) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.n) return;
    const uint32_t *pp = pip_point_ptr(a, n);
    G1A base;
    uint32_t any_p = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) { base.x.v[k] = pp[k]; base.y.v[k] = pp[12 + k]; any_p |= pp[k] | pp[12 + k]; }
    uint32_t s[8], any_s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { s[k] = a.scal[(size_t)n * 8 + k]; any_s |= s[k]; }
    {
        F28 x, y, bx, b28;
        f28_from_fp(x, base.x);
        f28_from_fp(y, base.y);
        Fp beta;
#pragma unroll
        for (int k = 0; k < 12; k++) beta.v[k] = FP_BETA_GLV[k];
        f28_from_fp(b28, beta);
        f28_mul(bx, x, b28);                       // phi(P) = (beta' x, y)                 (2, 1)
        uint32_t *dst = a.pts28 + (size_t)n * PIP_PT_DW;
#pragma unroll
        for (int k = 0; k < 14; k++) { dst[k] = x.l[k]; dst[14 + k] = y.l[k]; dst[28 + k] = bx.l[k]; }
    }
    const bool skip = any_p == 0 || any_s == 0;    // the point at infinity / a zero scalar contribute nothing
    uint32_t k1[4], k2[4];
    if (a.halves == 2) glv_split(k1, k2, s);
    else {
#pragma unroll
        for (int k = 0; k < 4; k++) { k1[k] = s[k]; k2[k] = 0; }
    }
    const uint32_t c = a.c, NB = a.NB, mask = (1u << c) - 1u;
#pragma unroll 1
    for (uint32_t h = 0; h < a.halves; h++) {
        uint32_t carry = 0;
#pragma unroll 1
        for (uint32_t w = 0; w < a.W; w++) {
            // bits [w c, w c + c) of the 128-bit half (zero beyond bit 127)
            const uint32_t bit = w * c, word = bit >> 5, sh = bit & 31;
            uint32_t kw[5];
#pragma unroll
            for (int k = 0; k < 4; k++) kw[k] = h ? k2[k] : k1[k];
            kw[4] = 0;
            uint64_t two = 0;
            if (word < 4) two = (uint64_t)kw[word] | ((uint64_t)kw[word + 1] << 32);
            uint32_t raw = ((uint32_t)(two >> sh) & mask) + carry;
            int d;
            if (raw > NB) { d = (int)raw - (int)(mask + 1u); carry = 1; } else { d = (int)raw; carry = 0; }
            if (skip) d = 0;
            a.dig[((size_t)n * a.halves + h) * a.W + w] = (int16_t)d;
            if (d != 0) atomicAdd(&a.cnt[w * NB + (uint32_t)(d < 0 ? -d : d) - 1u], 1u);
        }
        // W c > 128 and the top window holds fewer than c - 1 bits + carry <= NB: no carry leaves the last window
    }
}

// One block of 1024 threads: off[] = exclusive prefix sum of cnt[] (nb buckets, + the total at off[nb]); order[] = bucket
// ids by descending count (counting sort on min(count, 4095)); cls[] = the size classes; cnt[] is cleared (it becomes
// the scatter cursors).
static __device__ __forceinline__ void pip_scan_impl(const PipArgs &a) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t hist[4096];   // (T 2^7 = 2560 for the default chain T = 20 must be resolved: the class rule below)
    const uint32_t nb = a.W * a.NB, tid = threadIdx.x;
    const uint32_t per = (nb + 1023) / 1024, lo = tid * per, hi = lo + per < nb ? lo + per : nb;
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; b++) sum += a.cnt[b];
    part[tid] = sum;
    hist[tid] = 0; hist[tid + 1024] = 0; hist[tid + 2048] = 0; hist[tid + 3072] = 0;
    __syncthreads();
    // inclusive scan of the 1024 partial sums (Hillis-Steele in place, a barrier on either side of the update)
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = tid ? part[tid - 1] : 0u;
    for (uint32_t b = lo; b < hi; b++) {
        const uint32_t cv = a.cnt[b];
        a.off[b] = run;
        run += cv;
        atomicAdd(&hist[cv < 4095u ? cv : 4095u], 1u);
    }
    if (tid == 1023) a.off[nb] = part[1023];
    __syncthreads();
    if (tid == 0) {
        // hist[v] := number of buckets with count > v = the first rank of count v in descending order
        uint32_t acc = 0;
        for (int v = 4095; v >= 0; v--) { const uint32_t hv = hist[v]; hist[v] = acc; acc += hv; }
        // classes, largest buckets first: class k = counts in (T 2^(k-1), T 2^k] (k = 8: everything above T 2^7; k = 0:
        // 1 .. T); ranks [first, end) of the descending order, lanes from a multiple of 256 (a block holds one class)
        uint32_t lane = 0, first = 0;
        for (int k = PIP_N_CLASSES - 1; k >= 0; k--) {
            const uint32_t low = k == 0 ? 0u : (a.chain << (k - 1));       // the class holds counts > low
            // buckets with count > low.  (A forced chain above 31 puts T 2^(k-1) beyond the histogram: those classes then
            // start at 4095 entries - more entries per lane than T, never a lane short: the kernel below walks every
            // logical block whatever the grid is.)
            const uint32_t end = hist[low < 4095u ? low : 4094u];
            a.cls[3 * k] = first; a.cls[3 * k + 1] = end; a.cls[3 * k + 2] = lane;
            lane += (((end - first) << k) + 255u) & ~255u;
            first = end;
        }
        a.cls[3 * PIP_N_CLASSES] = lane;
    }
    __syncthreads();
    for (uint32_t b = lo; b < hi; b++) {
        const uint32_t cv = a.cnt[b];
        const uint32_t pos = atomicAdd(&hist[cv < 4095u ? cv : 4095u], 1u);
        a.order[pos] = b;
        a.cnt[b] = 0;
    }
}

static __device__ __forceinline__ void pip_scatter_impl(const PipArgs &a) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.n) return;
#pragma unroll 1
    for (uint32_t h = 0; h < a.halves; h++) {
#pragma unroll 1
        for (uint32_t w = 0; w < a.W; w++) {
            const int d = a.dig[((size_t)n * a.halves + h) * a.W + w];
            if (d == 0) continue;
            const uint32_t b = w * a.NB + (uint32_t)(d < 0 ? -d : d) - 1u;
            const uint32_t pos = a.off[b] + atomicAdd(&a.cnt[b], 1u);
            a.list[pos] = ((n << 1) | h) | (d < 0 ? 0x80000000u : 0u);
        }
    }
}

// acc (+flag) += (+-) the affine entry, complete (used by the slow path only)
H2V_DN void pip_add_entry_complete(G1J28 &acc, bool &acc_inf, const uint32_t *pt, const uint32_t half, const bool neg) {
    G1J28 o;
#pragma unroll
    for (int k = 0; k < 14; k++) { o.x.l[k] = pt[(half ? 28 : 0) + k]; o.y.l[k] = pt[14 + k]; }
    f28_set_one(o.z);
    g1j28_acc_add(acc, acc_inf, o, neg);
}

// The hot loop: sum of the entries [e0, e1) of the list with unchecked mixed additions (multiplier inlined).  Out of line so
// that its registers are its own: inside the kernel body the bucket bookkeeping that must survive the loop pushed seven
// values per iteration through scratch (115 MiB of HBM writes per launch for 1.2 MB of results).
H2V_DN void pip_sum_slice(G1J28 &acc_out, bool &inf_out, const uint32_t *__restrict__ list, const uint32_t *__restrict__ pts28,
                          const uint32_t e0, const uint32_t e1) {
    G1J28 acc;
    bool inf = true;
#pragma unroll 1
    for (uint32_t e = e0; e < e1; e++) {
        const uint32_t ent = list[e];
        const uint32_t *pt = pts28 + (size_t)((ent & 0x7fffffffu) >> 1) * PIP_PT_DW;
        const bool neg = (ent >> 31) != 0;
        const uint32_t xo = (ent & 1u) ? 28u : 0u;
        F28 qx, qy;
#pragma unroll
        for (int k = 0; k < 14; k++) { qx.l[k] = pt[xo + k]; qy.l[k] = pt[14 + k]; }
        if (inf) {
            acc.x = qx;
            acc.y = qy;
            if (neg) { F28_NEG(acc.y, qy, 3, 1); f28_carry(acc.y); }
            f28_set_one(acc.z);
            inf = false;
        } else {
            g1j28_madd_ladder_t<true>(acc, acc, qx, qy, neg);   // unchecked: the caller tests Z afterwards
        }
    }
    if (!inf) acc_out = acc;
    inf_out = inf;
}

static __device__ __forceinline__ void pip_accumulate_impl(const PipArgs &a) {
    __shared__ uint32_t red[43 * 256];   // partial sums of the block's lanes (dword d of thread t at red[d * 256 + t])
    const uint32_t tid = threadIdx.x;
    // The grid is the host's ESTIMATE of the lanes the class rule hands out; the blocks walk the logical blocks, so an
    // estimate that is short (round 2: counts between 2047 and T 2^7 were put in the 256-lane class by a clamped histogram,
    // and batches of 53-62 k simple_mul proofs left their last buckets without a lane) costs time, never a bucket.
    const uint32_t n_logical = (a.cls[3 * PIP_N_CLASSES] + 255u) >> 8;
#pragma unroll 1
    for (uint32_t blk = blockIdx.x; blk < n_logical; blk += gridDim.x) {
    const uint32_t g = blk * 256 + tid;
    // the block's class k (lane ranges are 256-aligned and laid out from k = 8 down to 0; empty classes have no lanes)
    uint32_t k = 0;
#pragma unroll 1
    for (int kk = PIP_N_CLASSES - 1; kk >= 0; kk--) {
        const uint32_t base = a.cls[3 * kk + 2], lanes = (a.cls[3 * kk + 1] - a.cls[3 * kk]) << kk;
        if (blk * 256 >= base && blk * 256 < base + ((lanes + 255u) & ~255u)) k = (uint32_t)kk;
    }
    const uint32_t lpb = 1u << k, rel = g - a.cls[3 * k + 2];
    const uint32_t rank = a.cls[3 * k] + (rel >> k), q = rel & (lpb - 1u);
    const bool live = rank < a.cls[3 * k + 1];
    const uint32_t b = a.order[live ? rank : 0u];
    const uint32_t lo = a.off[b], cnt = live ? a.off[b + 1] - lo : 0u;
    const uint32_t e0 = lo + (uint32_t)(((uint64_t)cnt * q) >> k), e1 = lo + (uint32_t)(((uint64_t)cnt * (q + 1)) >> k);
    G1J28 acc;
    bool inf = true;
    pip_sum_slice(acc, inf, a.list, a.pts28, e0, e1);
    if (!inf) {
        // An exceptional addition (acc == +-entry) has H = 0 and leaves Z3 = Z1 H = 0, which every later Z inherits.
        Fp zc;
        F28 z = acc.z;
        f28_carry(z);
        f28_to_fp(zc, z);
        if (fp_is_zero(zc)) {   // redo this slice with the complete group law (rare: repeated / opposite points)
            inf = true;
#pragma unroll 1
            for (uint32_t e = e0; e < e1; e++) {
                const uint32_t ent = a.list[e];
                pip_add_entry_complete(acc, inf, a.pts28 + (size_t)((ent & 0x7fffffffu) >> 1) * PIP_PT_DW, ent & 1u, (ent >> 31) != 0);
            }
        }
    }
    // the 2^k lanes of a bucket are neighbours in the block: tree through LDS, complete additions
#define PIP_ACC_STORE()                                                                      \
    do {                                                                                     \
        _Pragma("unroll") for (int k = 0; k < 14; k++) {                                     \
            red[k * 256 + tid] = acc.x.l[k]; red[(14 + k) * 256 + tid] = acc.y.l[k]; red[(28 + k) * 256 + tid] = acc.z.l[k]; \
        }                                                                                    \
        red[42 * 256 + tid] = inf ? 1u : 0u;                                                 \
    } while (0)
    if (lpb > 1) {
        PIP_ACC_STORE();
        __syncthreads();
        for (uint32_t s = lpb >> 1; s >= 1; s >>= 1) {
            if (q < s && red[42 * 256 + tid + s] == 0) {
                G1J28 o;
#pragma unroll
                for (int k = 0; k < 14; k++) { o.x.l[k] = red[k * 256 + tid + s]; o.y.l[k] = red[(14 + k) * 256 + tid + s]; o.z.l[k] = red[(28 + k) * 256 + tid + s]; }
                g1j28_acc_add(acc, inf, o, false);
                PIP_ACC_STORE();
            }
            __syncthreads();
        }
    }
#undef PIP_ACC_STORE
    if (live && q == 0) {
        uint32_t *dst = a.partial + (size_t)b * PIP_PART_DW;
        if (!inf) {
#pragma unroll
            for (int k = 0; k < 14; k++) { dst[k] = acc.x.l[k]; dst[14 + k] = acc.y.l[k]; dst[28 + k] = acc.z.l[k]; }
        }
        dst[42] = inf ? 1u : 0u;
    }
    __syncthreads();   // (red[] is reused by the next logical block)
    }
}

// (f28_bcast, g1j28_dbl_n_coop3: h2v_curve28.hpp - the per-proof MSM's small-launch shape uses them too)

// One block per window w, NB of its threads at work.  Thread t owns bucket j = t + 1 and computes
//   suffix scan  S_t = sum_{u >= t} B_u   (Hillis-Steele, log2 NB rounds)      and      T = sum_t S_t = sum_j j B_j
// with the values of the other threads read from LDS (dword d of thread t at red[d * NB + t]: conflict-free) and the
// thread's own value in registers.  All additions are complete.  Thread 0 then applies the window's weight 2^(c w).
static __device__ __forceinline__ void pip_reduce_impl(const PipArgs &a) {
    extern __shared__ uint32_t red[];
    const uint32_t w = blockIdx.x, t = threadIdx.x, NB = a.NB;
    if (w >= a.W) return;                       // (uniform per block: the grid covers the wider of the two MSMs)
    const bool mine = t < NB;                   // (the block is as wide as the larger NB of the two)
    G1J28 val;
    bool inf = true;
    if (mine && a.off[w * NB + t + 1] != a.off[w * NB + t]) {   // an empty bucket had no lane: nothing was written for it
        const uint32_t *src = a.partial + (size_t)(w * NB + t) * PIP_PART_DW;
        if (!src[42]) {
#pragma unroll
            for (int k = 0; k < 14; k++) { val.x.l[k] = src[k]; val.y.l[k] = src[14 + k]; val.z.l[k] = src[28 + k]; }
            inf = false;
        }
    }
#define PIP_RED_STORE()                                                                       \
    do {                                                                                      \
        _Pragma("unroll") for (int k = 0; k < 14; k++) {                                      \
            red[k * NB + t] = val.x.l[k]; red[(14 + k) * NB + t] = val.y.l[k]; red[(28 + k) * NB + t] = val.z.l[k]; \
        }                                                                                     \
        red[42 * NB + t] = inf ? 1u : 0u;                                                     \
    } while (0)
#define PIP_RED_LOAD(o, src_t)                                                                \
    do {                                                                                      \
        _Pragma("unroll") for (int k = 0; k < 14; k++) {                                      \
            (o).x.l[k] = red[k * NB + (src_t)]; (o).y.l[k] = red[(14 + k) * NB + (src_t)]; (o).z.l[k] = red[(28 + k) * NB + (src_t)]; \
        }                                                                                     \
    } while (0)
    if (mine) PIP_RED_STORE();
    __syncthreads();
    for (uint32_t d = 1; d < NB; d <<= 1) {
        const bool take = mine && t + d < NB && red[42 * NB + t + d] == 0;
        G1J28 o;
        if (take) PIP_RED_LOAD(o, t + d);
        __syncthreads();
        if (take) g1j28_acc_add(val, inf, o, false);
        if (mine) PIP_RED_STORE();
        __syncthreads();
    }
    for (uint32_t s = NB >> 1; s >= 1; s >>= 1) {
        if (t < s && red[42 * NB + t + s] == 0) {
            G1J28 o;
            PIP_RED_LOAD(o, t + s);
            g1j28_acc_add(val, inf, o, false);
            PIP_RED_STORE();
        }
        __syncthreads();
    }
#undef PIP_RED_STORE
#undef PIP_RED_LOAD
    if (t < 64) {   // the block's first wave: the window's weight 2^(c w), the doubling's multiplications on three lanes
        const bool inf0 = __builtin_amdgcn_readfirstlane(inf ? 1 : 0) != 0;
        if (!inf0 && a.c * w) g1j28_dbl_n_coop3(val, a.c * w);   // odd group order: never infinity
    }
    if (t == 0) {
        uint32_t *dst = a.wsum + (size_t)w * PIP_PART_DW;
        if (!inf) {
#pragma unroll
            for (int k = 0; k < 14; k++) { dst[k] = val.x.l[k]; dst[14 + k] = val.y.l[k]; dst[28 + k] = val.z.l[k]; }
        }
        dst[42] = inf ? 1u : 0u;
    }
}

// sum of the W weighted window values (one wave, tree through LDS) -> canonical Jacobian coordinates
static __device__ __forceinline__ void pip_combine_impl(const PipArgs &a) {
    __shared__ uint32_t red[43 * 64];
    const uint32_t t = threadIdx.x;
    G1J28 val;
    bool inf = true;
    if (t < a.W) {
        const uint32_t *src = a.wsum + (size_t)t * PIP_PART_DW;
        if (!src[42]) {
#pragma unroll
            for (int k = 0; k < 14; k++) { val.x.l[k] = src[k]; val.y.l[k] = src[14 + k]; val.z.l[k] = src[28 + k]; }
            inf = false;
        }
    }
#define PIP_CMB_STORE()                                                                      \
    do {                                                                                     \
        _Pragma("unroll") for (int k = 0; k < 14; k++) {                                     \
            red[k * 64 + t] = val.x.l[k]; red[(14 + k) * 64 + t] = val.y.l[k]; red[(28 + k) * 64 + t] = val.z.l[k]; \
        }                                                                                    \
        red[42 * 64 + t] = inf ? 1u : 0u;                                                    \
    } while (0)
    PIP_CMB_STORE();
    __syncthreads();
    for (uint32_t s = 16; s >= 1; s >>= 1) {
        if (t < s && red[42 * 64 + t + s] == 0) {
            G1J28 o;
#pragma unroll
            for (int k = 0; k < 14; k++) { o.x.l[k] = red[k * 64 + t + s]; o.y.l[k] = red[(14 + k) * 64 + t + s]; o.z.l[k] = red[(28 + k) * 64 + t + s]; }
            g1j28_acc_add(val, inf, o, false);
            PIP_CMB_STORE();
        }
        __syncthreads();
    }
#undef PIP_CMB_STORE
    if (t == 0) {
        G1J r;
        g1j28_to_g1j(r, val, inf);
#pragma unroll
        for (int k = 0; k < 12; k++) { a.out[k] = r.x.v[k]; a.out[12 + k] = r.y.v[k]; a.out[24 + k] = r.z.v[k]; }
    }
}

// ---- kernel entry points.  Two forms of every phase: the arguments of up to two MSMs by value (the batch check of the RLC
// mode: right- and left-hand sum side by side), or any number of MSMs from an argument array in device memory (blockIdx.y
// picks one; the group checks of the RLC fall-back: 2 x ceil(n / 64) small MSMs in one launch per phase), which return at
// once when *skip != 0 (the batch check passed: there is nothing to localise).
#define PIP_ENTRY(kname, impl, ...)                                                                  \
    extern "C" __global__ void __launch_bounds__(__VA_ARGS__) kname(PipArgs2 a2) { impl(a2.p[blockIdx.y]); }       \
    extern "C" __global__ void __launch_bounds__(__VA_ARGS__) kname##_many(const PipArgs *__restrict__ args, const uint32_t *__restrict__ skip) { \
        if (skip && skip[0]) return;                                                                 \
        const PipArgs a = args[blockIdx.y];                                                          \
        impl(a);                                                                                     \
    }
PIP_ENTRY(k_pip_digits, pip_digits_impl, 256)
PIP_ENTRY(k_pip_scan, pip_scan_impl, 1024)
PIP_ENTRY(k_pip_scatter, pip_scatter_impl, 256)
PIP_ENTRY(k_pip_accumulate, pip_accumulate_impl, 256, 2)
PIP_ENTRY(k_pip_reduce, pip_reduce_impl, 512)
PIP_ENTRY(k_pip_combine, pip_combine_impl, 64)
#undef PIP_ENTRY

// Bucket sums -> result for MANY SMALL MSMs (the group checks of the RLC fall-back), two launches.
//   k_pip_wsum_many    a QUAD of lanes per (MSM, window), densely packed (16 windows per wave).  Lane s of the quad takes
//                      the buckets t in [s NB/4, (s+1) NB/4) and the running-sum rule - running += B_t, total += running,
//                      from the top down - gives  T_s = sum (t - lo_s + 1) B_t  and  U_s = sum B_t;  lane 0 of the quad
//                      puts the window together:  S = sum_j j B_j = sum_s T_s + (NB/4) (U_1 + 2 U_2 + 3 U_3).
//                      2 NB/4 + 9 complete additions and log2(NB/4) doublings deep, no tree.
//   k_pip_horner_many  one wave per MSM folds the W window sums from the top window down, acc = 2^c acc + S_w: c (W - 1)
//                      doublings in all (each spread over three lanes of a quad), where weighting every window by itself
//                      (k_pip_reduce) costs c W^2 / 2.
// Few lanes and long chains - wrong for one big MSM, right for hundreds side by side: for 128 MSMs of 19 windows x 64
// buckets the tree kernels issued more field multiplications than the bucket accumulation itself.
H2V_DN void pip_acc_add_ool(G1J28 &acc, bool &inf, const G1J28 &q) { g1j28_acc_add(acc, inf, q, false); }
#define PIP_LDS_PUT(buf, stride, idx, val, flag)                                                  \
    do {                                                                                          \
        if (!(flag)) {                                                                            \
            _Pragma("unroll") for (int k = 0; k < 14; k++) {                                      \
                (buf)[k * (stride) + (idx)] = (val).x.l[k]; (buf)[(14 + k) * (stride) + (idx)] = (val).y.l[k]; (buf)[(28 + k) * (stride) + (idx)] = (val).z.l[k]; \
            }                                                                                     \
        }                                                                                         \
        (buf)[42 * (stride) + (idx)] = (flag) ? 1u : 0u;                                          \
    } while (0)
#define PIP_LDS_ADD(buf, stride, idx, acc, acc_inf)                                               \
    do {                                                                                          \
        if ((buf)[42 * (stride) + (idx)] == 0) {                                                  \
            G1J28 v_;                                                                             \
            _Pragma("unroll") for (int k = 0; k < 14; k++) {                                      \
                v_.x.l[k] = (buf)[k * (stride) + (idx)]; v_.y.l[k] = (buf)[(14 + k) * (stride) + (idx)]; v_.z.l[k] = (buf)[(28 + k) * (stride) + (idx)]; \
            }                                                                                     \
            pip_acc_add_ool(acc, acc_inf, v_);                                                    \
        }                                                                                         \
    } while (0)
extern "C" __global__ void __launch_bounds__(64)
k_pip_wsum_many(const PipArgs *__restrict__ args, uint32_t n_problems, uint32_t w_max, const uint32_t *__restrict__ skip) {
    if (skip && skip[0]) return;
    __shared__ uint32_t tot_s[43 * 64], run_s[43 * 64];
    const uint32_t tid = threadIdx.x, seg = tid & 3u, pw = (blockIdx.x * 64 + tid) >> 2;
    const uint32_t prob = pw / w_max, w = pw % w_max;
    const bool live = prob < n_problems && w < args[prob < n_problems ? prob : 0].W;
    const PipArgs a = args[live ? prob : 0];
    const uint32_t nbs = a.NB >> 2;                          // buckets per segment (NB >= 4: c >= 3)
    G1J28 run, tot;
    bool run_inf = true, tot_inf = true;
    if (live) {
#pragma unroll 1
        for (int t = (int)((seg + 1) * nbs) - 1; t >= (int)(seg * nbs); t--) {
            const uint32_t b = w * a.NB + (uint32_t)t;
            if (a.off[b + 1] != a.off[b]) {                  // (an empty bucket had no lane: nothing was written for it)
                const uint32_t *src = a.partial + (size_t)b * PIP_PART_DW;
                if (!src[42]) {
                    G1J28 v;
#pragma unroll
                    for (int k = 0; k < 14; k++) { v.x.l[k] = src[k]; v.y.l[k] = src[14 + k]; v.z.l[k] = src[28 + k]; }
                    pip_acc_add_ool(run, run_inf, v);
                }
            }
            if (!run_inf) pip_acc_add_ool(tot, tot_inf, run);
        }
    }
    PIP_LDS_PUT(tot_s, 64, tid, tot, tot_inf);
    PIP_LDS_PUT(run_s, 64, tid, run, run_inf);
    __syncthreads();
    if (live && seg == 0) {
        // U_1 + 2 U_2 + 3 U_3 = U_3 + (U_3 + U_2) + (U_3 + U_2 + U_1)
        G1J28 v, acc;
        bool v_inf = true, acc_inf = true;
#pragma unroll 1
        for (int q = 3; q >= 1; q--) {
            PIP_LDS_ADD(run_s, 64, tid + q, v, v_inf);
            if (!v_inf) pip_acc_add_ool(acc, acc_inf, v);
        }
        if (!acc_inf)
            for (uint32_t m = nbs; m > 1; m >>= 1) g1j28_dbl_ool(acc, acc);    // x NB/4 (odd group order: never infinity)
#pragma unroll 1
        for (int q = 0; q < 4; q++) PIP_LDS_ADD(tot_s, 64, tid + q, acc, acc_inf);
        uint32_t *dst = a.wsum + (size_t)w * PIP_PART_DW;
        if (!acc_inf) {
#pragma unroll
            for (int k = 0; k < 14; k++) { dst[k] = acc.x.l[k]; dst[14 + k] = acc.y.l[k]; dst[28 + k] = acc.z.l[k]; }
        }
        dst[42] = acc_inf ? 1u : 0u;
    }
}
#undef PIP_LDS_PUT
#undef PIP_LDS_ADD
extern "C" __global__ void __launch_bounds__(64)
k_pip_horner_many(const PipArgs *__restrict__ args, const uint32_t *__restrict__ skip) {
    if (skip && skip[0]) return;
    const PipArgs a = args[blockIdx.x];
    // every lane of the wave runs the same fold on the same values (the doublings need whole quads in step)
    G1J28 acc;
    bool inf = true;
#pragma unroll 1
    for (int t = (int)a.W - 1; t >= 0; t--) {
        if (!inf) g1j28_dbl_n_coop3(acc, a.c);                // odd group order: never infinity
        const uint32_t *src = a.wsum + (size_t)t * PIP_PART_DW;
        if (!src[42]) {
            G1J28 v;
#pragma unroll
            for (int k = 0; k < 14; k++) { v.x.l[k] = src[k]; v.y.l[k] = src[14 + k]; v.z.l[k] = src[28 + k]; }
            pip_acc_add_ool(acc, inf, v);
        }
    }
    if (threadIdx.x == 0) {
        G1J r;
        g1j28_to_g1j(r, acc, inf);
#pragma unroll
        for (int k = 0; k < 12; k++) { a.out[k] = r.x.v[k]; a.out[12 + k] = r.y.v[k]; a.out[24 + k] = r.z.v[k]; }
    }
}
