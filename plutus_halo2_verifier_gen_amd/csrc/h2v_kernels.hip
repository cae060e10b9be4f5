// gfx950 kernels of the Halo2/KZG batch verifier (one independent proof per lane / lane group).
//
//   k_transcript_combiner   P1-P6 scalars: blake2b transcript replay + Fr combiner + final MSM scalar vector
//                           (interprets the plan's straight-line program; wave-uniform control flow)
//   k_g1_decompress         48-byte zcash G1 -> affine Montgomery, with on-curve + subgroup validation
//   k_g1_msm                er = sum_t s_t * B_t per proof: one lane per term, wave/LDS segmented reduction
//   k_pairing_check         accept <=> e(pi, s_g2) == e(er, G2): 2 Miller loops over precomputed lines of the
//                           FIXED G2 arguments + one final exponentiation, fused in one kernel
//
// Algorithm of record: /root/reference/aiken-verifier/templates/verification_h2.hbs:21-129 and the library it
// calls (aiken-verifier/aiken_halo2/lib/{transcript,lagrange,halo2_kzg,bls_utils}.ak); hash plug-in
// /root/reference/src/plutus_gen/adjusted_types/mod.rs:30-72.
#include <hip/hip_runtime.h>
#include "h2v_curve28.hpp"
#include "h2v_plan.h"
#include "h2v_tower.hpp"
#include "h2v_pairing_coop.hpp"
#include "h2v_pairing_six.hpp"

// ============================================================================ blake2b-256 (RFC 7693)
__device__ static constexpr uint64_t B2_IV[8] = {
    0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
    0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
__device__ static constexpr uint8_t B2_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

H2V_DI uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

// One compression.  The 128-byte message block is read from LDS (dword d of the calling lane at msg[d * 64]) and the
// twelve rounds are a real loop whose message schedule is looked up per round (a scalar load of the sigma row, sixteen
// 64-bit LDS reads): 2.5 KB of code instead of the 24 KB of the unrolled form.  On the combiner's lone waves the
// unrolled form ran at the speed of instruction fetch - every instruction executed once per call, 32k cycles per
// compression - while this one stays in the instruction cache.
typedef __attribute__((address_space(3))) uint32_t h2v_lds_u32;
H2V_DN void b2_compress_lds(uint64_t (&h)[8], const uint32_t *msg_generic, uint64_t t, bool last) {
    const h2v_lds_u32 *msg = (const h2v_lds_u32 *)msg_generic;
    uint64_t v[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = B2_IV[i]; }
    v[12] ^= t;
    if (last) v[14] = ~v[14];
#define B2G(a, b, c, d, x, y)                                                                             \
    v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32); v[c] = v[c] + v[d]; v[b] = rotr64(v[b] ^ v[c], 24); \
    v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = rotr64(v[b] ^ v[c], 63);
#pragma unroll 1
    for (int r = 0; r < 12; r++) {
        uint64_t w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t sg = B2_SIGMA[r][i];
            w[i] = (uint64_t)msg[(2 * sg) * 64] | ((uint64_t)msg[(2 * sg + 1) * 64] << 32);
        }
        B2G(0, 4, 8, 12, w[0], w[1])
        B2G(1, 5, 9, 13, w[2], w[3])
        B2G(2, 6, 10, 14, w[4], w[5])
        B2G(3, 7, 11, 15, w[6], w[7])
        B2G(0, 5, 10, 15, w[8], w[9])
        B2G(1, 6, 11, 12, w[10], w[11])
        B2G(2, 7, 8, 13, w[12], w[13])
        B2G(3, 4, 9, 14, w[14], w[15])
    }
#undef B2G
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}

// Running transcript hash of one lane.  The 128-byte block buffer lives in LDS, dword d of lane l at
// sbuf[d*64 + l] (conflict-free); fill level and byte counter are identical in every lane (the proof layout is
// static), so they are wave-uniform scalars.
struct Transcript {
    uint64_t h[8];
    uint32_t t;       // bytes already compressed
    uint32_t buflen;  // bytes waiting in the LDS block
};

H2V_DI void tr_init(Transcript &s) {
#pragma unroll
    for (int i = 0; i < 8; i++) s.h[i] = B2_IV[i];
    s.h[0] ^= 0x01010000ull ^ 32ull;  // digest 32, key 0, fanout 1, depth 1 (blake2b_simd Params::new().hash_length(32))
    s.t = 0;
    s.buflen = 0;
}
H2V_DI void tr_put(Transcript &s, uint32_t *sbuf, int lane, uint32_t byte) {
    if (s.buflen == 128) {  // full and more input follows: not the last block
        s.t += 128;
        // through a copy: handing s.h itself to the out-of-line function would pin the whole struct - fill level and
        // byte counter included - in private memory, and every tr_put would then wait for a scratch round trip
        uint64_t hh[8];
#pragma unroll
        for (int i = 0; i < 8; i++) hh[i] = s.h[i];
        b2_compress_lds(hh, sbuf + lane, s.t, false);
#pragma unroll
        for (int i = 0; i < 8; i++) s.h[i] = hh[i];
        s.buflen = 0;
    }
    reinterpret_cast<uint8_t *>(&sbuf[(s.buflen >> 2) * 64 + lane])[s.buflen & 3] = (uint8_t)byte;
    s.buflen++;
}
// digest of everything absorbed so far, leaving the running state untouched (State::finalize on a clone).  The unused
// tail of the block buffer is zeroed in place (the padding of the last block); the pending bytes stay where they are.
H2V_DI void tr_digest(const Transcript &s, uint32_t *sbuf, int lane, uint64_t (&out)[4]) {
    uint64_t h[8];
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = s.h[i];
    const uint32_t bl = s.buflen;
    if (bl & 3) sbuf[(bl >> 2) * 64 + lane] &= (1u << (8 * (bl & 3))) - 1u;
#pragma unroll 1
    for (uint32_t d = (bl + 3) >> 2; d < 32; d++) sbuf[d * 64 + lane] = 0;
    b2_compress_lds(h, sbuf + lane, (uint64_t)s.t + bl, true);
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = h[i];
}
// blake2b-256 of a 32-byte message; borrows the lane's block buffer (saved and restored around the call)
H2V_DI void b2_hash32(const uint64_t (&in)[4], uint64_t (&out)[4], uint32_t *sbuf, int lane) {
    uint64_t h[8];
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = B2_IV[i];
    h[0] ^= 0x01010000ull ^ 32ull;
    uint32_t saved[32];
#pragma unroll
    for (int d = 0; d < 32; d++) saved[d] = sbuf[d * 64 + lane];
#pragma unroll
    for (int d = 0; d < 32; d++) sbuf[d * 64 + lane] = d < 8 ? (uint32_t)(in[d >> 1] >> (32 * (d & 1))) : 0u;
    b2_compress_lds(h, sbuf + lane, 32, true);
#pragma unroll
    for (int d = 0; d < 32; d++) sbuf[d * 64 + lane] = saved[d];
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = h[i];
}

// ============================================================================ K1+K3: transcript + Fr combiner
// Register file in global memory: limb l of register r of proof i at regs[(r*8 + l)*stride + i]
// (lanes of a wave touch consecutive dwords).
H2V_DI void reg_load(Fr &v, const uint32_t *regs, uint32_t r, uint32_t stride, uint32_t i) {
#pragma unroll
    for (int l = 0; l < 8; l++) v.v[l] = regs[(size_t)(r * 8 + l) * stride + i];
}
H2V_DI void reg_store(uint32_t *regs, uint32_t r, uint32_t stride, uint32_t i, const Fr &v) {
#pragma unroll
    for (int l = 0; l < 8; l++) regs[(size_t)(r * 8 + l) * stride + i] = v.v[l];
}
// The same register file in LDS when the plan is small enough for a few dozen proofs per CU: limb l of register r of
// the block's proof slot q at vm_lds[(r*8 + l)*P + q].  With 64 waves on a 256-CU chip this kernel is pure latency, and
// a dependent global load + store per instruction was most of it.
extern __shared__ uint32_t vm_lds[];
struct RegsGlobal {
    uint32_t *regs;
    uint32_t stride, i_load, i_store;
    H2V_DI void load(Fr &v, uint32_t r) const { reg_load(v, regs, r, stride, i_load); }
    H2V_DI void store(uint32_t r, const Fr &v) const { reg_store(regs, r, stride, i_store, v); }
};
struct RegsLds {
    uint32_t P, q;
    H2V_DI void load(Fr &v, uint32_t r) const {
#pragma unroll
        for (int l = 0; l < 8; l++) v.v[l] = vm_lds[(r * 8 + l) * P + q];
    }
    H2V_DI void store(uint32_t r, const Fr &v) const {
#pragma unroll
        for (int l = 0; l < 8; l++) vm_lds[(r * 8 + l) * P + q] = v.v[l];
    }
};
H2V_DI void fr_const(Fr &v, const uint32_t *c) {
#pragma unroll
    for (int l = 0; l < 8; l++) v.v[l] = c[l];
}
// absorb 0x01 || 32 little-endian bytes of a canonical (non-Montgomery) scalar
H2V_DI void tr_absorb_scalar(Transcript &s, uint32_t *sbuf, int lane, const Fr &plain) {
    tr_put(s, sbuf, lane, 1);
#pragma unroll
    for (int k = 0; k < 32; k++) tr_put(s, sbuf, lane, (plain.v[k >> 2] >> (8 * (k & 3))) & 0xff);   // static limb indices
}

// The program is a sequence of bundles of L records (h2v_plan.h); lane `sub` of the proof's L lanes executes record `sub`
// of each bundle.  The L lanes share the proof's register file (LDS), sit in the same wave, and a bundle's records are
// independent, so program order is all the synchronisation there is: LDS serves one wave's accesses in issue order.
// The transcript operations are on lane 0 by construction (the loader checks), so only that lane's hash state is real.
template <class RF>
H2V_DI void vm_run(const H2vDevPlan &plan, const RF rf, uint32_t *sbuf, const int lane, const uint32_t i, const uint32_t ii,
                   const bool live, const uint32_t L, const uint32_t sub, uint32_t *st_red /* 64 dwords of LDS */,
                   const uint32_t P /* proof slots of the block: lane l serves slot l & (P - 1) */,
                   const uint8_t *__restrict__ proofs, const uint64_t *__restrict__ proof_off,
                   const uint8_t *__restrict__ instances, const uint8_t *__restrict__ committed,
                   uint32_t *__restrict__ scalars, uint32_t *__restrict__ status, uint32_t *__restrict__ trace) {
    const uint64_t off0 = proof_off[ii];
    const uint64_t plen = proof_off[ii + 1] - off0;
    uint32_t st = 0;
    const bool short_proof = plen < plan.proof_len;
    if (short_proof) st |= H2V_ST_SHORT_PROOF;
    // a short proof is rejected up front; its lane replays a zero-length-safe window (no out-of-bounds reads)
    const uint8_t *proof = proofs + off0;
    // Pull the proof into the cache hierarchy up front: the proof's L lanes touch one byte per 64-byte line, all loads in
    // flight together.  Without it every READ_SCALAR / READ_POINT of the lone wave waits out an HBM miss of its own
    // (~11 k cycles per scalar read, 70 of them in the sha256 shape).  The bytes feed a value that is only stored under a
    // condition that never holds, so the loads stay.
    uint32_t warm = 0;
    if (!short_proof) {
#pragma unroll 1
        for (uint32_t o = 64 * sub; o < plan.proof_len; o += 64 * L) warm += proof[o];
    }
    Transcript tr;
    tr_init(tr);
    Fr a, b, r;
    // Record fetch: one 8-byte load per lane and bundle (the L lanes of a proof read L consecutive records), issued one
    // bundle ahead so that its latency - which a lone wave cannot hide behind anything else - overlaps the current bundle.
    // END is the last bundle (the loader checks), so the trip count is known without looking at the records.
    const uint64_t *code = reinterpret_cast<const uint64_t *>(plan.instr);
    uint64_t nxt = code[sub];
    for (uint32_t pc = 0; pc + L < plan.n_instr; pc += L) {
        const uint64_t rec = nxt;
        nxt = code[pc + L + sub];
        const struct { uint32_t x, y; } raw = {(uint32_t)rec, (uint32_t)(rec >> 32)};
        H2vInstr ins;
        ins.op = (uint8_t)(raw.x & 0xffu); ins.pad = 0; ins.dst = (uint16_t)(raw.x >> 16);
        ins.a = (uint16_t)(raw.y & 0xffffu); ins.b = (uint16_t)(raw.y >> 16);
        switch (ins.op) {
        case H2V_OP_ABSORB_REG: {
            rf.load(a, ins.a);
            fr_from_mont(b, a);
            tr_absorb_scalar(tr, sbuf, lane, b);
        } break;
        case H2V_OP_ABSORB_CI: {
            // (all the bytes are loaded before the first is hashed: a lone wave pays a full memory latency for every
            // load it waits on, and a load inside the byte loop is waited on 48 times)
            uint8_t raw[48];
#pragma unroll
            for (int k = 0; k < 48; k++) raw[k] = committed[(size_t)ii * 48 + k];
            tr_put(tr, sbuf, lane, 1);
#pragma unroll
            for (int k = 0; k < 48; k++) tr_put(tr, sbuf, lane, raw[k]);
        } break;
        case H2V_OP_LOAD_INSTANCE: {
            const uint8_t *p = instances + ((size_t)ii * plan.n_pi + ins.a) * 32;
#pragma unroll
            for (int l = 0; l < 8; l++)
                b.v[l] = (uint32_t)p[4 * l] | ((uint32_t)p[4 * l + 1] << 8) | ((uint32_t)p[4 * l + 2] << 16) | ((uint32_t)p[4 * l + 3] << 24);
            // public inputs are field elements on the reference side (Rust F / Aiken State<Scalar>): only the canonical
            // encoding exists there, so v + r is rejected here rather than verifying as a second encoding of v
            if (FrF::geq_mod(b.v)) st |= H2V_ST_BAD_SCALAR;
            fr_to_mont(r, b);
            if (live) rf.store(ins.dst, r);
        } break;
        case H2V_OP_READ_POINT: {
            const uint32_t off = (uint32_t)ins.a | ((uint32_t)ins.b << 16);
            // unconditional loads from one base (a short proof reads a table of the plan instead and is zeroed): a
            // load under a branch per byte is waited for before the next one issues - 560 cycles per byte on a lone wave
            const uint8_t *src = short_proof ? reinterpret_cast<const uint8_t *>(plan.lines_sg2) : proof + off;
            uint8_t raw[48];
#pragma unroll
            for (int k = 0; k < 48; k++) raw[k] = src[k];
#pragma unroll
            for (int k = 0; k < 48; k++) raw[k] = short_proof ? (uint8_t)0 : raw[k];
            tr_put(tr, sbuf, lane, 1);
#pragma unroll
            for (int k = 0; k < 48; k++) tr_put(tr, sbuf, lane, raw[k]);
        } break;
        case H2V_OP_READ_SCALAR: {
            const uint32_t off = (uint32_t)ins.a | ((uint32_t)ins.b << 16);
            const uint8_t *src = short_proof ? reinterpret_cast<const uint8_t *>(plan.lines_sg2) : proof + off;
            uint8_t raw[32];
#pragma unroll
            for (int k = 0; k < 32; k++) raw[k] = src[k];
#pragma unroll
            for (int k = 0; k < 32; k++) raw[k] = short_proof ? (uint8_t)0 : raw[k];
#pragma unroll
            for (int l = 0; l < 8; l++)
                b.v[l] = (uint32_t)raw[4 * l] | ((uint32_t)raw[4 * l + 1] << 8) | ((uint32_t)raw[4 * l + 2] << 16) | ((uint32_t)raw[4 * l + 3] << 24);
            tr_put(tr, sbuf, lane, 1);
#pragma unroll
            for (int k = 0; k < 32; k++) tr_put(tr, sbuf, lane, raw[k]);
            // canonical encodings only: the Rust reader (and Plinth's mkScalar, BlsTypes.hs:129-132) rejects >= r
            if (FrF::geq_mod(b.v)) st |= H2V_ST_BAD_SCALAR;
            fr_to_mont(r, b);
            if (live) rf.store(ins.dst, r);
        } break;
        case H2V_OP_SQUEEZE: {
            // adjusted_types/mod.rs:44-71: update(0x00); h = finalize; h2 = blake2b256(h);
            // challenge = from_uniform_bytes(h || h2) = LE(h) + LE(h2) * 2^256 mod r   (transcript.ak:85-106)
            tr_put(tr, sbuf, lane, 0);
            uint64_t h[4], h2[4];
            tr_digest(tr, sbuf, lane, h);
            b2_hash32(h, h2, sbuf, lane);
            Fr lo, hi, k;
#pragma unroll
            for (int l = 0; l < 4; l++) {
                lo.v[2 * l] = (uint32_t)h[l]; lo.v[2 * l + 1] = (uint32_t)(h[l] >> 32);
                hi.v[2 * l] = (uint32_t)h2[l]; hi.v[2 * l + 1] = (uint32_t)(h2[l] >> 32);
            }
            fr_to_mont(a, lo);
            fr_const(k, FR_R3);
            fr_mul(b, hi, k);  // hi * 2^768 / 2^256 = (hi * 2^256) * R
            fr_add(r, a, b);
            if (live) rf.store(ins.dst, r);
        } break;
        case H2V_OP_CONST: {
            fr_const(r, plan.consts + (size_t)ins.a * 8);
            if (live) rf.store(ins.dst, r);
        } break;
        case H2V_OP_ADD: case H2V_OP_SUB: case H2V_OP_MUL: {
            rf.load(a, ins.a);
            rf.load(b, ins.b);
            if (ins.op == H2V_OP_ADD) fr_add(r, a, b);
            else if (ins.op == H2V_OP_SUB) fr_sub(r, a, b);
            else fr_mul(r, a, b);
            if (live) rf.store(ins.dst, r);
        } break;
        case H2V_OP_NEG: {
            rf.load(a, ins.a);
            FrF::neg(r, a);
            if (live) rf.store(ins.dst, r);
        } break;
        case H2V_OP_INV: {
            rf.load(a, ins.a);
            // recip_eea of zero divides by zero in the reference (bls_utils.ak:151-154) => reject
            {
                Fr ia = a, ir;   // copies: fr_inv takes references, and a / r must not be pinned in private memory
                if (!fr_inv(ir, ia)) st |= H2V_ST_INVERSE_OF_ZERO;
                r = ir;
            }
            if (live) rf.store(ins.dst, r);
        } break;
        case H2V_OP_ASSERT_ZERO: {
            rf.load(a, ins.a);
            if (!FrF::is_zero(a)) st |= H2V_ST_RECURSION;   // expect transcript_rep == i_1 (emitters/aiken.rs:696)
        } break;
        case H2V_OP_OUT_SCALAR: {
            rf.load(a, ins.a);
            fr_from_mont(r, a);
            if (live) {
#pragma unroll
                for (int l = 0; l < 8; l++) scalars[((size_t)i * plan.n_terms + ins.dst) * 8 + l] = r.v[l];
            }
        } break;
        default: break;   // NOP: this lane idles through the bundle
        }
    }
    if (L > 1) {   // status bits of the proof's other lanes (inverse of zero, recursion check)
        st_red[lane] = st;
        __syncthreads();
        for (uint32_t k = 1; k < L; k++) st |= st_red[((uint32_t)lane & (P - 1)) + k * P];
    }
    if (live && sub == 0) {
        if (warm == 0xffffffffu) st |= 0x80000000u;   // never true (at most 2^24 / 64 bytes of 255 are summed): keeps `warm` live
        status[i] = st;
        if (trace) {
            for (uint32_t k = 0; k < plan.n_trace; k++) {
                rf.load(a, plan.trace[2 * k + 1]);
                fr_from_mont(r, a);
#pragma unroll
                for (int l = 0; l < 8; l++) trace[((size_t)i * plan.n_trace + k) * 8 + l] = r.v[l];
            }
        }
    }
}

extern "C" __global__ void __launch_bounds__(64)
k_transcript_combiner(H2vDevPlan plan, uint32_t n, uint32_t stride, const uint8_t *__restrict__ proofs,
                      const uint64_t *__restrict__ proof_off, const uint8_t *__restrict__ instances,
                      const uint8_t *__restrict__ committed, uint32_t *__restrict__ regs,
                      uint32_t *__restrict__ scalars, uint32_t *__restrict__ status, uint32_t *__restrict__ trace) {
    __shared__ uint32_t sbuf[32 * 64];
    const int lane = threadIdx.x;
    const uint32_t i = blockIdx.x * 64 + lane;
    const bool live = i < n;
    const uint32_t ii = live ? i : n - 1;  // dead lanes shadow the last proof (keeps control flow uniform), never write
    const RegsGlobal rf = {regs, stride, ii, i};
    vm_run(plan, rf, sbuf, lane, i, ii, live, 1u, 0u, nullptr, 64u, proofs, proof_off, instances, committed, scalars, status, trace);
}
// P proofs per block (power of two, <= 64), register file in dynamic LDS: P * n_regs * 32 bytes
extern "C" __global__ void __launch_bounds__(64)
k_transcript_combiner_lds(H2vDevPlan plan, uint32_t n, uint32_t P, const uint8_t *__restrict__ proofs,
                          const uint64_t *__restrict__ proof_off, const uint8_t *__restrict__ instances,
                          const uint8_t *__restrict__ committed, uint32_t *__restrict__ scalars,
                          uint32_t *__restrict__ status, uint32_t *__restrict__ trace) {
    __shared__ uint32_t sbuf[32 * 64];
    __shared__ uint32_t st_red[64];
    const int lane = threadIdx.x;
    const uint32_t L = plan.vm_lanes;
    const uint32_t q = (uint32_t)lane & (P - 1);            // proof slot of the block
    const uint32_t sub = L > 1 ? ((uint32_t)lane / P) % L : 0u;   // which of the proof's lanes
    const uint32_t i = blockIdx.x * P + q;
    // P * L may be below 64 (fewer proofs per block = less LDS per block = more blocks resident per CU, chosen by the
    // launcher): lanes >= P * L shadow a lane of the same proof (same work, same values) and never write
    const bool live = (uint32_t)lane < P * L && i < n;
    const uint32_t ii = i < n ? i : n - 1;
    // a dead slot of the last block shadows proof n-1 but must not touch a live slot's registers: it owns slot q anyway
    const RegsLds rf = {P, q};
    vm_run(plan, rf, sbuf, lane, i, ii, live, L, sub, st_red, P, proofs, proof_off, instances, committed, scalars, status, trace);
}

// ============================================================================ K2: G1 decompression
// zcash compressed encoding (bls_utils.ak:17-49, CompressUncompress.hs:53-100): bit7 compressed, bit6 infinity,
// bit5 "y is the lexicographically larger root"; y = (x^3+4)^((p+1)/4).
// Slot n_points is the committed instance when the circuit has one.
// out: affine Montgomery x||y (24 dwords, (0,0) = infinity); valid[...] = 1 iff the encoding is a point of G1.
//
// TWO waves per 64 points (block = 128 threads, the role is wave-uniform), because the square root and the subgroup
// test are the two long serial chains and the second does not need y:
//   wave 0: y = c^((p+1)/4) with c = x^3 + 4, y^2 == c, sign selection; then the point's MSM window tables;
//   wave 1: r-torsion test sigma(P) == [-x^2]P carried to the isomorphic curve E': Y^2 = X^3 + 4c^3 through
//           (x, y) -> (y^2 x, y^3 y) = (c x, c^2), which is known without the root.  The a = 0 group law does not
//           involve b, and the isomorphism commutes with sigma (X -> beta X), so the test on E' is the test on E for
//           either root y; when c is a non-residue wave 0 rejects and this result is ignored.  c == 0 means y == 0,
//           a 2-torsion point: rejected.
// Recursion (IVC) adds two slots per proof whose x and sign(y) are rebuilt from public inputs
// (g1_from_coords, aiken_halo2/lib/bls_utils.ak:32-49; limb packing emitters/aiken.rs:705-741):
//     coordinate = (1 + i_hi * 2^224 + i_lo) mod p ,  point = decompress(x with the parity flag of y).
#define H2V_DEC_PTS 64
H2V_DN void acc_coordinate(Fp &r, const uint8_t *ins, uint32_t idx_hi, uint32_t idx_lo) {
    Fp part[2];
#pragma unroll 1
    for (int q = 0; q < 2; q++) {
        const uint8_t *p = ins + (size_t)(q == 0 ? idx_hi : idx_lo) * 32;
        Fr b, m;
#pragma unroll
        for (int l = 0; l < 8; l++)
            b.v[l] = (uint32_t)p[4 * l] | ((uint32_t)p[4 * l + 1] << 8) | ((uint32_t)p[4 * l + 2] << 16) | ((uint32_t)p[4 * l + 3] << 24);
        FrF::to_mont(m, b);       // public inputs are field elements: reduced mod r exactly as LOAD_INSTANCE does
        FrF::from_mont(b, m);
        Fp plain;
#pragma unroll
        for (int l = 0; l < 12; l++) plain.v[l] = l < 8 ? b.v[l] : 0u;
        fp_to_mont(part[q], plain);
    }
    Fp c, one;
#pragma unroll
    for (int l = 0; l < 12; l++) c.v[l] = FP_TWO224[l];
    fp_set_one(one);
    fp_mul(r, part[0], c);
    fp_add(r, r, part[1]);
    fp_add(r, r, one);
}
// one group of 64 points (one wave): `role` 0 = square root (+ window tables), 1 = subgroup test
H2V_DI void dec_group(const H2vDevPlan &plan, uint32_t n, const uint8_t *__restrict__ proofs, const uint64_t *__restrict__ proof_off,
                      const uint8_t *__restrict__ committed, const uint8_t *__restrict__ instances,
                      uint32_t *__restrict__ pts, uint8_t *__restrict__ valid, uint32_t *__restrict__ pt_tab /* or NULL */,
                      const uint32_t mode, uint8_t *__restrict__ valid_sub, const uint32_t group, const uint32_t role,
                      const uint32_t lane, uint8_t *sub_ok) {
    // mode 0: both roles in one 128-thread block (valid = on curve && in G1);  modes 1 / 2: one role per wave, so that
    // the two chains can be scheduled apart: 1 = square root + tables (valid = encoding / on curve), 2 = subgroup test
    // (valid_sub)
    const uint32_t slots = H2V_SLOTS(plan);
    const uint32_t gid = group * H2V_DEC_PTS + lane;
    const bool live = gid < n * slots;
    const uint32_t gg = live ? gid : 0;
    const uint32_t i = gg / slots, j = gg - i * slots;
    bool ok = live;
    const uint8_t *src = nullptr;
    const bool is_acc = j >= plan.n_points + plan.n_ci;
    if (j < plan.n_points) {
        const uint64_t off0 = proof_off[i];
        if (proof_off[i + 1] - off0 < plan.proof_len) ok = false;  // short proof: nothing to read
        else src = proofs + off0 + plan.points[j];
    } else if (!is_acc) {
        src = committed + (size_t)i * 48;
    }
    uint32_t w[12];  // big-endian bytes -> little-endian limbs
#pragma unroll
    for (int k = 0; k < 12; k++) w[k] = 0;
    Fp acc_x;
    fp_set_zero(acc_x);
    if (is_acc) {
        // accumulator point: x (already a field element) and the parity of y; never the infinity encoding
        const uint32_t s4 = 4 * (j - plan.n_points - plan.n_ci);
        const uint8_t *ins = instances + (size_t)i * plan.n_pi * 32;
        Fp cy;
        acc_coordinate(acc_x, ins, plan.acc_idx[s4 + 0], plan.acc_idx[s4 + 1]);
        acc_coordinate(cy, ins, plan.acc_idx[s4 + 2], plan.acc_idx[s4 + 3]);
        w[11] = (4u | (fp_is_lex_larger(cy) ? 1u : 0u)) << 29;   // flags only; the limbs below stay unused
    } else if (ok) {
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const uint8_t *p = src + 44 - 4 * k;
            w[k] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
        }
    }
    const uint32_t flags = w[11] >> 29;
    w[11] &= 0x1fffffffu;
    G1A out;
    g1a_set_inf(out);
    ok = ok && (flags & 4) != 0;
    bool finite = false;
    if (ok && (flags & 2)) {
        // infinity: all other bits must be clear
        uint32_t any = flags & 1;
#pragma unroll
        for (int k = 0; k < 12; k++) any |= w[k];
        ok = any == 0;
    } else if (ok) {
        ok = is_acc || !FpF::geq_mod(w);
        finite = ok;
    }
    if (finite) {
        Fp x, c, t, b4;
#pragma unroll
        for (int k = 0; k < 12; k++) { t.v[k] = w[k]; b4.v[k] = FP_B[k]; }
        fp_to_mont(x, t);
        if (is_acc) x = acc_x;
        fp_sqr(c, x); fp_mul(c, c, x); fp_add(c, c, b4);
        if (role == 0) {
            Fp y, chk;
            {
                F28 c28, y28;
                f28_from_fp(c28, c);
                f28_sqrt_chain(y28, c28);
                f28_to_fp(y, y28);
            }
            fp_sqr(chk, y);
            ok = fp_eq(chk, c);
            if (ok) {
                if (fp_is_lex_larger(y) != ((flags & 1) != 0)) fp_neg(y, y);
                out.x = x; out.y = y;
                // this wave is done at ~55 % of the subgroup wave's chain: it spends the slack on the MSM's window
                // tables of the point ([1..8]P and [1..8]phi(P), affine), which depend on the point alone
                if (pt_tab) g1_build_window_tables_glv(pt_tab + (size_t)gid * 448, out);
            }
        } else {
            ok = !fp_is_zero(c);
            if (ok) {
                G1A q;
                fp_mul(q.x, c, x);
                fp_sqr(q.y, c);
                ok = g1a_in_subgroup28(q);
            }
        }
    }
    if (mode == 2) {
        if (live) valid_sub[gid] = ok ? 1 : 0;
        return;
    }
    if (mode == 0) {
        if (role == 1) sub_ok[lane] = ok ? 1 : 0;
        __syncthreads();
    }
    if (role == 0 && live) {
        if (mode == 0) ok = ok && sub_ok[lane] != 0;
        if (!ok) g1a_set_inf(out);
#pragma unroll
        for (int k = 0; k < 12; k++) { pts[(size_t)gid * 24 + k] = out.x.v[k]; pts[(size_t)gid * 24 + 12 + k] = out.y.v[k]; }
        valid[gid] = ok ? 1 : 0;
    }
}

extern "C" __global__ void __launch_bounds__(128, 2)
k_g1_decompress(H2vDevPlan plan, uint32_t n, const uint8_t *__restrict__ proofs, const uint64_t *__restrict__ proof_off,
                const uint8_t *__restrict__ committed, const uint8_t *__restrict__ instances,
                uint32_t *__restrict__ pts, uint8_t *__restrict__ valid, uint32_t *__restrict__ pt_tab /* or NULL */,
                uint32_t mode, uint8_t *__restrict__ valid_sub) {
    __shared__ uint8_t sub_ok[H2V_DEC_PTS];
    const uint32_t role = mode == 0 ? threadIdx.x >> 6 : mode - 1;
    dec_group(plan, n, proofs, proof_off, committed, instances, pts, valid, pt_tab, mode, valid_sub, blockIdx.x, role, threadIdx.x & 63, sub_ok);
}
// The same work from a queue: the launch has at most one wave per SIMD (256-thread blocks: four waves, one per SIMD of
// a CU) and every wave takes the next 64-point unit until none is left - the n_groups subgroup tests first (the longer
// chain), then the n_groups square roots.  Separate launches left the pairing of waves on SIMDs to the dispatcher: some
// subgroup waves shared a SIMD with a root wave for their whole life and finished at 1.45 ms while most SIMDs idled
// from 1.06 ms on.  `counter` is zeroed by the host before the launch; a wave leaves when the counter passes the end.
extern "C" __global__ void __launch_bounds__(256, 2)
k_g1_decompress_queue(H2vDevPlan plan, uint32_t n, const uint8_t *__restrict__ proofs, const uint64_t *__restrict__ proof_off,
                      const uint8_t *__restrict__ committed, const uint8_t *__restrict__ instances,
                      uint32_t *__restrict__ pts, uint8_t *__restrict__ valid, uint32_t *__restrict__ pt_tab,
                      uint8_t *__restrict__ valid_sub, uint32_t *__restrict__ counter, uint32_t n_groups) {
    const uint32_t lane = threadIdx.x & 63;
    for (;;) {
        uint32_t u = 0;
        if (lane == 0) u = atomicAdd(counter, 1u);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= 2 * n_groups) break;
        if (u < n_groups) dec_group(plan, n, proofs, proof_off, committed, instances, pts, valid, pt_tab, 2u, valid_sub, u, 1u, lane, nullptr);
        else dec_group(plan, n, proofs, proof_off, committed, instances, pts, valid, pt_tab, 1u, valid_sub, u - n_groups, 0u, lane, nullptr);
    }
}

// ============================================================================ K4: per-proof G1 MSM
// er = sum_t s_t * B_t with T = n_terms (16 ... ~60) 255-bit scalars per proof.  T is far too small for bucket
// (Pippenger) accumulation to pay - 2^c buckets per window would outnumber the terms - so the mapping is:
//   * TWO lanes per (proof, term): the GLV split k = k1 + k2*lambda gives lane 0 the pair (k1, P) and lane 1
//     (k2, phi(P) = (beta' x, y)), both scalars below 2^128 (half the doubling chain, twice the waves);
//   * per lane a signed 4-bit window ladder on the lazily reduced field: table [1..8]*P, affine, in a per-lane slab
//     of the workspace; 33 windows of 4 doublings + one mixed addition (digits in [-8, 8]);
//   * a segmented tree reduction of the 2*T partial sums of each (proof, term group) through LDS.
// Bytes per term: 32 (scalar) + 96 (affine base) in, 144 per proof out (Jacobian).
// One launch sums a RANGE of a term table (the proof's own MSM; with recursion also acc_left, acc_right + fixed bases,
// and the two folds el + c*acc_left, er + c*acc_right over the fold's own point / scalar buffers):
struct H2vMsmArgs {
    const uint32_t *terms;     // (kind, index) pairs; kind = VK base or per-proof slot of `pts`
    uint32_t term_base;        // first term of the range
    uint32_t n_terms;          // terms in the range (2 * n_terms lanes per proof, <= 512)
    uint32_t scal_stride;      // scalars per proof in `scalars`
    uint32_t scal_col_base;    // column of the range's first term
    uint32_t slots;            // point slots per proof in `pts`
    // the range is cut into up to three consecutive groups, each summed into its own output (recursion: the proof's
    // MSM, acc_left and acc_right + fixed bases in ONE launch): group g = terms [grp_end[g-1], grp_end[g])
    uint32_t grp_end[3];
    uint32_t *out[3];          // n x 36 dwords each (Jacobian); unused groups: grp_end == n_terms
    // window tables built ahead of the MSM (2 x 224 dwords per point: P and phi(P)): per-proof slots by the
    // decompression kernel, VK bases at plan load.  NULL: every lane builds its own table into `tabws` (fold MSMs).
    const uint32_t *pt_tab;    // [proof][slot][2][224]
    const uint32_t *vk_tab;    // [base][2][224]
    // fixed-base launches (k_g1_msm_fixed; one group, every term of the range a VK base): lane f of a proof's n_fixl
    // lanes sums the fix_k terms f * fix_k ... of the range from the all-window tables fix_tab [base][65 windows][8][28]
    const uint32_t *fix_tab;
    uint32_t fix_k, n_fixl;
    // RLC batch mode: the per-proof MSM queued behind the batch check (k_g1_msm*_cond) returns at once when *skip != 0
    const uint32_t *skip;
};
// A proof owns exactly LPT * n_terms consecutive lanes of a block (no power-of-two padding: 34 terms used to occupy
// 128 lanes); the block holds as many whole proofs as fit, the rest of its lanes idle.
//
// LPT = lanes per term.  2: one lane per GLV half (128 doublings + 33 additions per lane).
// 8: a QUAD per GLV half (h2v_curve28.hpp: quad-cooperative arithmetic): the four lanes hold the same accumulator and
// share the multiplications of every doubling (depth 3 instead of 7) and addition (6 instead of 11): 33 x (4 x 3 + 6) =
// 594 multiplications deep instead of 1287, on four times the lanes - for launches that leave most SIMDs idle anyway.
// Measured, it buys far less than the depth suggests: T = 16 x 64 proofs 1.35 -> 1.17 ms; T = 58 (464 lanes per proof: a
// 512-thread block, two waves per SIMD) 1.35 -> 1.69 ms; a quad per TERM (both halves, 780 deep, 256-thread blocks) 1.45-1.51
// ms, slower than two plain lanes per term.  A lone wave is bound by the latency of its dependent multiply-adds, and the one-lane
// formulas already overlap their independent multiplications (A = X^2, B = Y^2, ...) inside the lane; a level of the quad
// schedule is exactly one multiplication between two barriers (select, multiply, broadcast).  Kept as a forced shape
// (H2V_MSM_LPT=8, parity-tested), never chosen by the launcher: see msm_ladder_shape.
// 1: one lane runs both halves of its term on ONE accumulator, so the 128 doublings are shared (128 doublings + 66
// additions per lane, 36 % less work per proof, half the waves).  Measured on MI355X one wave per SIMD of this code
// already reaches 85 % of what two deliver (2048 proofs: 1.54 ms, 4096: 2.60 ms with LPT = 2), so whenever the batch
// still fills the SIMDs with one lane per term the smaller total wins; the launcher picks (h2v_capi.hip).
// FIX (with LPT = 1): a fixed-base launch over a range of VK-base terms.  The VK bases are the same for every proof, so
// ALL their window multiples [d] 16^w V (w = 0..64, d = 1..8) are precomputed at plan load and a VK term costs 65 mixed
// additions and no doubling; a lane sums fix_k of them (one accumulator per base - additions inside one base's sum
// cannot be exceptional, see below - merged with complete additions).  It runs beside the ladder launch of the
// per-proof terms (separate waves: one wave mixing the two kinds would execute them one after the other) and leaves
// fewer lanes per proof: for T >= 34 at 2048 proofs that brings the MSM back to one wave per SIMD.
template <int LPT, bool FIX = false, bool MADD_INL = false>
H2V_DI void msm_body(const H2vDevPlan &plan, const H2vMsmArgs &ma, uint32_t n, uint32_t per_block,
                     const uint32_t *__restrict__ scalars, const uint32_t *__restrict__ pts, uint32_t *__restrict__ tabws,
                     uint32_t *red /* Jacobian partial sums, dword d of thread t at red[d*blockDim.x + t] */,
                     const uint32_t bid /* logical block: blockIdx.x, or the loop index of a conditional launch */) {
    static_assert(!FIX || LPT == 1, "fixed-base mode runs merged ladders");
    static_assert(LPT == 1 || LPT == 2 || LPT == 8, "one lane per term, one per GLV half, or a quad per GLV half");
    constexpr bool QUAD = LPT >= 4;                 // four lanes share one accumulator
    constexpr bool SPLIT = LPT == 2 || LPT == 8;    // one GLV half per lane (or quad); otherwise both halves on one accumulator
    constexpr int NH = SPLIT ? 1 : 2;               // GLV halves per accumulator
    const uint32_t tid = threadIdx.x, bs = blockDim.x;
    const uint32_t lanes_per_proof = FIX ? ma.n_fixl : LPT * ma.n_terms;
    const uint32_t seg = tid / lanes_per_proof;       // which of the block's proofs
    const uint32_t sub = tid - seg * lanes_per_proof; // position inside the proof's segment
    const bool fix_lane = FIX;   // a fixed-base launch has no ladder lanes (a wave mixing the two kinds would run them one after the other)
    const uint32_t term = FIX ? 0u : (LPT == 8 ? sub >> 3 : LPT == 2 ? sub >> 1 : sub), half = LPT == 8 ? (sub >> 2) & 1 : LPT == 2 ? sub & 1 : 0;
    const uint32_t i = bid * per_block + seg;
    const bool active = seg < per_block && i < n;
    // this lane's group: position and length of its reduction segment inside the proof's lanes
    const uint32_t grp = FIX ? 0u : (term < ma.grp_end[0] ? 0u : (term < ma.grp_end[1] ? 1u : 2u));
    const uint32_t g_lo = grp == 0 ? 0u : ma.grp_end[grp - 1];
    const uint32_t gsub = FIX ? sub : sub - LPT * g_lo, glen = FIX ? lanes_per_proof : LPT * (ma.grp_end[grp] - g_lo);
    // this lane's partial sum, on the lazily reduced field with an explicit infinity flag (h2v_curve28.hpp); idle
    // lanes and skipped terms contribute the point at infinity
    G1J28 lad;
    bool lad_inf = true;
    if (FIX && active && fix_lane) {
        // The accumulator of one base is [a]V with a = the signed-digit prefix read so far (most significant window first),
        // a multiple of B^(w+1), B = 2^c; adding [d B^w]V is exceptional iff a -+ d B^w = 0 mod r.  As integers |a -+ d B^w| <
        // 2^265, so that means a -+ d B^w = m r with |m| small: 0 is excluded by the digits' size (|d| <= B/2 < B) unless
        // a = d = 0, and m r is not = -+d B^w modulo B^(w+1) for w > 0 (r is odd: m r = 0 mod B^w needs B^w | m).  So the
        // additions of the windows w > 0 are unchecked.  For w = 0 the sum a + d_0 is the whole scalar (canonical, not 0: never
        // the inverse), but the DOUBLING case a = d_0 mod r is reachable: r = 1 mod 2^32, so the scalar r - 2 recodes to
        // d_0 = -1 with prefix a = r - 1, i.e. acc = -V and the entry added is -V (the one such scalar, for c = 4, 8 and 12 -
        // ADVICE r3).  The last window of a base therefore goes through the complete law (one addition in 22 .. 65); sums of
        // different bases meet in complete ones as well.
        const uint32_t f0 = sub * ma.fix_k;
#pragma unroll 1
        for (uint32_t j = 0; j < ma.fix_k; j++) {
            if (f0 + j >= ma.n_terms) break;
            const uint32_t ft = f0 + j;
            const uint32_t idx = ma.terms[2 * (ma.term_base + ft) + 1];
            const uint32_t *bp = plan.vk_bases + (size_t)idx * 24;
            uint32_t any_b = 0, sc[8], any_s = 0;
#pragma unroll
            for (int k = 0; k < 24; k++) any_b |= bp[k];
            const uint32_t *sp = scalars + ((size_t)i * ma.scal_stride + ma.scal_col_base + ft) * 8;
#pragma unroll
            for (int k = 0; k < 8; k++) { sc[k] = sp[k]; any_s |= sc[k]; }
            if (any_b == 0 || any_s == 0) continue;
            // signed c-bit digits, least significant first (c = plan.fix_c: 4, 8 or 12; W windows; the last window takes the
            // carry: for c = 4 / 8 it is a window of its own, for c = 12 the top window holds 4 bits and has the room)
            const uint32_t c = plan.fix_c, W = plan.fix_W, E = 1u << (c - 1), mask = (1u << c) - 1u;
            int16_t dg[65];
            uint32_t carry = 0;
#pragma unroll 1
            for (uint32_t q = 0; q < W; q++) {
                const uint32_t bit = q * c, word = bit >> 5, sh = bit & 31;
                uint64_t two = 0;
                if (word < 8) two = (uint64_t)sc[word] | (word + 1 < 8 ? (uint64_t)sc[word + 1] << 32 : 0ull);
                uint32_t d = ((uint32_t)(two >> sh) & mask) + carry;
                carry = d > E ? 1u : 0u;
                dg[q] = (int16_t)(carry ? (int)d - (int)(mask + 1u) : (int)d);
            }
            G1J28 acc;
            bool acc_inf = true;
            const uint32_t *tabb = ma.fix_tab + (size_t)idx * W * E * 28;
#pragma unroll 1
            for (int q = (int)W - 1; q >= 0; q--) {
                const int d = dg[q];
                if (d == 0) continue;
                const uint32_t *ent = tabb + ((size_t)q * E + (uint32_t)((d < 0 ? -d : d) - 1)) * 28;
                F28 qx, qy;
#pragma unroll
                for (int k = 0; k < 14; k++) { qx.l[k] = ent[k]; qy.l[k] = ent[14 + k]; }
                if (acc_inf) {
                    acc.x = qx;
                    acc.y = qy;
                    if (d < 0) { F28_NEG(acc.y, qy, 3, 1); f28_carry(acc.y); }
                    f28_set_one(acc.z);
                    acc_inf = false;
                } else if (q == 0) {
                    G1J28 qq;                                        // w = 0: complete (see above)
                    qq.x = qx; qq.y = qy;
                    f28_set_one(qq.z);
                    g1j28_acc_add(acc, acc_inf, qq, d < 0);
                } else {
                    g1j28_madd_ladder(acc, acc, qx, qy, d < 0);
                }
            }
            if (!acc_inf) g1j28_acc_add(lad, lad_inf, acc, false);   // complete
        }
    }
    if (active && !fix_lane) {
        // terms[] as uploaded by h2v_plan_load: kind is VK base (1) or per-proof slot (0); the committed instance has
        // been rewritten to slot n_points there.  Two-way integer selects only: a nested three-way pointer select was
        // miscompiled by ROCm 7.2 (the copy of i feeding the scalar address was left undefined on the third path).
        const uint32_t kind = ma.terms[2 * (ma.term_base + term)], idx = ma.terms[2 * (ma.term_base + term) + 1];
        const uint32_t slots = ma.slots;
        const bool is_vk = kind == H2V_TERM_VK_BASE;
        const uint32_t *src = is_vk ? plan.vk_bases : pts;
        const size_t elem = is_vk ? (size_t)idx : (size_t)i * slots + idx;
        const uint32_t *bp = src + elem * 24;
        G1A base;
#pragma unroll
        for (int k = 0; k < 12; k++) { base.x.v[k] = bp[k]; base.y.v[k] = bp[12 + k]; }
        uint32_t s[8], k1[4], k2[4];
        const uint32_t *sp = scalars + ((size_t)i * ma.scal_stride + ma.scal_col_base + term) * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = sp[k];
        glv_split(k1, k2, s);
        uint32_t any = 0;
        if (SPLIT) {
#pragma unroll
            for (int k = 0; k < 4; k++) any |= half ? k2[k] : k1[k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) any |= k1[k] | k2[k];
        }
        if (!g1a_is_inf(base) && any != 0) {
            // signed 4-bit recoding, least significant digit first: digit = dg[h][q] in [-8, 8]
            int8_t dg[NH][33];
#pragma unroll
            for (int h = 0; h < NH; h++) {
                uint32_t kk[4];
#pragma unroll
                for (int k = 0; k < 4; k++) kk[k] = (SPLIT ? half != 0 : h != 0) ? k2[k] : k1[k];
                uint32_t carry = 0;
#pragma unroll 1
                for (int q = 0; q < 32; q++) {
                    uint32_t d = ((kk[q >> 3] >> (4 * (q & 7))) & 15u) + carry;
                    carry = d > 8 ? 1u : 0u;
                    dg[h][q] = (int8_t)(carry ? (int)d - 16 : (int)d);
                }
                dg[h][32] = (int8_t)carry;
            }
            // table[m-1] = m*P (half 0) or m*phi(P) (half 1), m = 1..8, AFFINE (x, y: 2 x 14 limbs), 112 contiguous
            // bytes per entry, the two halves 224 dwords apart.  Normally it was built ahead of this kernel -
            // per-proof points by the decompression kernel (its square-root wave has the slack), VK bases at plan
            // load - and is only read here; the fold MSMs of the recursion path build theirs on the spot in a per-lane
            // slab of the workspace.  The multiples are affine (one inversion through Montgomery's trick) so that the
            // window additions are mixed ones.
            const uint32_t *tab;   // of this lane's first half
            if (ma.pt_tab) {
                tab = (is_vk ? ma.vk_tab + (size_t)idx * 448 : ma.pt_tab + ((size_t)i * slots + idx) * 448) + half * 224;
            } else if (SPLIT) {   // (the quad shapes are only launched with prebuilt tables)
                uint32_t *mine = tabws + (((size_t)i * ma.n_terms + term) * 2 + half) * 224;
                if (half) {  // phi(P)
                    Fp beta;
#pragma unroll
                    for (int k = 0; k < 12; k++) beta.v[k] = FP_BETA_GLV[k];
                    fp_mul(base.x, base.x, beta);
                }
                g1_build_window_table(mine, base);
                tab = mine;
            } else {
                uint32_t *mine = tabws + ((size_t)i * ma.n_terms + term) * 448;
                g1_build_window_tables_glv(mine, base);
                tab = mine;
            }
            // The ladder runs on the lazily reduced 28-bit field (h2v_fp28.hpp / h2v_curve28.hpp) and is only ever
            // touched by inlined code, so `lad` stays in VGPRs.
            //
            // Exceptional additions.  The accumulator is [a]P + [b]phi(P) = [a + b lambda]P with (a, b) the signed
            // prefixes read so far (not both zero once a digit was non-zero: a signed-digit prefix with a non-zero
            // leading digit is non-zero).  Adding [d]P (or [d]phi(P)) is exceptional iff (a -+ d, b) (or (a, b -+ d))
            // lies in the lattice {(u, v): u + v lambda = 0 mod r}, whose non-zero vectors are longer than 2^126
            // (reduced basis (lambda, -1), (1, lambda + 1), lambda = x^2 - 1 > 2^127).  With one half per lane b = 0
            // and a < 2^129 < r: never.  With both halves on one lane the prefixes stay below 2^(4 (33 - q)) + 8 in
            // window q, so windows q >= 2 are safe and only the last two take the complete addition.
#pragma unroll 1
            for (int q = 32; q >= 0; q--) {
                if (q != 32 && !lad_inf) {
                    if (QUAD) g1j28_dbl_n_coop3(lad, 4);   // the quad's lanes share the multiplications of a doubling: depth 3 instead of 7
                    else {
#pragma unroll 1
                        for (int rep = 0; rep < 4; rep++) g1j28_dbl_t<true>(lad, lad);   // multiplier inlined: no argument marshalling
                    }
                }
#pragma unroll 1
                for (int h = 0; h < NH; h++) {
                    const int d = dg[h][q];
                    if (d == 0) continue;
                    const uint32_t *ent = tab + h * 224 + ((d < 0 ? -d : d) - 1) * 28;
                    F28 qx, qy;
#pragma unroll
                    for (int k = 0; k < 14; k++) { qx.l[k] = ent[k]; qy.l[k] = ent[14 + k]; }
                    if (lad_inf) {
                        lad.x = qx;
                        lad.y = qy;
                        if (d < 0) { F28_NEG(lad.y, qy, 3, 1); f28_carry(lad.y); }
                        f28_set_one(lad.z);
                        lad_inf = false;
                    } else if (QUAD && (SPLIT || q >= 2)) {
                        g1j28_madd_quad(lad, qx, qy, d < 0);                     // never exceptional (above); depth 6 instead of 11
                    } else if (SPLIT || q >= 2) {
                        g1j28_madd_ladder_t<MADD_INL>(lad, lad, qx, qy, d < 0);   // never an exceptional case (above)
                    } else {
                        G1J28 o;
                        o.x = qx; o.y = qy;
                        f28_set_one(o.z);
                        g1j28_acc_add(lad, lad_inf, o, d < 0);
                    }
                }
            }
        }
    }
    if (QUAD && (sub & 3) != 0) lad_inf = true;   // the quad's four lanes hold the same sum: its first lane carries it
    // segmented reduction over the lanes of each (proof, group), still on the lazy field: 42 limbs + the flag per lane
    // in LDS (dword d of thread t at red[d*bs + t]); only the lane that ends up with a group's sum converts it back
#define MSM_RED_STORE()                                                                     \
    do {                                                                                    \
        _Pragma("unroll") for (int k = 0; k < 14; k++) {                                    \
            red[k * bs + tid] = lad.x.l[k]; red[(14 + k) * bs + tid] = lad.y.l[k]; red[(28 + k) * bs + tid] = lad.z.l[k]; \
        }                                                                                   \
        red[42 * bs + tid] = lad_inf ? 1u : 0u;                                             \
    } while (0)
    MSM_RED_STORE();
    __syncthreads();
    uint32_t top = 1;
    while (top < lanes_per_proof) top <<= 1;
    for (uint32_t s = top >> 1; s >= 1; s >>= 1) {
        if (seg < per_block && gsub < s && gsub + s < glen) {
            if (red[42 * bs + tid + s] == 0) {
                G1J28 other;
#pragma unroll
                for (int k = 0; k < 14; k++) {
                    other.x.l[k] = red[k * bs + tid + s];
                    other.y.l[k] = red[(14 + k) * bs + tid + s];
                    other.z.l[k] = red[(28 + k) * bs + tid + s];
                }
                g1j28_acc_add(lad, lad_inf, other, false);   // complete: equal / opposite partial sums, infinity
                MSM_RED_STORE();
            }
        }
        __syncthreads();
    }
#undef MSM_RED_STORE
    if (gsub == 0 && seg < per_block && i < n) {
        G1J acc;
        g1j28_to_g1j(acc, lad, lad_inf);
        uint32_t *out = grp == 0 ? ma.out[0] : (grp == 1 ? ma.out[1] : ma.out[2]);
#pragma unroll
        for (int k = 0; k < 12; k++) {
            out[(size_t)i * 36 + k] = acc.x.v[k];
            out[(size_t)i * 36 + 12 + k] = acc.y.v[k];
            out[(size_t)i * 36 + 24 + k] = acc.z.v[k];
        }
    }
}
extern "C" __global__ void __launch_bounds__(512, 2)
k_g1_msm(H2vDevPlan plan, H2vMsmArgs ma, uint32_t n, uint32_t per_block /* proofs per block */,
         const uint32_t *__restrict__ scalars, const uint32_t *__restrict__ pts, uint32_t *__restrict__ tabws) {
    extern __shared__ uint32_t red[];
    msm_body<2, false, true>(plan, ma, n, per_block, scalars, pts, tabws, red, blockIdx.x);
}
extern "C" __global__ void __launch_bounds__(512, 2)
k_g1_msm_quad(H2vDevPlan plan, H2vMsmArgs ma, uint32_t n, uint32_t per_block /* proofs per block */,
              const uint32_t *__restrict__ scalars, const uint32_t *__restrict__ pts, uint32_t *__restrict__ tabws) {
    extern __shared__ uint32_t red[];
    msm_body<8, false, true>(plan, ma, n, per_block, scalars, pts, tabws, red, blockIdx.x);
}
extern "C" __global__ void __launch_bounds__(512, 2)
k_g1_msm_merged(H2vDevPlan plan, H2vMsmArgs ma, uint32_t n, uint32_t per_block /* proofs per block */,
                const uint32_t *__restrict__ scalars, const uint32_t *__restrict__ pts, uint32_t *__restrict__ tabws) {
    extern __shared__ uint32_t red[];
    msm_body<1, false, true>(plan, ma, n, per_block, scalars, pts, tabws, red, blockIdx.x);
}

// the same two ladders as the fall-back of the RLC batch mode: they return at once when the batch check passed.  (Separate
// entry points: the check costs the parity path's kernels nothing, not even a different register allocation.)  The grid
// is small and walks the logical blocks in a loop: a launch that only has to find out that it is not needed should not
// cost a thousand workgroup dispatches (40-50 us per skipped kernel on the RLC mode's critical path before).
// skip[1 + g] != 0: group g (proofs 64 g .. 64 g + 63) passed its own check after the batch check failed (k_pairing_rlc_groups)
// and needs no per-proof verdicts; a logical block is skipped when every group it touches passed
H2V_DI bool msm_groups_passed(const uint32_t *__restrict__ skip, uint32_t bid, uint32_t per_block, uint32_t n) {
    const uint32_t p0 = bid * per_block, p1 = (p0 + per_block < n ? p0 + per_block : n) - 1;
    bool all = true;
    for (uint32_t g = p0 >> 6; g <= (p1 >> 6); g++) all = all && skip[1 + g] != 0;
    return all;
}
extern "C" __global__ void __launch_bounds__(512, 2)
k_g1_msm_cond(H2vDevPlan plan, H2vMsmArgs ma, uint32_t n, uint32_t per_block, const uint32_t *__restrict__ scalars,
              const uint32_t *__restrict__ pts, uint32_t *__restrict__ tabws) {
    extern __shared__ uint32_t red[];
    if (ma.skip[0]) return;   // (uniform over the launch: before any barrier)
    const uint32_t n_blocks = (n + per_block - 1) / per_block;
    for (uint32_t bid = blockIdx.x; bid < n_blocks; bid += gridDim.x) {
        if (msm_groups_passed(ma.skip, bid, per_block, n)) continue;   // (uniform over the block)
        msm_body<2, false, true>(plan, ma, n, per_block, scalars, pts, tabws, red, bid);
        __syncthreads();
    }
}
extern "C" __global__ void __launch_bounds__(512, 2)
k_g1_msm_merged_cond(H2vDevPlan plan, H2vMsmArgs ma, uint32_t n, uint32_t per_block, const uint32_t *__restrict__ scalars,
                     const uint32_t *__restrict__ pts, uint32_t *__restrict__ tabws) {
    extern __shared__ uint32_t red[];
    if (ma.skip[0]) return;
    const uint32_t n_blocks = (n + per_block - 1) / per_block;
    for (uint32_t bid = blockIdx.x; bid < n_blocks; bid += gridDim.x) {
        if (msm_groups_passed(ma.skip, bid, per_block, n)) continue;   // (uniform over the block)
        msm_body<1, false, true>(plan, ma, n, per_block, scalars, pts, tabws, red, bid);
        __syncthreads();
    }
}

extern "C" __global__ void __launch_bounds__(512, 2)
k_g1_msm_fixed(H2vDevPlan plan, H2vMsmArgs ma, uint32_t n, uint32_t per_block /* proofs per block */,
               const uint32_t *__restrict__ scalars, const uint32_t *__restrict__ pts, uint32_t *__restrict__ tabws) {
    extern __shared__ uint32_t red[];
    msm_body<1, true>(plan, ma, n, per_block, scalars, pts, tabws, red, blockIdx.x);
}


// Several GLV halves per lane (H = 4 .. 8: two to four terms): the halves a lane carries share ONE accumulator and therefore the
// 128 doublings - per lane 128 doublings + 33 H mixed additions instead of one ladder per half (H = 4: 26 % fewer multiply-adds
// per proof, H = 8: 39 %), on 2 / H of the waves with a chain (128 x 7 + 33 H x 11) / 1622 as long.  It loses when the launch
// is alone on the chip (half or a quarter of the SIMDs get a wave) and wins when other kernels fill them: several steps in
// flight, where the step time is the instruction count.  Non-recursive plans with prebuilt tables.
// The proof's 2 T halves are dealt out H per lane in order (half hh = 2 term + glv half): lanes of one proof differ by at most
// the last one's load, and no lane waits through addition slots of halves it does not have - with whole terms per lane ten terms
// on four lanes were 3 + 3 + 3 + 1 and every wave ran six addition slots per window for five halves per lane on average.
// One accumulator over DIFFERENT points has no lattice argument against exceptional additions (crafted proofs can make
// P_2 = [m] P_1), so the ladder runs unchecked and is judged once at the end: an exceptional addition or doubling leaves
// Z = 0, which every later Z inherits; such a lane (never an honest one) redoes its ladder with the complete group law.
#define MSM_MAX_HALVES 8
template <bool COMPLETE>
H2V_DN void msm_multi_ladder(G1J28 &out, bool &out_inf, const int8_t (&dg)[MSM_MAX_HALVES][33], const uint32_t *const (&tab)[MSM_MAX_HALVES], const int H) {
    G1J28 lad;
    bool lad_inf = true;
    // The table entry of the NEXT addition slot is fetched one slot ahead (round 4): a slot's entry is a scattered 112-byte read
    // out of a 73 MB table whose address hangs on a digit in private memory - two dependent round trips (~2 k cycles) in front of
    // every 11-multiplication addition otherwise.  The first slot of the next window is fetched across the four doublings.
    F28 nx, ny;
    int nd;
    auto fetch = [&](const int q, const int h) {
        nd = dg[h][q];
        const int e = nd < 0 ? -nd : nd;
        const uint32_t *ent = tab[h] + (e ? e - 1 : 0) * 28;      // (a zero digit reads entry 0 and does not use it)
#pragma unroll
        for (int k = 0; k < 14; k++) { nx.l[k] = ent[k]; ny.l[k] = ent[14 + k]; }
    };
    fetch(32, 0);
#pragma unroll 1
    for (int q = 32; q >= 0; q--) {
        if (q != 32 && !lad_inf) {
#pragma unroll 1
            for (int rep = 0; rep < 4; rep++) g1j28_dbl_t<true>(lad, lad);
        }
#pragma unroll 1
        for (int h = 0; h < H; h++) {
            const F28 qx = nx, qy = ny;
            const int d = nd;
            if (h + 1 < H) fetch(q, h + 1);
            else if (q > 0) fetch(q - 1, 0);
            if (d == 0) continue;
            if (lad_inf) {
                lad.x = qx;
                lad.y = qy;
                if (d < 0) { F28_NEG(lad.y, qy, 3, 1); f28_carry(lad.y); }
                f28_set_one(lad.z);
                lad_inf = false;
            } else if (!COMPLETE) {
                g1j28_madd_ladder_t<true>(lad, lad, qx, qy, d < 0);
            } else {
                G1J28 o;
                o.x = qx; o.y = qy;
                f28_set_one(o.z);
                g1j28_acc_add(lad, lad_inf, o, d < 0);
            }
        }
    }
    out = lad;
    out_inf = lad_inf;
}
H2V_DI void msm_multi_body(const H2vDevPlan &plan, const H2vMsmArgs &ma, uint32_t n, uint32_t per_block, const uint32_t H,
                           const uint32_t *__restrict__ scalars, const uint32_t *__restrict__ pts, uint32_t *red) {
    const uint32_t tid = threadIdx.x, bs = blockDim.x;
    const uint32_t lanes_per_proof = (2 * ma.n_terms + H - 1) / H;
    const uint32_t seg = tid / lanes_per_proof, sub = tid - seg * lanes_per_proof;
    const uint32_t i = blockIdx.x * per_block + seg;
    const bool active = seg < per_block && i < n;
    G1J28 lad;
    bool lad_inf = true;
    if (active) {
        int8_t dg[MSM_MAX_HALVES][33];
        const uint32_t *tab[MSM_MAX_HALVES];
#pragma unroll 1
        for (uint32_t j = 0; j < H; j++) {
            const uint32_t hh = sub * H + j, term = hh >> 1, h = hh & 1;
            bool use = term < ma.n_terms;
            const uint32_t tt = use ? term : 0u;
            const uint32_t kind = ma.terms[2 * (ma.term_base + tt)], idx = ma.terms[2 * (ma.term_base + tt) + 1];
            const bool is_vk = kind == H2V_TERM_VK_BASE;
            const uint32_t *bp = (is_vk ? plan.vk_bases : pts) + (is_vk ? (size_t)idx : (size_t)i * ma.slots + idx) * 24;
            uint32_t any_b = 0, s[8], any_s = 0;
#pragma unroll
            for (int k = 0; k < 24; k++) any_b |= bp[k];
            const uint32_t *sp = scalars + ((size_t)i * ma.scal_stride + ma.scal_col_base + tt) * 8;
#pragma unroll
            for (int k = 0; k < 8; k++) { s[k] = sp[k]; any_s |= s[k]; }
            use = use && any_b != 0 && any_s != 0;       // the point at infinity / a zero scalar contribute nothing
            uint32_t k1[4], k2[4];
            glv_split(k1, k2, s);                        // (a term whose halves sit in two lanes is split in both)
            const uint32_t *t0 = is_vk ? ma.vk_tab + (size_t)idx * 448 : ma.pt_tab + ((size_t)i * ma.slots + idx) * 448;
            tab[j] = t0 + h * 224;
            uint32_t carry = 0;
#pragma unroll 1
            for (int q = 0; q < 32; q++) {
                const uint32_t kw = h ? k2[q >> 3] : k1[q >> 3];
                uint32_t d = ((kw >> (4 * (q & 7))) & 15u) + carry;
                carry = d > 8 ? 1u : 0u;
                dg[j][q] = use ? (int8_t)(carry ? (int)d - 16 : (int)d) : (int8_t)0;
            }
            dg[j][32] = use ? (int8_t)carry : (int8_t)0;
        }
        msm_multi_ladder<false>(lad, lad_inf, dg, tab, (int)H);
        if (!lad_inf) {
            Fp zc;
            F28 z = lad.z;
            f28_carry(z);
            f28_to_fp(zc, z);
            if (fp_is_zero(zc)) msm_multi_ladder<true>(lad, lad_inf, dg, tab, (int)H);   // crafted points only
        }
    }
    // reduction over the lanes of each proof (as in msm_body: lazy field, complete additions)
#define MSMM_RED_STORE()                                                                    \
    do {                                                                                    \
        _Pragma("unroll") for (int k = 0; k < 14; k++) {                                    \
            red[k * bs + tid] = lad.x.l[k]; red[(14 + k) * bs + tid] = lad.y.l[k]; red[(28 + k) * bs + tid] = lad.z.l[k]; \
        }                                                                                   \
        red[42 * bs + tid] = lad_inf ? 1u : 0u;                                             \
    } while (0)
    MSMM_RED_STORE();
    __syncthreads();
    uint32_t top = 1;
    while (top < lanes_per_proof) top <<= 1;
    for (uint32_t s = top >> 1; s >= 1; s >>= 1) {
        if (seg < per_block && sub < s && sub + s < lanes_per_proof) {
            if (red[42 * bs + tid + s] == 0) {
                G1J28 other;
#pragma unroll
                for (int k = 0; k < 14; k++) {
                    other.x.l[k] = red[k * bs + tid + s];
                    other.y.l[k] = red[(14 + k) * bs + tid + s];
                    other.z.l[k] = red[(28 + k) * bs + tid + s];
                }
                g1j28_acc_add(lad, lad_inf, other, false);
                MSMM_RED_STORE();
            }
        }
        __syncthreads();
    }
#undef MSMM_RED_STORE
    if (sub == 0 && seg < per_block && i < n) {
        G1J acc;
        g1j28_to_g1j(acc, lad, lad_inf);
        uint32_t *out = ma.out[0];
#pragma unroll
        for (int k = 0; k < 12; k++) {
            out[(size_t)i * 36 + k] = acc.x.v[k];
            out[(size_t)i * 36 + 12 + k] = acc.y.v[k];
            out[(size_t)i * 36 + 24 + k] = acc.z.v[k];
        }
    }
}
extern "C" __global__ void __launch_bounds__(256, 2)
k_g1_msm_multi(H2vDevPlan plan, H2vMsmArgs ma, uint32_t n, uint32_t per_block, uint32_t halves_per_lane, const uint32_t *__restrict__ scalars,
               const uint32_t *__restrict__ pts, uint32_t *__restrict__ tabws) {
    extern __shared__ uint32_t red[];
    (void)tabws;
    msm_multi_body(plan, ma, n, per_block, halves_per_lane, scalars, pts, red);
}

// er += er_fix (complete Jacobian addition, one lane per proof): joins the two launches of a split MSM
extern "C" __global__ void __launch_bounds__(64)
k_g1_sum_pairs(uint32_t n, uint32_t *__restrict__ er, const uint32_t *__restrict__ er_fix) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1J a, b, r;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        a.x.v[k] = er[(size_t)i * 36 + k]; a.y.v[k] = er[(size_t)i * 36 + 12 + k]; a.z.v[k] = er[(size_t)i * 36 + 24 + k];
        b.x.v[k] = er_fix[(size_t)i * 36 + k]; b.y.v[k] = er_fix[(size_t)i * 36 + 12 + k]; b.z.v[k] = er_fix[(size_t)i * 36 + 24 + k];
    }
    g1j_add(r, a, b);
#pragma unroll
    for (int k = 0; k < 12; k++) { er[(size_t)i * 36 + k] = r.x.v[k]; er[(size_t)i * 36 + 12 + k] = r.y.v[k]; er[(size_t)i * 36 + 24 + k] = r.z.v[k]; }
}

// ============================================================================ K5: pairing check
// accept <=> e(el, s_g2) == e(er, G2)  (verification_h2.hbs:125-128), evaluated as
//   FE( ML(el; lines(s_g2)) * ML(-er; lines(G2)) ) == 1 .
// Both G2 arguments are fixed per plan, so their Miller-loop line coefficients (lambda, c) are precomputed at
// plan-compile time; with the untwist (x,y)->(x/w^2, y/w^3) a line at P=(xP,yP), scaled by w^3 (subfield
// element, killed by the final exponentiation), is  c + (-lambda xP) w^2 + yP w^3.
// Final exponentiation: easy part (p^6-1)(p^2+1); hard part via
//   3 (p^4-p^2+1)/r = (x-1)^2 (x+p)(x^2+p^2-1) + 3      (factor 3 is coprime to r: "== 1" unchanged).
H2V_DI void load_line(Fp2 &lam, Fp2 &c, const uint32_t *tab, int idx) {
    const uint32_t *p = tab + (size_t)idx * 48;
#pragma unroll
    for (int k = 0; k < 12; k++) { lam.c0.v[k] = p[k]; lam.c1.v[k] = p[12 + k]; c.c0.v[k] = p[24 + k]; c.c1.v[k] = p[36 + k]; }
}
H2V_DN void miller_loop(Fp12 &f, const G1A &p1, const uint32_t *lines1, bool skip1, const G1A &p2, const uint32_t *lines2, bool skip2) {
    fp12_set_one(f);
    int n = 0;
    Fp2 lam, c, b;
    for (int i = 62; i >= 0; i--) {
        fp12_sqr(f, f);
        const int steps = ((BLS_X_ABS >> i) & 1) ? 2 : 1;
        for (int s = 0; s < steps; s++) {
            if (!skip1) {
                load_line(lam, c, lines1, n);
                fp2_mul_fp(b, lam, p1.x); fp2_neg(b, b);
                fp12_mul_line(f, c, b, p1.y);
            }
            if (!skip2) {
                load_line(lam, c, lines2, n);
                fp2_mul_fp(b, lam, p2.x); fp2_neg(b, b);
                fp12_mul_line(f, c, b, p2.y);
            }
            n++;
        }
    }
    fp12_conj(f, f);  // x < 0
}
// a^x for a in the cyclotomic subgroup (x negative: conjugate)
H2V_DN void fp12_exp_x(Fp12 &r, const Fp12 &a) {
    Fp12 acc = a;
    for (int i = 62; i >= 0; i--) {
        fp12_sqr(acc, acc);
        if ((BLS_X_ABS >> i) & 1) fp12_mul(acc, acc, a);
    }
    fp12_conj(r, acc);
}
H2V_DN bool final_exp_is_one(const Fp12 &f) {
    Fp12 t, a, u, t0, t1, t2, t3;
    if (!fp12_inv(a, f)) return false;
    fp12_conj(t, f); fp12_mul(t, t, a);                       // f^(p^6-1)
    fp12_frob(a, t); fp12_frob(a, a); fp12_mul(t, a, t);      // ^(p^2+1)
    fp12_exp_x(a, t); fp12_conj(u, t); fp12_mul(t0, a, u);    // t^(x-1)
    fp12_exp_x(a, t0); fp12_conj(u, t0); fp12_mul(t1, a, u);  // ^(x-1)
    fp12_exp_x(a, t1); fp12_frob(u, t1); fp12_mul(t2, a, u);  // ^(x+p)
    fp12_exp_x(a, t2); fp12_exp_x(a, a);
    fp12_frob(u, t2); fp12_frob(u, u); fp12_mul(t3, a, u);
    fp12_conj(u, t2); fp12_mul(t3, t3, u);                    // ^(x^2+p^2-1)
    fp12_sqr(u, t); fp12_mul(u, u, t);                        // t^3
    fp12_mul(t3, t3, u);
    return fp12_is_one(t3);
}

extern "C" __global__ void __launch_bounds__(64)
k_pairing_check(H2vDevPlan plan, uint32_t n, const uint32_t *__restrict__ pts, const uint8_t *__restrict__ valid, const uint8_t *__restrict__ valid_sub,
                const uint32_t *__restrict__ er_jac, const uint32_t *__restrict__ el_jac /* folded el (recursion) or NULL */,
                uint32_t *__restrict__ status, uint8_t *__restrict__ accept, uint32_t *__restrict__ dbg) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t slots = H2V_SLOTS(plan);
    uint32_t st = status[i];
    for (uint32_t j = 0; j < slots; j++)
        if (!valid[(size_t)i * slots + j] || (valid_sub && !valid_sub[(size_t)i * slots + j])) st |= H2V_ST_BAD_POINT;
    if (st == 0) {
        G1A el, er;
        G1J ej;
        const uint32_t *pp = pts + ((size_t)i * slots + plan.pi_point) * 24;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            el.x.v[k] = pp[k]; el.y.v[k] = pp[12 + k];
            ej.x.v[k] = er_jac[(size_t)i * 36 + k]; ej.y.v[k] = er_jac[(size_t)i * 36 + 12 + k]; ej.z.v[k] = er_jac[(size_t)i * 36 + 24 + k];
        }
        g1j_to_affine(er, ej);
        if (el_jac) {
#pragma unroll
            for (int k = 0; k < 12; k++) { ej.x.v[k] = el_jac[(size_t)i * 36 + k]; ej.y.v[k] = el_jac[(size_t)i * 36 + 12 + k]; ej.z.v[k] = el_jac[(size_t)i * 36 + 24 + k]; }
            g1j_to_affine(el, ej);
        }
        const bool el_inf = g1a_is_inf(el), er_inf = g1a_is_inf(er);
        fp_neg(er.y, er.y);  // -er (harmless on the infinity sentinel: skipped below)
        Fp12 f;
        miller_loop(f, el, plan.lines_sg2, el_inf, er, plan.lines_g2, er_inf);
        if (dbg) {  // flat order (k, part): tower c0 = (w^0, w^2, w^4), c1 = (w^1, w^3, w^5)
            const Fp2 *co[6] = {&f.c0.c0, &f.c1.c0, &f.c0.c1, &f.c1.c1, &f.c0.c2, &f.c1.c2};
            for (int k = 0; k < 6; k++) {
                Fp o0, o1;
                fp_from_mont(o0, co[k]->c0);
                fp_from_mont(o1, co[k]->c1);
                for (int q = 0; q < 12; q++) { dbg[((size_t)i * 24 + 2 * k) * 12 + q] = o0.v[q]; dbg[((size_t)i * 24 + 2 * k + 1) * 12 + q] = o1.v[q]; }
            }
        }
        if (!final_exp_is_one(f)) st |= H2V_ST_PAIRING;
    }
    status[i] = st;
    accept[i] = st == 0 ? 1 : 0;
}

// ============================================================================ K4b: recursion (IVC) batching challenge
// c = LE(blake2b_256(compress(el) || compress(er) || compress(acc_left) || compress(acc_right_final))) mod r
// (emitters/aiken.rs:743-748).  One lane per proof: the three Jacobian sums are normalised with ONE inversion
// (Montgomery's trick), compressed (x big-endian + flags, sign = "y > (p-1)/2"), hashed, and the four affine points and
// the scalars {1, c, 1, c} are written for the two fold MSMs  el' = el + c acc_left ,  er' = er + c acc_right_final.
H2V_DN void g1_compress_dev(uint8_t (&out)[48], const G1A &a) {
    if (g1a_is_inf(a)) {
#pragma unroll
        for (int k = 0; k < 48; k++) out[k] = 0;
        out[0] = 0xc0;
        return;
    }
    Fp x;
    fp_from_mont(x, a.x);
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const uint32_t wv = x.v[11 - k];
        out[4 * k] = wv >> 24; out[4 * k + 1] = wv >> 16; out[4 * k + 2] = wv >> 8; out[4 * k + 3] = wv;
    }
    out[0] |= fp_is_lex_larger(a.y) ? 0xa0 : 0x80;
}
extern "C" __global__ void __launch_bounds__(64)
k_ivc_challenge(H2vDevPlan plan, uint32_t n, const uint32_t *__restrict__ pts, const uint32_t *__restrict__ er_jac,
                const uint32_t *__restrict__ accl_jac, const uint32_t *__restrict__ accr_jac,
                uint32_t *__restrict__ fold_pts /* n x 4 x 24 */, uint32_t *__restrict__ fold_scal /* n x 4 x 8 */) {
    __shared__ uint32_t sbuf[32 * 64];
    const int lane = threadIdx.x;
    const uint32_t i = blockIdx.x * 64 + lane;
    const bool live = i < n;
    const uint32_t ii = live ? i : n - 1;
    const uint32_t slots = H2V_SLOTS(plan);
    G1A p[4];   // el, er, acc_left, acc_right_final
    {
        const uint32_t *pp = pts + ((size_t)ii * slots + plan.pi_point) * 24;
#pragma unroll
        for (int k = 0; k < 12; k++) { p[0].x.v[k] = pp[k]; p[0].y.v[k] = pp[12 + k]; }
    }
    {
        // batch normalisation of the three Jacobian points; a point at infinity contributes Z := 1 to the product
        G1J j[3];
        const uint32_t *src[3] = {er_jac, accl_jac, accr_jac};
        Fp z[3], pre[3], inv, t;
        bool inf[3];
        for (int q = 0; q < 3; q++) {
#pragma unroll
            for (int k = 0; k < 12; k++) { j[q].x.v[k] = src[q][(size_t)ii * 36 + k]; j[q].y.v[k] = src[q][(size_t)ii * 36 + 12 + k]; j[q].z.v[k] = src[q][(size_t)ii * 36 + 24 + k]; }
            inf[q] = g1j_is_inf(j[q]);
            z[q] = j[q].z;
            if (inf[q]) fp_set_one(z[q]);
        }
        pre[0] = z[0];
        fp_mul(pre[1], pre[0], z[1]);
        fp_mul(pre[2], pre[1], z[2]);
        (void)fp_inv(inv, pre[2]);          // never zero: every factor is a non-zero Z or 1
        for (int q = 2; q >= 0; q--) {
            Fp zi;
            if (q > 0) { fp_mul(zi, inv, pre[q - 1]); fp_mul(t, inv, z[q]); inv = t; } else zi = inv;
            G1A &o = p[q + 1];
            if (inf[q]) { g1a_set_inf(o); continue; }
            Fp zi2;
            fp_sqr(zi2, zi);
            fp_mul(o.x, j[q].x, zi2);
            fp_mul(zi2, zi2, zi);
            fp_mul(o.y, j[q].y, zi2);
        }
    }
    Transcript tr;
    tr_init(tr);
#pragma unroll 1
    for (int q = 0; q < 4; q++) {
        uint8_t enc[48];
        g1_compress_dev(enc, p[q]);
#pragma unroll 1
        for (int k = 0; k < 48; k++) tr_put(tr, sbuf, lane, enc[k]);
    }
    uint64_t h[4];
    tr_digest(tr, sbuf, lane, h);
    Fr c, cm;
#pragma unroll
    for (int l = 0; l < 4; l++) { c.v[2 * l] = (uint32_t)h[l]; c.v[2 * l + 1] = (uint32_t)(h[l] >> 32); }
    FrF::to_mont(cm, c);     // reduces the 256-bit integer mod r
    FrF::from_mont(c, cm);
    if (live) {
        const int order[4] = {0, 2, 1, 3};   // fold points: el, acc_left, er, acc_right_final
#pragma unroll 1
        for (int q = 0; q < 4; q++) {
            const G1A &o = p[order[q]];
            uint32_t *dp = fold_pts + ((size_t)i * 4 + q) * 24;
            uint32_t *ds = fold_scal + ((size_t)i * 4 + q) * 8;
#pragma unroll
            for (int k = 0; k < 12; k++) { dp[k] = o.x.v[k]; dp[12 + k] = o.y.v[k]; }
#pragma unroll
            for (int l = 0; l < 8; l++) ds[l] = (q & 1) ? c.v[l] : (l == 0 ? 1u : 0u);
        }
    }
}

// ============================================================================ plan load: window tables of the VK bases
extern "C" __global__ void __launch_bounds__(64)
k_vk_tables(const uint32_t *__restrict__ vk_bases, uint32_t n, uint32_t *__restrict__ vk_tab) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    G1A base;
#pragma unroll
    for (int k = 0; k < 12; k++) { base.x.v[k] = vk_bases[(size_t)b * 24 + k]; base.y.v[k] = vk_bases[(size_t)b * 24 + 12 + k]; }
    if (g1a_is_inf(base)) return;   // the MSM skips an infinite base before it looks at the table
    g1_build_window_tables_glv(vk_tab + (size_t)b * 448, base);
}

// All-window tables of the VK bases for the fixed-base MSM lanes, window width c (4, 8, 12): for every base b and window w
// the multiples [e + 1] (2^(c w) V_b), e = 0 .. 2^(c-1) - 1, affine, at fix_tab[((b W + w) 2^(c-1) + e) 28 ...].  Plan-load
// work (once per key): the larger c is, the fewer additions a VK term costs in every proof (65 / 33 / 22 per base) - paid
// for in HBM, which this part has (c = 12: 5 MB per base).  Two launches: the window bases 2^(c w) V_b (one lane per
// (b, w): c w doublings), then one lane per table entry (double-and-add over the c - 1 bits of e + 1, one inversion).
extern "C" __global__ void __launch_bounds__(64)
k_vk_fixed_window_bases(const uint32_t *__restrict__ vk_bases, uint32_t n, uint32_t c, uint32_t W, uint32_t *__restrict__ wbase /* [n][W][24] affine, Montgomery; all-zero = infinity */) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t / W, w = t - b * W;
    if (b >= n) return;
    G1A base;
#pragma unroll
    for (int k = 0; k < 12; k++) { base.x.v[k] = vk_bases[(size_t)b * 24 + k]; base.y.v[k] = vk_bases[(size_t)b * 24 + 12 + k]; }
    uint32_t *dst = wbase + (size_t)t * 24;
    if (g1a_is_inf(base)) {
#pragma unroll
        for (int k = 0; k < 24; k++) dst[k] = 0;
        return;
    }
    G1J28 p;
    g1j28_from_affine(p, base);
#pragma unroll 1
    for (uint32_t q = 0; q < c * w; q++) g1j28_dbl_ool(p, p);   // a point of prime order: never infinity
    G1J pj;
    g1j28_to_g1j(pj, p, false);
    G1A aff;
    g1j_to_affine(aff, pj);
#pragma unroll
    for (int k = 0; k < 12; k++) { dst[k] = aff.x.v[k]; dst[12 + k] = aff.y.v[k]; }
}
extern "C" __global__ void __launch_bounds__(64)
k_vk_fixed_tables(const uint32_t *__restrict__ wbase, uint32_t n_bw /* bases x windows */, uint32_t c, uint32_t *__restrict__ fix_tab) {
    const uint32_t E = 1u << (c - 1);
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t bw = (uint32_t)(t / E), e = (uint32_t)(t - (uint64_t)bw * E);
    if (bw >= n_bw) return;
    G1A base;
    const uint32_t *src = wbase + (size_t)bw * 24;
#pragma unroll
    for (int k = 0; k < 12; k++) { base.x.v[k] = src[k]; base.y.v[k] = src[12 + k]; }
    if (g1a_is_inf(base)) return;             // (the MSM skips an infinite base before it looks at the table)
    G1J28 p, acc;
    g1j28_from_affine(p, base);
    bool inf = true;
    const uint32_t m = e + 1;                 // 1 .. 2^(c-1): up to c bits
#pragma unroll 1
    for (int bit = (int)c - 1; bit >= 0; bit--) {
        if (!inf) g1j28_dbl_ool(acc, acc);
        if ((m >> bit) & 1u) g1j28_acc_add(acc, inf, p, false);   // complete (m = 2: P + P)
    }
    G1J pj;
    g1j28_to_g1j(pj, acc, inf);
    G1A aff;
    g1j_to_affine(aff, pj);
    F28 x, y;
    f28_from_fp(x, aff.x);
    f28_from_fp(y, aff.y);
    g1_store_table_entry(fix_tab + (size_t)t * 28, x, y);
}

// ============================================================================ primitive probes (parity tests)
// op: 0 fp_mul, 1 fp_add, 2 fp_sub, 3 fp_inv, 4 fr_mul, 5 fr_inv (a, b canonical little-endian limbs; 12 or 8)
extern "C" __global__ void k_probe_field(int op, uint32_t n, const uint32_t *__restrict__ a, const uint32_t *__restrict__ b,
                                         uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (op < 4) {
        Fp x, y, r;
#pragma unroll
        for (int k = 0; k < 12; k++) { x.v[k] = a[(size_t)i * 12 + k]; y.v[k] = b[(size_t)i * 12 + k]; }
        fp_to_mont(x, x); fp_to_mont(y, y);
        if (op == 0) fp_mul(r, x, y);
        else if (op == 1) fp_add(r, x, y);
        else if (op == 2) fp_sub(r, x, y);
        else fp_inv(r, x);
        fp_from_mont(r, r);
#pragma unroll
        for (int k = 0; k < 12; k++) out[(size_t)i * 12 + k] = r.v[k];
    } else {
        Fr x, y, r;
#pragma unroll
        for (int k = 0; k < 8; k++) { x.v[k] = a[(size_t)i * 8 + k]; y.v[k] = b[(size_t)i * 8 + k]; }
        FrF::to_mont(x, x); FrF::to_mont(y, y);
        if (op == 4) fr_mul(r, x, y);
        else fr_inv(r, x);
        FrF::from_mont(r, r);
#pragma unroll
        for (int k = 0; k < 8; k++) out[(size_t)i * 8 + k] = r.v[k];
    }
}
// dev probe: 2P + (neg ? -Q : Q) by the one-lane mixed addition and by the quad-cooperative one (P, Q affine canonical limbs);
// out: 5 x 42 dwords of raw lazily reduced limbs: the one-lane result, then the quad's lanes 0..3
extern "C" __global__ void __launch_bounds__(64)
k_probe_quad_madd(const uint32_t *__restrict__ pq, int neg, uint32_t *__restrict__ out) {
    G1A P, Q;
#pragma unroll
    for (int k = 0; k < 12; k++) { P.x.v[k] = pq[k]; P.y.v[k] = pq[12 + k]; Q.x.v[k] = pq[24 + k]; Q.y.v[k] = pq[36 + k]; }
    fp_to_mont(P.x, P.x); fp_to_mont(P.y, P.y); fp_to_mont(Q.x, Q.x); fp_to_mont(Q.y, Q.y);
    G1J28 p1, p2, a, b;
    g1j28_from_affine(p1, P);
    g1j28_dbl(p2, p1);
    F28 qx, qy;
    f28_from_fp(qx, Q.x);
    f28_from_fp(qy, Q.y);
    g1j28_madd_ladder(a, p2, qx, qy, neg != 0);
    b = p2;
    g1j28_madd_quad(b, qx, qy, neg != 0);
    const int l = threadIdx.x;
    if (l == 0)
        for (int k = 0; k < 14; k++) { out[k] = a.x.l[k]; out[14 + k] = a.y.l[k]; out[28 + k] = a.z.l[k]; }
    if (l < 4)
        for (int k = 0; k < 14; k++) { out[42 * (1 + l) + k] = b.x.l[k]; out[42 * (1 + l) + 14 + k] = b.y.l[k]; out[42 * (1 + l) + 28 + k] = b.z.l[k]; }
}
// blake2b-256 of n messages of `len` bytes each through the LDS transcript path (one message per lane)
extern "C" __global__ void __launch_bounds__(64)
k_probe_blake2b(uint32_t n, uint32_t len, const uint8_t *__restrict__ msgs, uint32_t *__restrict__ out) {
    __shared__ uint32_t sbuf[32 * 64];
    const int lane = threadIdx.x;
    const uint32_t i = blockIdx.x * 64 + lane;
    const uint32_t ii = i < n ? i : n - 1;
    Transcript tr;
    tr_init(tr);
    for (uint32_t k = 0; k < len; k++) tr_put(tr, sbuf, lane, msgs[(size_t)ii * len + k]);
    uint64_t h[4];
    tr_digest(tr, sbuf, lane, h);
    if (i < n) {
#pragma unroll
        for (int k = 0; k < 4; k++) { out[(size_t)i * 8 + 2 * k] = (uint32_t)h[k]; out[(size_t)i * 8 + 2 * k + 1] = (uint32_t)(h[k] >> 32); }
    }
}

// affine (24 dwords) or Jacobian (36 dwords) Montgomery points -> 96 bytes x||y big-endian canonical (zero = infinity)
extern "C" __global__ void k_export_points(uint32_t n, int jacobian, const uint32_t *__restrict__ in, uint8_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1A a;
    if (jacobian) {
        G1J j;
#pragma unroll
        for (int k = 0; k < 12; k++) { j.x.v[k] = in[(size_t)i * 36 + k]; j.y.v[k] = in[(size_t)i * 36 + 12 + k]; j.z.v[k] = in[(size_t)i * 36 + 24 + k]; }
        g1j_to_affine(a, j);
    } else {
#pragma unroll
        for (int k = 0; k < 12; k++) { a.x.v[k] = in[(size_t)i * 24 + k]; a.y.v[k] = in[(size_t)i * 24 + 12 + k]; }
    }
    Fp x, y;
    fp_from_mont(x, a.x);
    fp_from_mont(y, a.y);
    uint8_t *o = out + (size_t)i * 96;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const uint32_t wx = x.v[11 - k], wy = y.v[11 - k];
        o[4 * k] = wx >> 24; o[4 * k + 1] = wx >> 16; o[4 * k + 2] = wx >> 8; o[4 * k + 3] = wx;
        o[48 + 4 * k] = wy >> 24; o[48 + 4 * k + 1] = wy >> 16; o[48 + 4 * k + 2] = wy >> 8; o[48 + 4 * k + 3] = wy;
    }
}

// ============================================================================ RLC batch mode + bucket MSM
#include "h2v_rlc.hpp"
#include "h2v_coalesce.hpp"
