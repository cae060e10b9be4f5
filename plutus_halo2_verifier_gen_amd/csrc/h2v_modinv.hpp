// Modular inversion by batched division steps (Bernstein-Yang "safegcd", half-delta variant): 30 branch-free
// division steps on the low words produce a 2x2 transition matrix, which is then applied to the full-length
// (f, g) and, modulo M with an exact division by 2^30, to (d, e).  About 26 batches for the 381-bit base field and
// 18 for the scalar field, ~800 instructions each, versus ~350 instructions for every single bit of the binary
// (Kaliski) inversion this replaces - which was a third of the pairing kernel's instruction count, run on one lane.
//
// Numbers are signed, L limbs of 30 bits (value = sum v[i] 2^(30 i), lower limbs in [0, 2^30), top limb signed).
// Invariants (as in the published algorithm): d*x = f, e*x = g (mod M) up to the common power of two removed by the
// exact divisions, |f|, |g| <= M, d, e in (-2M, M).  The loop ends when g == 0; then f = +-1 (M prime, x != 0) and
// x^-1 = +-d.  Model and cross-check: tests/test_host_logic.py::test_safegcd_model.
#pragma once
#include <stdint.h>
// (included by h2v_field.hpp after its H2V_DI / H2V_DN definitions)

#define H2V_M30 0x3fffffff

template <int L>
struct ModInv30 {
    // one batch of 30 division steps on the low words; t = {u, v, q, r} with [f', g'] = t [f, g] / 2^30
    H2V_DI static int32_t divsteps(int32_t zeta, uint32_t f, uint32_t g, int32_t (&t)[4]) {
        int32_t u = 1, v = 0, q = 0, r = 1;
#pragma unroll 6
        for (int i = 0; i < 30; i++) {
            int32_t c1 = zeta >> 31;
            const int32_t c2 = -(int32_t)(g & 1u);
            const uint32_t x = (f ^ (uint32_t)c1) - (uint32_t)c1;
            const int32_t y = (u ^ c1) - c1, z = (v ^ c1) - c1;
            g += x & (uint32_t)c2; q += y & c2; r += z & c2;
            c1 &= c2;
            zeta = (zeta ^ c1) - 1;
            f += g & (uint32_t)c1; u += q & c1; v += r & c1;
            g >>= 1; u <<= 1; v <<= 1;
        }
        t[0] = u; t[1] = v; t[2] = q; t[3] = r;
        return zeta;
    }
    H2V_DI static void update_fg(int32_t (&f)[L], int32_t (&g)[L], const int32_t (&t)[4]) {
        const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
        int64_t cf = u * f[0] + v * g[0], cg = q * f[0] + r * g[0];
        cf >>= 30; cg >>= 30;   // the low 30 bits are zero by construction
#pragma unroll
        for (int i = 1; i < L; i++) {
            cf += u * f[i] + v * g[i];
            cg += q * f[i] + r * g[i];
            f[i - 1] = (int32_t)cf & H2V_M30; g[i - 1] = (int32_t)cg & H2V_M30;
            cf >>= 30; cg >>= 30;
        }
        f[L - 1] = (int32_t)cf; g[L - 1] = (int32_t)cg;
    }
    H2V_DI static void update_de(int32_t (&d)[L], int32_t (&e)[L], const int32_t (&t)[4], const uint32_t *mod30, uint32_t minv30) {
        const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
        const int32_t sd = d[L - 1] >> 31, se = e[L - 1] >> 31;
        int32_t md = (t[0] & sd) + (t[1] & se), me = (t[2] & sd) + (t[3] & se);
        int64_t cd = u * d[0] + v * e[0], ce = q * d[0] + r * e[0];
        md -= (int32_t)((minv30 * (uint32_t)cd + (uint32_t)md) & H2V_M30);
        me -= (int32_t)((minv30 * (uint32_t)ce + (uint32_t)me) & H2V_M30);
        cd += (int64_t)mod30[0] * md; ce += (int64_t)mod30[0] * me;
        cd >>= 30; ce >>= 30;
#pragma unroll
        for (int i = 1; i < L; i++) {
            cd += u * d[i] + v * e[i] + (int64_t)mod30[i] * md;
            ce += q * d[i] + r * e[i] + (int64_t)mod30[i] * me;
            d[i - 1] = (int32_t)cd & H2V_M30; e[i - 1] = (int32_t)ce & H2V_M30;
            cd >>= 30; ce >>= 30;
        }
        d[L - 1] = (int32_t)cd; e[L - 1] = (int32_t)ce;
    }
    // out = x^-1 mod M as NW 32-bit words; x given as NW words, 0 < x < M (x == 0 returns 0)
    template <int NW>
    H2V_DI static void inverse(uint32_t (&out)[NW], const uint32_t (&x)[NW], const uint32_t *mod30, uint32_t minv30) {
        int32_t f[L], g[L], d[L], e[L];
#pragma unroll
        for (int i = 0; i < L; i++) {
            const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
            uint32_t lo = w < NW ? x[w] : 0u, hi = (w + 1 < NW) ? x[w + 1] : 0u;
            const uint32_t val = sh ? ((lo >> sh) | (sh > 2 ? (hi << (32 - sh)) : 0u)) : lo;
            g[i] = (int32_t)(val & H2V_M30);
            f[i] = (int32_t)mod30[i];
            d[i] = 0; e[i] = 0;
        }
        e[0] = 1;
        int32_t zeta = -1;
#pragma unroll 1
        for (int n = 0; n < 48; n++) {
            int32_t t[4];
            zeta = divsteps(zeta, (uint32_t)f[0] | ((uint32_t)f[1] << 30), (uint32_t)g[0] | ((uint32_t)g[1] << 30), t);
            update_de(d, e, t, mod30, minv30);
            update_fg(f, g, t);
            int32_t any = 0;
#pragma unroll
            for (int i = 0; i < L; i++) any |= g[i];
            if (any == 0) break;
        }
        // x^-1 = sign(f) * d, brought to [0, M)
        const int32_t neg = f[L - 1] >> 31;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            const int32_t v = ((d[i] ^ neg) - neg) + c;
            if (i < L - 1) { d[i] = v & H2V_M30; c = v >> 30; } else d[i] = v;
        }
#pragma unroll
        for (int rep = 0; rep < 2; rep++) {
            const int32_t m = d[L - 1] >> 31;
            c = 0;
#pragma unroll
            for (int i = 0; i < L; i++) {
                const int32_t v = d[i] + (int32_t)(mod30[i] & (uint32_t)m) + c;
                if (i < L - 1) { d[i] = v & H2V_M30; c = v >> 30; } else d[i] = v;
            }
        }
        {   // d in [0, 2M): subtract M once if it does not go negative
            int32_t s[L];
            c = 0;
#pragma unroll
            for (int i = 0; i < L; i++) {
                const int32_t v = d[i] - (int32_t)mod30[i] + c;
                if (i < L - 1) { s[i] = v & H2V_M30; c = v >> 30; } else s[i] = v;
            }
            const bool use = s[L - 1] >= 0;
#pragma unroll
            for (int i = 0; i < L; i++) d[i] = use ? s[i] : d[i];
        }
#pragma unroll
        for (int j = 0; j < NW; j++) {
            const int bit = 32 * j, i = bit / 30, sh = bit % 30;
            uint32_t v = (uint32_t)d[i] >> sh;
            if (i + 1 < L) v |= (uint32_t)d[i + 1] << (30 - sh);
            if (i + 2 < L && 60 - sh < 32) v |= (uint32_t)d[i + 2] << (60 - sh);
            out[j] = v;
        }
    }
};
