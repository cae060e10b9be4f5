// Host-side big-integer arithmetic for the plan compiler (h2v_plan_compile): Fr, Fp, Fp2, G1 and G2 of BLS12-381 on the
// CPU, plan-time only - nothing here runs per proof and nothing here is a verification path (there is no CPU fallback:
// the verify entry points need the GPU).  It exists so that a Rust / C++ host can go from a verifying-key description to
// a plan blob through the C-ABI alone, where round 2 needed the Python package (plan.py + bls12_381.py) for it.
//
// Definitions followed (the same ones bls12_381.py restates): field primes /root/reference/plinth-verifier/plutus-halo2/
// src/Plutus/Crypto/BlsTypes.hs:97-103, curve y^2 = x^3 + 4 and the zcash compressed encoding
// (aiken-verifier/aiken_halo2/lib/bls_utils.ak:17-28; CompressUncompress.hs:70-100), G2 on the twist y^2 = x^3 + 4(1 + u).
#pragma once
#include <stdint.h>
#include <string.h>

#include <array>
#include <stdexcept>
#include <string>
#include <vector>

namespace h2vhost {

typedef unsigned __int128 u128;

template <int N>
struct UInt {
    uint64_t w[N];
    UInt() { memset(w, 0, sizeof w); }
    explicit UInt(uint64_t v) { memset(w, 0, sizeof w); w[0] = v; }
    bool is_zero() const { for (int i = 0; i < N; i++) if (w[i]) return false; return true; }
    bool bit(int k) const { return (w[k >> 6] >> (k & 63)) & 1; }
    int bits() const { for (int k = 64 * N - 1; k >= 0; k--) if (bit(k)) return k + 1; return 0; }
    bool operator==(const UInt &o) const { return memcmp(w, o.w, sizeof w) == 0; }
    bool operator!=(const UInt &o) const { return !(*this == o); }
    bool operator<(const UInt &o) const { for (int i = N - 1; i >= 0; i--) if (w[i] != o.w[i]) return w[i] < o.w[i]; return false; }
};
template <int N> static inline uint64_t add_to(UInt<N> &a, const UInt<N> &b) {
    u128 c = 0;
    for (int i = 0; i < N; i++) { c += (u128)a.w[i] + b.w[i]; a.w[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
template <int N> static inline uint64_t sub_from(UInt<N> &a, const UInt<N> &b) {
    uint64_t br = 0;
    for (int i = 0; i < N; i++) {
        const u128 d = (u128)a.w[i] - b.w[i] - br;
        a.w[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}
template <int N> static UInt<N> from_hex(const char *h) {
    UInt<N> r;
    const size_t n = strlen(h);
    for (size_t i = 0; i < n; i++) {
        const char c = h[n - 1 - i];
        const uint64_t v = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : 99;
        if (v > 15 || i / 16 >= (size_t)N) throw std::invalid_argument("bad hex constant");
        r.w[i / 16] |= v << (4 * (i % 16));
    }
    return r;
}

// A prime field with Montgomery arithmetic (R = 2^(64 N)); elements are kept in Montgomery form.
template <int N>
struct Field {
    UInt<N> p, r2, one_m;     // modulus, R^2 mod p, R mod p
    uint64_t n0;              // -p^-1 mod 2^64
    explicit Field(const char *p_hex) {
        p = from_hex<N>(p_hex);
        uint64_t inv = 1;
        for (int i = 0; i < 6; i++) inv *= 2 - p.w[0] * inv;
        n0 = ~inv + 1;
        // R mod p and R^2 mod p by doubling
        UInt<N> x(1);
        for (int i = 0; i < 64 * N; i++) dbl_mod(x);
        one_m = x;
        for (int i = 0; i < 64 * N; i++) dbl_mod(x);
        r2 = x;
    }
    void dbl_mod(UInt<N> &x) const {
        const UInt<N> y = x;
        const uint64_t c = add_to(x, y);
        if (c || !(x < p)) sub_from(x, p);
    }
    UInt<N> add(UInt<N> a, const UInt<N> &b) const {
        const uint64_t c = add_to(a, b);
        if (c || !(a < p)) sub_from(a, p);
        return a;
    }
    UInt<N> sub(UInt<N> a, const UInt<N> &b) const {
        if (sub_from(a, b)) add_to(a, p);
        return a;
    }
    UInt<N> neg(const UInt<N> &a) const { return a.is_zero() ? a : sub(UInt<N>(), a); }
    UInt<N> mul(const UInt<N> &a, const UInt<N> &b) const {   // Montgomery product (CIOS)
        uint64_t t[N + 2];
        memset(t, 0, sizeof t);
        for (int i = 0; i < N; i++) {
            u128 c = 0;
            for (int j = 0; j < N; j++) { c += (u128)a.w[j] * b.w[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[N]; t[N] = (uint64_t)c; t[N + 1] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * n0;
            c = (u128)m * p.w[0] + t[0];
            c >>= 64;
            for (int j = 1; j < N; j++) { c += (u128)m * p.w[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[N]; t[N - 1] = (uint64_t)c;
            t[N] = t[N + 1] + (uint64_t)(c >> 64);
        }
        UInt<N> r;
        memcpy(r.w, t, sizeof r.w);
        if (t[N] || !(r < p)) sub_from(r, p);
        return r;
    }
    UInt<N> sqr(const UInt<N> &a) const { return mul(a, a); }
    UInt<N> to_mont(const UInt<N> &canonical) const { return mul(canonical, r2); }
    UInt<N> from_mont(const UInt<N> &m) const { return mul(m, UInt<N>(1)); }
    UInt<N> from_u64(uint64_t v) const { return to_mont(UInt<N>(v)); }
    template <int M> UInt<N> pow(const UInt<N> &a, const UInt<M> &e) const {
        UInt<N> r = one_m;
        for (int k = e.bits() - 1; k >= 0; k--) {
            r = sqr(r);
            if (e.bit(k)) r = mul(r, a);
        }
        return r;
    }
    UInt<N> inv(const UInt<N> &a) const {    // a^(p-2); 0 for 0
        UInt<N> e = p;
        UInt<N> two(2);
        sub_from(e, two);
        return pow(a, e);
    }
    bool reduced(const UInt<N> &canonical) const { return canonical < p; }
};

typedef UInt<4> U256;
typedef UInt<6> U384;
static const Field<4> &FR() {
    static const Field<4> f("73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001");
    return f;
}
static const Field<6> &FP() {
    static const Field<6> f("1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab");
    return f;
}

// ---- byte helpers
static inline U384 u384_from_be48(const uint8_t *b) {
    U384 r;
    for (int i = 0; i < 48; i++) r.w[(47 - i) / 8] |= (uint64_t)b[i] << (8 * ((47 - i) % 8));
    return r;
}
template <int N> static inline void to_le_bytes(const UInt<N> &v, uint8_t *out) {
    for (int i = 0; i < 8 * N; i++) out[i] = (uint8_t)(v.w[i / 8] >> (8 * (i % 8)));
}
static inline int hex_nibble(char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }
static inline bool hex_to_bytes(const std::string &h, std::vector<uint8_t> &out) {
    if (h.size() % 2) return false;
    out.resize(h.size() / 2);
    for (size_t i = 0; i < out.size(); i++) {
        const int a = hex_nibble(h[2 * i]), b = hex_nibble(h[2 * i + 1]);
        if (a < 0 || b < 0) return false;
        out[i] = (uint8_t)(a << 4 | b);
    }
    return true;
}
// decimal digits -> U256; false when it does not fit
static inline bool u256_from_decimal(const std::string &s, U256 &out) {
    out = U256();
    if (s.empty()) return false;
    for (char c : s) {
        if (c < '0' || c > '9') return false;
        u128 carry = (u128)(c - '0');
        for (int i = 0; i < 4; i++) { carry += (u128)out.w[i] * 10; out.w[i] = (uint64_t)carry; carry >>= 64; }
        if (carry) return false;
    }
    return true;
}

// ---- Fp helpers (Montgomery form, R = 2^384 internally; the blob wants x 2^392 mod p)
struct FpE { U384 v; };   // Montgomery form
static inline FpE fp_from_canonical(const U384 &c) { return {FP().to_mont(c)}; }
static inline U384 fp_canonical(const FpE &a) { return FP().from_mont(a.v); }
static inline FpE fp_add(const FpE &a, const FpE &b) { return {FP().add(a.v, b.v)}; }
static inline FpE fp_sub(const FpE &a, const FpE &b) { return {FP().sub(a.v, b.v)}; }
static inline FpE fp_mul(const FpE &a, const FpE &b) { return {FP().mul(a.v, b.v)}; }
static inline FpE fp_neg(const FpE &a) { return {FP().neg(a.v)}; }
static inline FpE fp_inv(const FpE &a) { return {FP().inv(a.v)}; }
static inline bool fp_eq(const FpE &a, const FpE &b) { return a.v == b.v; }
static inline bool fp_is_zero(const FpE &a) { return a.v.is_zero(); }
static inline FpE fp_small(uint64_t k) { return {FP().from_u64(k)}; }
// y lexicographically larger than -y  <=>  y > (p - 1) / 2
static inline bool fp_lex_larger(const FpE &y) {
    const U384 c = fp_canonical(y), n = fp_canonical(fp_neg(y));
    return n < c;
}
static inline bool fp_sqrt(const FpE &a, FpE &out) {   // p = 3 mod 4: a^((p+1)/4)
    U384 e = FP().p;
    U384 one(1);
    add_to(e, one);
    for (int i = 0; i < 6; i++) e.w[i] = (e.w[i] >> 2) | (i + 1 < 6 ? e.w[i + 1] << 62 : 0);
    out = {FP().pow(a.v, e)};
    return fp_eq(fp_mul(out, out), a);
}
// the 48 little-endian bytes of x 2^392 mod p (csrc: 12 x u32 storage limbs, R = 2^392)
static inline void fp_mont392_bytes(const FpE &a, uint8_t *out) {
    static const FpE c256 = {FP().from_u64(256)};
    to_le_bytes(fp_mul(a, c256).v, out);     // Montgomery form with R = 2^384 is x 2^384; times 2^8
}

// ---- Fp2 = Fp[u] / (u^2 + 1)
struct F2 { FpE a, b; };
static inline F2 f2_add(const F2 &x, const F2 &y) { return {fp_add(x.a, y.a), fp_add(x.b, y.b)}; }
static inline F2 f2_sub(const F2 &x, const F2 &y) { return {fp_sub(x.a, y.a), fp_sub(x.b, y.b)}; }
static inline F2 f2_neg(const F2 &x) { return {fp_neg(x.a), fp_neg(x.b)}; }
static inline F2 f2_mul(const F2 &x, const F2 &y) {
    return {fp_sub(fp_mul(x.a, y.a), fp_mul(x.b, y.b)), fp_add(fp_mul(x.a, y.b), fp_mul(x.b, y.a))};
}
static inline F2 f2_sqr(const F2 &x) { return f2_mul(x, x); }
static inline F2 f2_scale(const F2 &x, uint64_t k) { const FpE s = fp_small(k); return {fp_mul(x.a, s), fp_mul(x.b, s)}; }
static inline F2 f2_conj(const F2 &x) { return {x.a, fp_neg(x.b)}; }
static inline F2 f2_inv(const F2 &x) {
    const FpE n = fp_inv(fp_add(fp_mul(x.a, x.a), fp_mul(x.b, x.b)));
    return {fp_mul(x.a, n), fp_neg(fp_mul(x.b, n))};
}
static inline bool f2_eq(const F2 &x, const F2 &y) { return fp_eq(x.a, y.a) && fp_eq(x.b, y.b); }
static inline bool f2_is_zero(const F2 &x) { return fp_is_zero(x.a) && fp_is_zero(x.b); }
static inline F2 f2_one() { return {fp_small(1), fp_small(0)}; }
static inline F2 f2_xi() { return {fp_small(1), fp_small(1)}; }
template <int M> static inline F2 f2_pow(const F2 &x, const UInt<M> &e) {
    F2 r = f2_one();
    for (int k = e.bits() - 1; k >= 0; k--) {
        r = f2_sqr(r);
        if (e.bit(k)) r = f2_mul(r, x);
    }
    return r;
}
// Algorithm 9 of Adj / Rodriguez-Henriquez (p = 3 mod 4), as bls12_381.py: f2_sqrt
static inline bool f2_sqrt(const F2 &a, F2 &out) {
    if (f2_is_zero(a)) { out = a; return true; }
    U384 e1 = FP().p, three(3), one(1);
    sub_from(e1, three);
    for (int i = 0; i < 6; i++) e1.w[i] = (e1.w[i] >> 2) | (i + 1 < 6 ? e1.w[i + 1] << 62 : 0);    // (p - 3) / 4
    const F2 a1 = f2_pow(a, e1);
    const F2 alpha = f2_mul(f2_sqr(a1), a);
    const F2 a0 = f2_mul(f2_conj(alpha), alpha);
    const F2 minus_one = {fp_neg(fp_small(1)), fp_small(0)};
    if (f2_eq(a0, minus_one)) return false;
    const F2 x0 = f2_mul(a1, a);
    F2 res;
    if (f2_eq(alpha, minus_one)) {
        res = f2_mul(F2{fp_small(0), fp_small(1)}, x0);
    } else {
        U384 e2 = FP().p;
        sub_from(e2, one);
        for (int i = 0; i < 6; i++) e2.w[i] = (e2.w[i] >> 1) | (i + 1 < 6 ? e2.w[i + 1] << 63 : 0);  // (p - 1) / 2
        res = f2_mul(f2_pow(f2_add(f2_one(), alpha), e2), x0);
    }
    out = res;
    return f2_eq(f2_sqr(res), a);
}
// zcash order: compare c1 first, then c0
static inline bool f2_lex_larger(const F2 &y) {
    const F2 n = f2_neg(y);
    const U384 y1 = fp_canonical(y.b), n1 = fp_canonical(n.b);
    if (y1 != n1) return n1 < y1;
    return fp_canonical(n.a) < fp_canonical(y.a);
}

// ---- short Weierstrass curve y^2 = x^3 + b over a field given by the operations above (affine with an infinity flag;
// plan-time code: one inversion per addition is fine)
template <class E> struct Ops;
template <> struct Ops<FpE> {
    static FpE add(const FpE &a, const FpE &b) { return fp_add(a, b); }
    static FpE sub(const FpE &a, const FpE &b) { return fp_sub(a, b); }
    static FpE mul(const FpE &a, const FpE &b) { return fp_mul(a, b); }
    static FpE inv(const FpE &a) { return fp_inv(a); }
    static FpE neg(const FpE &a) { return fp_neg(a); }
    static bool eq(const FpE &a, const FpE &b) { return fp_eq(a, b); }
    static bool zero(const FpE &a) { return fp_is_zero(a); }
    static FpE small(uint64_t k) { return fp_small(k); }
};
template <> struct Ops<F2> {
    static F2 add(const F2 &a, const F2 &b) { return f2_add(a, b); }
    static F2 sub(const F2 &a, const F2 &b) { return f2_sub(a, b); }
    static F2 mul(const F2 &a, const F2 &b) { return f2_mul(a, b); }
    static F2 inv(const F2 &a) { return f2_inv(a); }
    static F2 neg(const F2 &a) { return f2_neg(a); }
    static bool eq(const F2 &a, const F2 &b) { return f2_eq(a, b); }
    static bool zero(const F2 &a) { return f2_is_zero(a); }
    static F2 small(uint64_t k) { return F2{fp_small(k), fp_small(0)}; }
};
template <class E> struct Pt { E x, y; bool inf; };
template <class E> static Pt<E> pt_add(const Pt<E> &p, const Pt<E> &q) {
    typedef Ops<E> O;
    if (p.inf) return q;
    if (q.inf) return p;
    E lam;
    if (O::eq(p.x, q.x)) {
        if (!O::eq(p.y, q.y) || O::zero(p.y)) return Pt<E>{p.x, p.y, true};
        lam = O::mul(O::mul(O::small(3), O::mul(p.x, p.x)), O::inv(O::add(p.y, p.y)));
    } else {
        lam = O::mul(O::sub(q.y, p.y), O::inv(O::sub(q.x, p.x)));
    }
    const E x3 = O::sub(O::sub(O::mul(lam, lam), p.x), q.x);
    const E y3 = O::sub(O::mul(lam, O::sub(p.x, x3)), p.y);
    return Pt<E>{x3, y3, false};
}
// [k]P, Jacobian inside (one inversion at the end): used for the subgroup checks [r]P == O
template <class E, int M> static Pt<E> pt_mul(const Pt<E> &p, const UInt<M> &k) {
    typedef Ops<E> O;
    if (p.inf) return p;
    E X = p.x, Y = p.y, Z = O::small(1);
    bool inf = true;
    for (int bit = k.bits() - 1; bit >= 0; bit--) {
        if (!inf) {   // doubling (a = 0): dbl-2009-l
            const E A = O::mul(X, X), B = O::mul(Y, Y), C = O::mul(B, B);
            E t = O::add(X, B);
            E D = O::sub(O::sub(O::mul(t, t), A), C);
            D = O::add(D, D);
            const E Ee = O::add(O::add(A, A), A), F = O::mul(Ee, Ee);
            const E X3 = O::sub(F, O::add(D, D));
            E C8 = O::add(C, C); C8 = O::add(C8, C8); C8 = O::add(C8, C8);
            const E Y3 = O::sub(O::mul(Ee, O::sub(D, X3)), C8);
            E Z3 = O::mul(Y, Z);
            Z3 = O::add(Z3, Z3);
            X = X3; Y = Y3; Z = Z3;
            if (O::zero(Z)) inf = true;
        }
        if (k.bit(bit)) {
            if (inf) { X = p.x; Y = p.y; Z = O::small(1); inf = false; }
            else {    // mixed addition, exceptional cases by falling back to affine arithmetic
                const E Z2 = O::mul(Z, Z), U2 = O::mul(p.x, Z2), S2 = O::mul(p.y, O::mul(Z2, Z));
                if (O::eq(U2, X)) {
                    const E zi = O::inv(Z), zi2 = O::mul(zi, zi);
                    const Pt<E> a{O::mul(X, zi2), O::mul(Y, O::mul(zi2, zi)), false};
                    const Pt<E> s = pt_add(a, p);
                    if (s.inf) inf = true;
                    else { X = s.x; Y = s.y; Z = O::small(1); }
                } else {
                    const E H = O::sub(U2, X), HH = O::mul(H, H), HHH = O::mul(H, HH), Rr = O::sub(S2, Y), V = O::mul(X, HH);
                    const E X3 = O::sub(O::sub(O::mul(Rr, Rr), HHH), O::add(V, V));
                    const E Y3 = O::sub(O::mul(Rr, O::sub(V, X3)), O::mul(Y, HHH));
                    Z = O::mul(Z, H);
                    X = X3; Y = Y3;
                }
            }
        }
    }
    if (inf) return Pt<E>{p.x, p.y, true};
    const E zi = O::inv(Z), zi2 = O::mul(zi, zi);
    return Pt<E>{O::mul(X, zi2), O::mul(Y, O::mul(zi2, zi)), false};
}

typedef Pt<FpE> G1;
typedef Pt<F2> G2;
static inline G1 g1_generator() {
    static const G1 g = {fp_from_canonical(from_hex<6>("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")),
                         fp_from_canonical(from_hex<6>("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1")), false};
    return g;
}
static inline G2 g2_generator() {
    static const G2 g = {
        F2{fp_from_canonical(from_hex<6>("024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")),
           fp_from_canonical(from_hex<6>("13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"))},
        F2{fp_from_canonical(from_hex<6>("0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801")),
           fp_from_canonical(from_hex<6>("0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be"))}, false};
    return g;
}
// zcash compressed G1 (48 bytes): returns an error text, or "" and the point
static inline std::string g1_decompress(const uint8_t *b, G1 &out, bool check_subgroup = true) {
    const int flags = b[0] >> 5;
    if (!(flags & 4)) return "G1: compression flag not set";
    uint8_t raw[48];
    memcpy(raw, b, 48);
    raw[0] &= 0x1f;
    const U384 x = u384_from_be48(raw);
    if (flags & 2) {
        if (!x.is_zero() || (flags & 1)) return "G1: bad infinity encoding";
        out = G1{fp_small(0), fp_small(0), true};
        return "";
    }
    if (!FP().reduced(x)) return "G1: x not canonical";
    const FpE xe = fp_from_canonical(x);
    FpE y;
    if (!fp_sqrt(fp_add(fp_mul(fp_mul(xe, xe), xe), fp_small(4)), y)) return "G1: not on curve";
    if (fp_lex_larger(y) != (bool)(flags & 1)) y = fp_neg(y);
    out = G1{xe, y, false};
    if (check_subgroup && !pt_mul(out, FR().p).inf) return "G1: not in subgroup";
    return "";
}
static inline void g1_compress(const G1 &p, uint8_t *out) {
    memset(out, 0, 48);
    if (p.inf) { out[0] = 0xc0; return; }
    const U384 x = fp_canonical(p.x);
    for (int i = 0; i < 48; i++) out[i] = (uint8_t)(x.w[(47 - i) / 8] >> (8 * ((47 - i) % 8)));
    out[0] |= 0x80 | (fp_lex_larger(p.y) ? 0x20 : 0);
}
static inline std::string g2_decompress(const uint8_t *b, G2 &out, bool check_subgroup = true) {
    const int flags = b[0] >> 5;
    if (!(flags & 4)) return "G2: compression flag not set";
    uint8_t raw[48];
    memcpy(raw, b, 48);
    raw[0] &= 0x1f;
    const U384 x1 = u384_from_be48(raw), x0 = u384_from_be48(b + 48);
    if (flags & 2) {
        if (!x0.is_zero() || !x1.is_zero() || (flags & 1)) return "G2: bad infinity encoding";
        out = G2{F2{fp_small(0), fp_small(0)}, F2{fp_small(0), fp_small(0)}, true};
        return "";
    }
    if (!FP().reduced(x0) || !FP().reduced(x1)) return "G2: x not canonical";
    const F2 x = {fp_from_canonical(x0), fp_from_canonical(x1)};
    F2 y;
    if (!f2_sqrt(f2_add(f2_mul(f2_sqr(x), x), f2_scale(f2_xi(), 4)), y)) return "G2: not on curve";
    if (f2_lex_larger(y) != (bool)(flags & 1)) y = f2_neg(y);
    out = G2{x, y, false};
    if (check_subgroup && !pt_mul(out, FR().p).inf) return "G2: not in subgroup";
    return "";
}

}  // namespace h2vhost
