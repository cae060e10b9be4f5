/* ORACLE - test infrastructure, not product code.
 *
 * blake2b-256 (RFC 7693, unkeyed, no salt/personal) and the Fiat-Shamir transcript of the reference:
 *   hash plug-in  /root/reference/src/plutus_gen/adjusted_types/mod.rs:30-72
 *                 (absorb = 0x01 || bytes ; squeeze = 0x00, h = finalize, h2 = blake2b256(h), h||h2 (64 B)
 *                  -> Scalar::from_uniform_bytes  == LE(h) + LE(h2) * 2^256  mod r)
 *   restatements  aiken-verifier/aiken_halo2/lib/transcript.ak:19-106,
 *                 plinth-verifier/plutus-halo2/src/Plutus/Crypto/Halo2/Transcript.hs:74-102
 */
#ifndef ORC_TRANSCRIPT_H
#define ORC_TRANSCRIPT_H
#include "field.h"

typedef struct {
    uint64_t h[8];
    uint64_t t;       /* bytes compressed so far (low word; inputs are < 2^64) */
    uint8_t buf[128];
    size_t buflen;
} blake2b_state;

static const uint64_t BLAKE2B_IV[8] = {
    0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
    0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
static const uint8_t BLAKE2B_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

static inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
static inline void blake2b_compress(blake2b_state *s, const uint8_t *block, int last) {
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; i++) {
        uint64_t w = 0;
        for (int j = 7; j >= 0; j--) w = (w << 8) | block[i * 8 + j];
        m[i] = w;
    }
    for (int i = 0; i < 8; i++) { v[i] = s->h[i]; v[i + 8] = BLAKE2B_IV[i]; }
    v[12] ^= s->t;
    if (last) v[14] = ~v[14];
#define B2G(a, b, c, d, x, y) \
    v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32); v[c] = v[c] + v[d]; v[b] = rotr64(v[b] ^ v[c], 24); \
    v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = rotr64(v[b] ^ v[c], 63);
    for (int r = 0; r < 12; r++) {
        const uint8_t *sg = BLAKE2B_SIGMA[r];
        B2G(0, 4, 8, 12, m[sg[0]], m[sg[1]]) B2G(1, 5, 9, 13, m[sg[2]], m[sg[3]])
        B2G(2, 6, 10, 14, m[sg[4]], m[sg[5]]) B2G(3, 7, 11, 15, m[sg[6]], m[sg[7]])
        B2G(0, 5, 10, 15, m[sg[8]], m[sg[9]]) B2G(1, 6, 11, 12, m[sg[10]], m[sg[11]])
        B2G(2, 7, 8, 13, m[sg[12]], m[sg[13]]) B2G(3, 4, 9, 14, m[sg[14]], m[sg[15]])
    }
#undef B2G
    for (int i = 0; i < 8; i++) s->h[i] ^= v[i] ^ v[i + 8];
}
static inline void blake2b256_init(blake2b_state *s) {
    for (int i = 0; i < 8; i++) s->h[i] = BLAKE2B_IV[i];
    s->h[0] ^= 0x01010000ULL ^ 32; /* depth 1, fanout 1, key 0, digest 32 */
    s->t = 0;
    s->buflen = 0;
}
static inline void blake2b_update(blake2b_state *s, const uint8_t *in, size_t len) {
    while (len) {
        if (s->buflen == 128) { /* buffer full and more input follows: it is not the last block */
            s->t += 128;
            blake2b_compress(s, s->buf, 0);
            s->buflen = 0;
        }
        size_t take = 128 - s->buflen;
        if (take > len) take = len;
        memcpy(s->buf + s->buflen, in, take);
        s->buflen += take; in += take; len -= take;
    }
}
/* finalize a COPY of the state (the running state stays usable, as blake2b_simd::State::finalize) */
static inline void blake2b256_final_copy(const blake2b_state *s0, uint8_t out[32]) {
    blake2b_state s = *s0;
    s.t += s.buflen;
    memset(s.buf + s.buflen, 0, 128 - s.buflen);
    blake2b_compress(&s, s.buf, 1);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) out[i * 8 + j] = (uint8_t)(s.h[i] >> (8 * j));
}
static inline void blake2b256(uint8_t out[32], const uint8_t *in, size_t len) {
    blake2b_state s;
    blake2b256_init(&s);
    blake2b_update(&s, in, len);
    blake2b256_final_copy(&s, out);
}

/* ------------------------------------------------------------------ transcript */
typedef struct {
    blake2b_state st;
    const uint8_t *proof;
    size_t len, pos;
} transcript;

static inline void tr_init(transcript *t, const uint8_t *proof, size_t len) {
    blake2b256_init(&t->st);
    t->proof = proof; t->len = len; t->pos = 0;
}
static inline void tr_absorb(transcript *t, const uint8_t *b, size_t n) {
    uint8_t pfx = 1; /* BLAKE2B_PREFIX_COMMON, adjusted_types/mod.rs:11 */
    blake2b_update(&t->st, &pfx, 1);
    blake2b_update(&t->st, b, n);
}
/* common_scalar (transcript.ak:47-58) */
static inline void tr_common_scalar(transcript *t, const fr *s) {
    uint8_t b[32];
    fr_to_le32(b, s);
    tr_absorb(t, b, 32);
}
/* common_g1 (transcript.ak:60-66): 48 compressed bytes */
static inline void tr_common_point(transcript *t, const uint8_t *p48) { tr_absorb(t, p48, 48); }
/* read_scalar (transcript.ak:29-45).  returns 0 = short proof; *canonical tells whether bytes < r.
 * The Rust reader rejects non-canonical encodings (as Plinth's mkScalar, BlsTypes.hs:129-132);
 * the Aiken builtin silently reduces (transcript.ak:169-179): `s` holds the reduced value either way. */
static inline int tr_read_scalar(transcript *t, fr *s, int *canonical) {
    if (t->pos + 32 > t->len) return 0;
    const uint8_t *b = t->proof + t->pos;
    t->pos += 32;
    *canonical = fr_from_le32(s, b);
    tr_absorb(t, b, 32);
    return 1;
}
/* read_point (transcript.ak:68-83): returns pointer to the 48 raw bytes, NULL = short proof */
static inline const uint8_t *tr_read_point(transcript *t) {
    if (t->pos + 48 > t->len) return 0;
    const uint8_t *b = t->proof + t->pos;
    t->pos += 48;
    tr_absorb(t, b, 48);
    return b;
}
/* squeeze_challenge (transcript.ak:85-106; adjusted_types/mod.rs:44-71) */
static inline void tr_squeeze(transcript *t, fr *c) {
    uint8_t pfx = 0, h[32], h2[32];
    blake2b_update(&t->st, &pfx, 1);
    blake2b256_final_copy(&t->st, h);
    blake2b256(h2, h, 32);
    fr a, b, k;
    fr_from_le32(&a, h);
    fr_from_le32(&b, h2);
    fr_set(&k, FR_2_256);
    fr_mul(&b, &b, &k);
    fr_add(c, &a, &b);
}
#endif
