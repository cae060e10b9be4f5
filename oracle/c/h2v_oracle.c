/* ORACLE - test infrastructure, not product code (see h2v_oracle.h).
 *
 * Literal CPU restatement of the per-proof verifier the reference emits:
 *   skeleton        /root/reference/aiken-verifier/templates/verification_h2.hbs:21-129
 *   slot contents   /root/reference/src/plutus_gen/emitters/aiken.rs:94-646
 *   proof layout    /root/reference/src/plutus_gen/extraction/data/extraction_steps/proof.rs:13-143
 *   queries         /root/reference/src/plutus_gen/extraction/mod.rs:120-226
 *   point sets      /root/reference/src/plutus_gen/extraction/pcs/mod.rs:36-109
 *   multi-open      /root/reference/aiken-verifier/aiken_halo2/lib/halo2_kzg.ak:15-171
 *   lagrange        /root/reference/aiken-verifier/aiken_halo2/lib/lagrange.ak:40-130
 */
#include "h2v_oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include "curve.h"
#include "transcript.h"

/* ------------------------------------------------------------------ expressions */
enum { EX_CONST = 0, EX_FIXED = 1, EX_ADVICE = 2, EX_NEG = 3, EX_SUM = 4, EX_PROD = 5, EX_SCALED = 6 };
typedef struct { int tag, idx, a, b; fr c; } enode;
typedef struct { enode *n; int cnt, cap; } epool;

typedef struct { const uint8_t *p; size_t len, pos; int err; } rd;
static uint32_t rd_u32(rd *r) {
    if (r->pos + 4 > r->len) { r->err = 1; return 0; }
    uint32_t v = (uint32_t)r->p[r->pos] | ((uint32_t)r->p[r->pos + 1] << 8) | ((uint32_t)r->p[r->pos + 2] << 16) | ((uint32_t)r->p[r->pos + 3] << 24);
    r->pos += 4;
    return v;
}
static uint8_t rd_u8(rd *r) {
    if (r->pos + 1 > r->len) { r->err = 1; return 0; }
    return r->p[r->pos++];
}
static const uint8_t *rd_bytes(rd *r, size_t n) {
    if (r->pos + n > r->len) { r->err = 1; return 0; }
    const uint8_t *p = r->p + r->pos;
    r->pos += n;
    return p;
}
static int parse_expr(rd *r, epool *pool, int depth) {
    if (r->err || depth > 4096) { r->err = 1; return -1; }
    if (pool->cnt == pool->cap) {
        pool->cap = pool->cap ? pool->cap * 2 : 64;
        pool->n = (enode *)realloc(pool->n, sizeof(enode) * pool->cap);
    }
    int id = pool->cnt++;
    enode e;
    memset(&e, 0, sizeof e);
    e.tag = rd_u8(r);
    e.a = e.b = -1;
    switch (e.tag) {
    case EX_CONST: { const uint8_t *b = rd_bytes(r, 32); if (b) fr_from_le32(&e.c, b); break; }
    case EX_FIXED: case EX_ADVICE: e.idx = (int)rd_u32(r); break;
    case EX_NEG: e.a = parse_expr(r, pool, depth + 1); break;
    case EX_SUM: case EX_PROD: e.a = parse_expr(r, pool, depth + 1); e.b = parse_expr(r, pool, depth + 1); break;
    case EX_SCALED: { e.a = parse_expr(r, pool, depth + 1); const uint8_t *b = rd_bytes(r, 32); if (b) fr_from_le32(&e.c, b); break; }
    default: r->err = 1;
    }
    pool->n[id] = e;
    return id;
}
/* Expression<Scalar> semantics, src/plutus_gen/extraction/data/languages/aiken.rs:122-182 */
static void eval_expr(const enode *n, int id, const fr *adv, const fr *fix, fr *out) {
    const enode *e = &n[id];
    fr a, b;
    switch (e->tag) {
    case EX_CONST: *out = e->c; break;
    case EX_FIXED: *out = fix[e->idx]; break;
    case EX_ADVICE: *out = adv[e->idx]; break;
    case EX_NEG: eval_expr(n, e->a, adv, fix, &a); fr_neg(out, &a); break;
    case EX_SUM: eval_expr(n, e->a, adv, fix, &a); eval_expr(n, e->b, adv, fix, &b); fr_add(out, &a, &b); break;
    case EX_PROD: eval_expr(n, e->a, adv, fix, &a); eval_expr(n, e->b, adv, fix, &b); fr_mul(out, &a, &b); break;
    case EX_SCALED: eval_expr(n, e->a, adv, fix, &a); fr_mul(out, &a, &e->c); break;
    }
}
static int expr_max_idx(const enode *n, int id, int tag) {
    const enode *e = &n[id];
    int m = (e->tag == tag) ? e->idx : -1;
    if (e->a >= 0) { int x = expr_max_idx(n, e->a, tag); if (x > m) m = x; }
    if (e->b >= 0) { int x = expr_max_idx(n, e->b, tag); if (x > m) m = x; }
    return m;
}

/* ------------------------------------------------------------------ queries / point sets */
enum { CK_INSTANCE, CK_ADVICE, CK_FIXED, CK_PERM, CK_LOOKUP, CK_PERM_INPUT, CK_PERM_TABLE, CK_COMMON, CK_VANISH_G, CK_VANISH_RAND, CK_TRASH };
enum { EK_INSTANCE, EK_ADVICE, EK_FIXED, EK_PERM, EK_LOOKUP, EK_LOOKUP_NEXT, EK_PERM_INPUT, EK_PERM_INPUT_INV, EK_PERM_TABLE, EK_COMMON, EK_VANISH_S, EK_RANDOM, EK_TRASH };
/* RotationDescription with its derive(Ord): Last < Previous < Current < Next < Custom(n)
 * (src/plutus_gen/extraction/data/base_types/rotation_description.rs:14-24) */
enum { ROT_LAST = 0, ROT_PREV = 1, ROT_CUR = 2, ROT_NEXT = 3, ROT_CUSTOM = 4 };
typedef struct { int kind, n; } rot;
static rot rot_from_i32(int v) {
    rot r = {ROT_CUSTOM, v};
    if (v == -1) { r.kind = ROT_PREV; r.n = 0; }
    else if (v == 0) { r.kind = ROT_CUR; r.n = 0; }
    else if (v == 1) { r.kind = ROT_NEXT; r.n = 0; }
    return r;
}
static int rot_cmp(rot a, rot b) {
    if (a.kind != b.kind) return a.kind < b.kind ? -1 : 1;
    if (a.n != b.n) return a.n < b.n ? -1 : 1;
    return 0;
}
typedef struct { int ck, cidx, ek, eidx, esub; rot pt; } query;
typedef struct { int ck, cidx; int set; int npts; rot pts[8]; int ek[8], eidx[8], esub[8]; } commitment_data;
#define MAX_SET_PTS 8
typedef struct { int npts; rot pts[MAX_SET_PTS]; } point_set;

struct orc_vk {
    uint32_t k, bf, degree;
    fr transcript_repr, omega, omega_inv, bary;
    uint32_t n_adv_cols, n_fix_cols;
    uint32_t n_aq, n_fq, n_iq;
    int (*aq)[2], (*fq)[2], (*iq)[2];
    epool pool;
    uint32_t n_gates; int *gates;
    uint32_t n_lookups; int **lk_in, **lk_tab; uint32_t *lk_nin, *lk_ntab;
    uint32_t n_trash; int *tr_sel; int **tr_exprs; uint32_t *tr_n;
    uint32_t n_perm_cols; int (*perm_cols)[2];
    uint32_t n_fixed_comm, n_perm_comm;
    g1a *fixed_comm, *perm_comm;
    g2prep s_g2, g2gen;
    uint32_t n_pi, n_ci;
    /* recursion (IVC): fixed bases of the accumulator, emitters/aiken.rs:659-694; has_rec = 0 for plain circuits */
    int has_rec;
    uint32_t n_rec_bases;
    g1a *rec_bases;
    /* multi-phase circuits: phase of every advice column / challenge (extraction_steps/proof.rs:22-46); NULL = one phase */
    uint8_t *adv_phase, *chal_phase;
    uint32_t n_challenges, max_phase;
    /* derived */
    uint32_t chunk_len, n_chunks, n_splits;
    int n_queries; query *queries;
    int n_comm; commitment_data *cd;
    int n_sets; point_set *sets; /* unique, first-seen order */
    int *sort_order;             /* sorted set s -> old index */
    size_t proof_len;
};

static void add_query(orc_vk *vk, int ck, int cidx, int ek, int eidx, int esub, rot pt) {
    query q = {ck, cidx, ek, eidx, esub, pt};
    vk->queries = (query *)realloc(vk->queries, sizeof(query) * (vk->n_queries + 1));
    vk->queries[vk->n_queries++] = q;
}
static int find_query(int (*qs)[2], uint32_t n, int col, int rotv) {
    for (uint32_t i = 0; i < n; i++) if (qs[i][0] == col && qs[i][1] == rotv) return (int)i;
    return -1;
}

/* src/plutus_gen/extraction/mod.rs:120-226 + circuit_queries.rs:30-41 (order actually coded: advice, instance,
 * permutation, lookup, trashcan, fixed, common, vanishing) + pcs/mod.rs:36-109 */
static int build_sets(orc_vk *vk) {
    uint32_t i;
    for (i = 0; i < vk->n_aq; i++) add_query(vk, CK_ADVICE, vk->aq[i][0], EK_ADVICE, (int)i, 0, rot_from_i32(vk->aq[i][1]));
    for (i = 0; i < vk->n_iq; i++)
        if ((uint32_t)vk->iq[i][0] < vk->n_ci) add_query(vk, CK_INSTANCE, vk->iq[i][0], EK_INSTANCE, (int)i, 0, rot_from_i32(vk->iq[i][1]));
    rot cur = {ROT_CUR, 0}, next = {ROT_NEXT, 0}, prev = {ROT_PREV, 0}, last = {ROT_LAST, 0};
    for (i = 0; i < vk->n_chunks; i++) {
        add_query(vk, CK_PERM, (int)i, EK_PERM, (int)i, 1, cur);
        add_query(vk, CK_PERM, (int)i, EK_PERM, (int)i, 2, next);
    }
    for (int s = (int)vk->n_chunks - 2; s >= 0; s--) add_query(vk, CK_PERM, s, EK_PERM, s, 3, last);
    for (i = 0; i < vk->n_lookups; i++) {
        add_query(vk, CK_LOOKUP, (int)i, EK_LOOKUP, (int)i, 0, cur);
        add_query(vk, CK_PERM_INPUT, (int)i, EK_PERM_INPUT, (int)i, 0, cur);
        add_query(vk, CK_PERM_TABLE, (int)i, EK_PERM_TABLE, (int)i, 0, cur);
        add_query(vk, CK_PERM_INPUT, (int)i, EK_PERM_INPUT_INV, (int)i, 0, prev);
        add_query(vk, CK_LOOKUP, (int)i, EK_LOOKUP_NEXT, (int)i, 0, next);
    }
    for (i = 0; i < vk->n_trash; i++) add_query(vk, CK_TRASH, (int)i, EK_TRASH, (int)i, 0, cur);
    for (i = 0; i < vk->n_fq; i++) add_query(vk, CK_FIXED, vk->fq[i][0], EK_FIXED, (int)i, 0, rot_from_i32(vk->fq[i][1]));
    for (i = 0; i < vk->n_perm_comm; i++) add_query(vk, CK_COMMON, (int)i, EK_COMMON, (int)i, 0, cur);
    add_query(vk, CK_VANISH_G, 0, EK_VANISH_S, 0, 0, cur);
    add_query(vk, CK_VANISH_RAND, 0, EK_RANDOM, 0, 0, cur);

    /* unique commitments in first-seen order; their (point, eval) pairs sorted by point */
    vk->cd = (commitment_data *)calloc(vk->n_queries, sizeof(commitment_data));
    vk->n_comm = 0;
    for (int q = 0; q < vk->n_queries; q++) {
        query *Q = &vk->queries[q];
        int c;
        for (c = 0; c < vk->n_comm; c++) if (vk->cd[c].ck == Q->ck && vk->cd[c].cidx == Q->cidx) break;
        if (c == vk->n_comm) { vk->cd[c].ck = Q->ck; vk->cd[c].cidx = Q->cidx; vk->cd[c].npts = 0; vk->n_comm++; }
        commitment_data *C = &vk->cd[c];
        if (C->npts >= MAX_SET_PTS) return 0;
        /* insertion keeps the list sorted by point (stable) */
        int pos = C->npts;
        while (pos > 0 && rot_cmp(Q->pt, C->pts[pos - 1]) < 0) {
            C->pts[pos] = C->pts[pos - 1]; C->ek[pos] = C->ek[pos - 1]; C->eidx[pos] = C->eidx[pos - 1]; C->esub[pos] = C->esub[pos - 1];
            pos--;
        }
        C->pts[pos] = Q->pt; C->ek[pos] = Q->ek; C->eidx[pos] = Q->eidx; C->esub[pos] = Q->esub;
        C->npts++;
    }
    /* unique point sets in first-seen (commitment) order */
    vk->sets = (point_set *)calloc(vk->n_comm, sizeof(point_set));
    vk->n_sets = 0;
    for (int c = 0; c < vk->n_comm; c++) {
        commitment_data *C = &vk->cd[c];
        int s;
        for (s = 0; s < vk->n_sets; s++) {
            if (vk->sets[s].npts != C->npts) continue;
            int same = 1;
            for (int j = 0; j < C->npts; j++) if (rot_cmp(vk->sets[s].pts[j], C->pts[j]) != 0) same = 0;
            if (same) break;
        }
        if (s == vk->n_sets) {
            vk->sets[s].npts = C->npts;
            memcpy(vk->sets[s].pts, C->pts, sizeof(rot) * C->npts);
            vk->n_sets++;
        }
        C->set = s;
    }
    /* sort point sets by (cardinality, first-seen index): emitters/aiken.rs:580-587 */
    vk->sort_order = (int *)malloc(sizeof(int) * vk->n_sets);
    for (int s = 0; s < vk->n_sets; s++) vk->sort_order[s] = s;
    for (int a = 1; a < vk->n_sets; a++) {
        int v = vk->sort_order[a], b = a;
        while (b > 0 && (vk->sets[vk->sort_order[b - 1]].npts > vk->sets[v].npts)) { vk->sort_order[b] = vk->sort_order[b - 1]; b--; }
        vk->sort_order[b] = v;
    }
    return 1;
}

static int read_queries(rd *r, uint32_t *n, int (**out)[2]) {
    *n = rd_u32(r);
    if (r->err || *n > 4096) return 0;
    *out = (int(*)[2])calloc(*n ? *n : 1, sizeof(int[2]));
    for (uint32_t i = 0; i < *n; i++) { (*out)[i][0] = (int)rd_u32(r); (*out)[i][1] = (int)rd_u32(r); }
    return !r->err;
}
static int *read_exprs(rd *r, epool *pool, uint32_t *n) {
    *n = rd_u32(r);
    if (r->err || *n > 65536) { r->err = 1; return 0; }
    int *ids = (int *)calloc(*n ? *n : 1, sizeof(int));
    for (uint32_t i = 0; i < *n; i++) ids[i] = parse_expr(r, pool, 0);
    return ids;
}

orc_vk *orc_vk_parse(const uint8_t *desc, size_t len) {
    rd r = {desc, len, 0, 0};
    const uint8_t *magic = rd_bytes(&r, 8);
    if (!magic || memcmp(magic, "ORCVK001", 8) != 0) return 0;
    orc_vk *vk = (orc_vk *)calloc(1, sizeof(orc_vk));
    vk->k = rd_u32(&r); vk->bf = rd_u32(&r); vk->degree = rd_u32(&r);
    const uint8_t *b;
    if ((b = rd_bytes(&r, 32))) fr_from_le32(&vk->transcript_repr, b);
    if ((b = rd_bytes(&r, 32))) fr_from_le32(&vk->omega, b);
    if ((b = rd_bytes(&r, 32))) fr_from_le32(&vk->omega_inv, b);
    if ((b = rd_bytes(&r, 32))) fr_from_le32(&vk->bary, b);
    vk->n_adv_cols = rd_u32(&r); vk->n_fix_cols = rd_u32(&r);
    if (!read_queries(&r, &vk->n_aq, &vk->aq) || !read_queries(&r, &vk->n_fq, &vk->fq) || !read_queries(&r, &vk->n_iq, &vk->iq)) goto fail;
    vk->gates = read_exprs(&r, &vk->pool, &vk->n_gates);
    vk->n_lookups = rd_u32(&r);
    if (r.err || vk->n_lookups > 1024) goto fail;
    vk->lk_in = (int **)calloc(vk->n_lookups + 1, sizeof(int *)); vk->lk_tab = (int **)calloc(vk->n_lookups + 1, sizeof(int *));
    vk->lk_nin = (uint32_t *)calloc(vk->n_lookups + 1, 4); vk->lk_ntab = (uint32_t *)calloc(vk->n_lookups + 1, 4);
    for (uint32_t i = 0; i < vk->n_lookups; i++) {
        vk->lk_in[i] = read_exprs(&r, &vk->pool, &vk->lk_nin[i]);
        vk->lk_tab[i] = read_exprs(&r, &vk->pool, &vk->lk_ntab[i]);
    }
    vk->n_trash = rd_u32(&r);
    if (r.err || vk->n_trash > 1024) goto fail;
    vk->tr_sel = (int *)calloc(vk->n_trash + 1, sizeof(int)); vk->tr_exprs = (int **)calloc(vk->n_trash + 1, sizeof(int *)); vk->tr_n = (uint32_t *)calloc(vk->n_trash + 1, 4);
    for (uint32_t i = 0; i < vk->n_trash; i++) {
        vk->tr_sel[i] = parse_expr(&r, &vk->pool, 0);
        vk->tr_exprs[i] = read_exprs(&r, &vk->pool, &vk->tr_n[i]);
    }
    vk->n_perm_cols = rd_u32(&r);
    if (r.err || vk->n_perm_cols > 4096) goto fail;
    vk->perm_cols = (int(*)[2])calloc(vk->n_perm_cols + 1, sizeof(int[2]));
    for (uint32_t i = 0; i < vk->n_perm_cols; i++) { vk->perm_cols[i][0] = rd_u8(&r); vk->perm_cols[i][1] = (int)rd_u32(&r); }
    vk->n_fixed_comm = rd_u32(&r);
    if (r.err || vk->n_fixed_comm > 4096) goto fail;
    vk->fixed_comm = (g1a *)calloc(vk->n_fixed_comm + 1, sizeof(g1a));
    for (uint32_t i = 0; i < vk->n_fixed_comm; i++) { b = rd_bytes(&r, 48); if (!b || !g1_decompress(&vk->fixed_comm[i], b)) goto fail; }
    vk->n_perm_comm = rd_u32(&r);
    if (r.err || vk->n_perm_comm > 4096) goto fail;
    vk->perm_comm = (g1a *)calloc(vk->n_perm_comm + 1, sizeof(g1a));
    for (uint32_t i = 0; i < vk->n_perm_comm; i++) { b = rd_bytes(&r, 48); if (!b || !g1_decompress(&vk->perm_comm[i], b)) goto fail; }
    {
        g2a sg2, gen;
        b = rd_bytes(&r, 96);
        if (!b || !g2_decompress(&sg2, b) || !g2_prepare(&vk->s_g2, &sg2)) goto fail;
        fp_set(&gen.x.c0, G2_GEN_X0); fp_set(&gen.x.c1, G2_GEN_X1); fp_set(&gen.y.c0, G2_GEN_Y0); fp_set(&gen.y.c1, G2_GEN_Y1); gen.inf = 0;
        if (!g2a_on_curve(&gen) || !g2_prepare(&vk->g2gen, &gen)) goto fail;
    }
    vk->n_pi = rd_u32(&r); vk->n_ci = rd_u32(&r);
    if (r.err || vk->n_ci > 1 || vk->degree < 3) goto fail;
    if (r.pos < r.len) { /* optional recursion section */
        vk->has_rec = (int)rd_u32(&r);
        vk->n_rec_bases = rd_u32(&r);
        if (r.err || vk->has_rec > 1 || vk->n_rec_bases > 4096) goto fail;
        vk->rec_bases = (g1a *)calloc(vk->n_rec_bases + 1, sizeof(g1a));
        for (uint32_t i = 0; i < vk->n_rec_bases; i++) { b = rd_bytes(&r, 48); if (!b || !g1_decompress(&vk->rec_bases[i], b)) goto fail; }
        /* not enough public inputs to support recursion: aiken.rs:702 (nb_vks + F + 10; nb_vks >= 1) */
        if (vk->has_rec && vk->n_pi < 1 + vk->n_rec_bases + 10) goto fail;
    }
    if (r.pos < r.len) { /* optional phase section */
        uint32_t na = rd_u32(&r);
        if (r.err || na != vk->n_adv_cols) goto fail;
        b = rd_bytes(&r, na);
        if (!b && na) goto fail;
        vk->adv_phase = (uint8_t *)calloc(na + 1, 1);
        for (uint32_t i = 0; i < na; i++) { vk->adv_phase[i] = b[i]; if (b[i] > vk->max_phase) vk->max_phase = b[i]; }
        vk->n_challenges = rd_u32(&r);
        if (r.err || vk->n_challenges > 4096) goto fail;
        b = rd_bytes(&r, vk->n_challenges);
        if (!b && vk->n_challenges) goto fail;
        vk->chal_phase = (uint8_t *)calloc(vk->n_challenges + 1, 1);
        for (uint32_t i = 0; i < vk->n_challenges; i++) vk->chal_phase[i] = b[i];
    }
    if (vk->n_perm_comm != vk->n_perm_cols) goto fail;
    /* every expression index must exist */
    for (int i = 0; i < vk->pool.cnt; i++) {
        enode *e = &vk->pool.n[i];
        if (e->tag == EX_FIXED && (uint32_t)e->idx >= vk->n_fq) goto fail;
        if (e->tag == EX_ADVICE && (uint32_t)e->idx >= vk->n_aq) goto fail;
    }
    (void)expr_max_idx;
    vk->chunk_len = vk->degree - 2;                                          /* extraction/mod.rs:57 */
    vk->n_chunks = (vk->n_perm_cols + vk->chunk_len - 1) / vk->chunk_len;    /* proof.rs:61 */
    vk->n_splits = vk->degree - 1;                                           /* get_quotient_poly_degree, proof.rs:78 */
    if (vk->n_perm_cols == 0 || vk->n_chunks > 26) goto fail;
    for (uint32_t i = 0; i < vk->n_perm_cols; i++) {
        int t = vk->perm_cols[i][0], c = vk->perm_cols[i][1], qi;
        if (t == 0) qi = find_query(vk->aq, vk->n_aq, c, 0);
        else if (t == 1) qi = find_query(vk->fq, vk->n_fq, c, 0);
        else qi = find_query(vk->iq, vk->n_iq, c, 0);
        if (qi < 0) goto fail; /* get_any_query_index panics: permutation.rs:13-78 */
    }
    if (!build_sets(vk)) goto fail;
    {
        /* proof length: A.1 of SURVEY / proof.rs:13-143 / pcs/kzg.rs:55-79 */
        size_t g = vk->n_adv_cols + 3 * vk->n_lookups + vk->n_chunks + vk->n_trash + 1 + vk->n_splits + 2;
        size_t s = vk->n_aq + vk->n_fq + 1 + vk->n_perm_comm + (3 * vk->n_chunks - 1) + 5 * vk->n_lookups + vk->n_trash + (size_t)vk->n_sets;
        for (uint32_t i = 0; i < vk->n_iq; i++) if ((uint32_t)vk->iq[i][0] < vk->n_ci) s++;
        vk->proof_len = 48 * g + 32 * s;
    }
    return vk;
fail:
    orc_vk_free(vk);
    return 0;
}
void orc_vk_free(orc_vk *vk) {
    if (!vk) return;
    free(vk->aq); free(vk->fq); free(vk->iq); free(vk->pool.n); free(vk->gates);
    for (uint32_t i = 0; i < vk->n_lookups; i++) { if (vk->lk_in) free(vk->lk_in[i]); if (vk->lk_tab) free(vk->lk_tab[i]); }
    free(vk->lk_in); free(vk->lk_tab); free(vk->lk_nin); free(vk->lk_ntab);
    for (uint32_t i = 0; i < vk->n_trash; i++) if (vk->tr_exprs) free(vk->tr_exprs[i]);
    free(vk->tr_sel); free(vk->tr_exprs); free(vk->tr_n);
    free(vk->perm_cols); free(vk->fixed_comm); free(vk->perm_comm); free(vk->rec_bases); free(vk->adv_phase); free(vk->chal_phase);
    free(vk->queries); free(vk->cd); free(vk->sets); free(vk->sort_order);
    free(vk);
}
int orc_vk_num_point_sets(const orc_vk *vk) { return vk->n_sets; }
int orc_vk_num_msm_terms(const orc_vk *vk) { return (vk->n_comm - 1) + (int)vk->n_splits + 3; }
size_t orc_vk_proof_len(const orc_vk *vk) { return vk->proof_len; }

/* ------------------------------------------------------------------ lagrange.ak */
/* batch_inverses (lagrange.ak:98-130): Montgomery trick; a zero anywhere makes the single recip_eea fail. */
static int batch_inverses(fr *out, const fr *in, int n) {
    if (n == 0) return 1;
    fr *pre = (fr *)malloc(sizeof(fr) * n);
    pre[0] = in[0];
    for (int i = 1; i < n; i++) fr_mul(&pre[i], &pre[i - 1], &in[i]);
    fr inv;
    if (!fr_inv(&inv, &pre[n - 1])) { free(pre); return 0; }
    for (int i = n - 1; i > 0; i--) {
        fr a_i = in[i]; /* in may alias out */
        fr_mul(&out[i], &inv, &pre[i - 1]);
        fr_mul(&inv, &inv, &a_i);
    }
    out[0] = inv;
    free(pre);
    return 1;
}
/* rotate_omega (omega_rotations.ak:19-30): omega^rotation * value */
static void rotate_omega(fr *out, const fr *omega, const fr *omega_inv, const fr *value, int rotation) {
    fr p;
    if (rotation < 0) fr_pow_u64(&p, omega_inv, (uint64_t)(-(int64_t)rotation));
    else fr_pow_u64(&p, omega, (uint64_t)rotation);
    fr_mul(out, &p, value);
}
/* lagrange_polynomial_basis (lagrange.ak:79-96): l_i(x) = omega^i (x^n-1) n^-1 / (x - omega^i) */
static int lagrange_basis(fr *out, const fr *x, const fr *xn, const fr *bary, const fr *rotations, int n) {
    fr one, common;
    fr_one(&one);
    fr_sub(&common, xn, &one);
    fr_mul(&common, &common, bary);
    fr *d = (fr *)malloc(sizeof(fr) * (n ? n : 1));
    for (int i = 0; i < n; i++) fr_sub(&d[i], x, &rotations[i]);
    int ok = batch_inverses(d, d, n);
    if (ok)
        for (int i = 0; i < n; i++) { fr_mul(&out[i], &d[i], &common); fr_mul(&out[i], &out[i], &rotations[i]); }
    free(d);
    return ok;
}
/* lagrange_evaluation (lagrange.ak:40-77): interpolate (xi, yi), evaluate at x */
static int lagrange_evaluation(fr *out, const fr *xs, const fr *ys, int n, const fr *x) {
    fr num[MAX_SET_PTS + 8], den[MAX_SET_PTS + 8];
    if (n > MAX_SET_PTS + 8) return 0;
    for (int i = 0; i < n; i++) {
        fr_one(&num[i]); fr_one(&den[i]);
        for (int j = 0; j < n; j++) {
            if (fr_eq(&xs[j], &xs[i])) continue;
            fr t;
            fr_sub(&t, x, &xs[j]); fr_mul(&num[i], &num[i], &t);
            fr_sub(&t, &xs[i], &xs[j]); fr_mul(&den[i], &den[i], &t);
        }
    }
    if (!batch_inverses(den, den, n)) return 0;
    fr acc;
    fr_zero(&acc);
    for (int i = 0; i < n; i++) {
        fr t;
        fr_mul(&t, &num[i], &den[i]); fr_mul(&t, &ys[i], &t); fr_add(&acc, &acc, &t);
    }
    *out = acc;
    return 1;
}

/* ------------------------------------------------------------------ halo2_kzg.ak scalar part */
/* compute_f_eval (halo2_kzg.ak:121-159) and compute_v (:161-171).  q_eval_sets: per set, per point. */
static int multiopen_f_v(int n_sets, const int *set_sizes, fr **points, fr **q_eval_sets, const fr *x2, const fr *x3,
                         const fr *x4, const fr *q_evals, fr *f_eval, fr *v) {
    fr *r_eval = (fr *)malloc(sizeof(fr) * n_sets), *den = (fr *)malloc(sizeof(fr) * n_sets);
    int ok = 1;
    for (int s = 0; s < n_sets && ok; s++) {
        ok = lagrange_evaluation(&r_eval[s], points[s], q_eval_sets[s], set_sizes[s], x3);
        fr_one(&den[s]);
        for (int j = 0; j < set_sizes[s]; j++) { fr t; fr_sub(&t, x3, &points[s][j]); fr_mul(&den[s], &den[s], &t); }
    }
    if (ok) ok = batch_inverses(den, den, n_sets);
    if (ok) {
        fr acc;
        fr_zero(&acc);
        for (int s = n_sets - 1; s >= 0; s--) { /* foldl over the reversed list */
            fr e;
            fr_sub(&e, &q_evals[s], &r_eval[s]); fr_mul(&e, &e, &den[s]);
            fr_mul(&acc, &acc, x2); fr_add(&acc, &acc, &e);
        }
        *f_eval = acc;
        fr p, vv;
        fr_one(&p); fr_zero(&vv);
        for (int s = 0; s <= n_sets; s++) {
            fr t;
            fr_mul(&t, &p, s < n_sets ? &q_evals[s] : f_eval); fr_add(&vv, &vv, &t);
            fr_mul(&p, &p, x4);
        }
        *v = vv;
    }
    free(r_eval); free(den);
    return ok;
}

/* ------------------------------------------------------------------ the verifier */
static void put_fr(uint8_t *dst, const fr *a) { fr_to_le32(dst, a); }
static void put_g1(uint8_t *dst, const g1a *a) {
    if (a->inf) { memset(dst, 0, 96); return; }
    fp_to_be48(dst, &a->x); fp_to_be48(dst + 48, &a->y);
}
static void g1_scale_add(g1j *acc, const g1a *p, const fr *k) { /* acc += k*P */
    g1j pj, t;
    g1j_from_affine(&pj, p);
    g1j_mul_fr(&t, &pj, k);
    g1j_add(acc, acc, &t);
}

/* The five identities of one lookup argument (aiken.rs:264-330) from the already compressed input / table values and
 * the argument's five evaluations ev = {product, product_next, permuted_input, permuted_input_inv, permuted_table}.
 * orc_verify and the golden-vector entry point orc_lookup_identities both run THIS code. */
static void lookup_identities(fr out[5], const fr *l_0, const fr *l_last, const fr *active_rows, const fr *beta,
                              const fr *gamma, const fr *inp, const fr *tab, const fr ev[5]) {
    const fr *prod = &ev[0], *prod_next = &ev[1], *pin = &ev[2], *pinv = &ev[3], *ptab = &ev[4];
    fr one, t, u, left, right;
    fr_one(&one);
    fr_sub(&t, &one, prod); fr_mul(&out[0], l_0, &t);
    fr_mul(&t, prod, prod); fr_sub(&t, &t, prod); fr_mul(&out[1], l_last, &t);
    fr_add(&t, pin, beta); fr_mul(&left, prod_next, &t); fr_add(&t, ptab, gamma); fr_mul(&left, &left, &t);
    fr_add(&t, inp, beta); fr_mul(&right, prod, &t); fr_add(&t, tab, gamma); fr_mul(&right, &right, &t);
    fr_sub(&t, &left, &right); fr_mul(&out[2], &t, active_rows);
    fr_sub(&t, pin, ptab); fr_mul(&out[3], l_0, &t);
    fr_sub(&u, pin, pinv); fr_mul(&t, &t, &u); fr_mul(&out[4], &t, active_rows);
}
/* One lookup argument: theta-compression of its input / table expression lists (Horner, languages/aiken.rs:18-29)
 * followed by the five identities.  Shared by orc_verify and orc_lookup_argument (golden vectors of gates_test.hbs). */
static void lookup_argument(fr out[5], const enode *pool, const int *in_ids, uint32_t n_in, const int *tab_ids, uint32_t n_tab,
                            const fr *advice_eval, const fr *fixed_eval, const fr *theta, const fr *beta, const fr *gamma,
                            const fr *l_0, const fr *l_last, const fr *active_rows, const fr ev[5]) {
    fr tab, inp, e;
    fr_zero(&tab); fr_zero(&inp);
    for (uint32_t j = 0; j < n_tab; j++) { eval_expr(pool, tab_ids[j], advice_eval, fixed_eval, &e); fr_mul(&tab, &tab, theta); fr_add(&tab, &tab, &e); }
    for (uint32_t j = 0; j < n_in; j++) { eval_expr(pool, in_ids[j], advice_eval, fixed_eval, &e); fr_mul(&inp, &inp, theta); fr_add(&inp, &inp, &e); }
    lookup_identities(out, l_0, l_last, active_rows, beta, gamma, &inp, &tab, ev);
}
/* One commitment's contribution to the q_eval set of its point set (compute_q_evals_and_final_comm,
 * halo2_kzg.ak:46-89): q[j] += x1^pos * eval_j.  Shared by orc_verify and orc_multiopen_scalars. */
static void q_eval_accumulate(fr *q, const fr *const *evs, int npts, const fr *x1p) {
    for (int j = 0; j < npts; j++) {
        fr t;
        fr_mul(&t, evs[j], x1p);
        fr_add(&q[j], &q[j], &t);
    }
}

#define REJECT(code) do { status = (code); goto done; } while (0)

int orc_verify(const orc_vk *vk, const uint8_t *proof, size_t proof_len, const uint8_t *instances,
               const uint8_t *committed, orc_trace *trace) {
    int status = ORC_REJ_PAIRING;
    uint32_t i, L = vk->n_lookups, C = vk->n_chunks;
    int canonical, bad_scalar = 0;
    const uint8_t *pb;
    fr one, zero;
    fr_one(&one); fr_zero(&zero);
    if (trace) memset(trace, 0, sizeof *trace);

    /* scratch */
    size_t nfr = vk->n_aq + vk->n_fq + vk->n_iq + vk->n_perm_comm + 3 * C + 5 * L + vk->n_trash + vk->n_pi + 64;
    fr *mem = (fr *)calloc(nfr + 4 * (vk->n_pi + vk->bf + 8), sizeof(fr));
    fr *advice_eval = mem, *fixed_eval = advice_eval + vk->n_aq, *instance_eval = fixed_eval + vk->n_fq;
    fr *perm_common = instance_eval + vk->n_iq, *perm_eval = perm_common + vk->n_perm_comm; /* [chunk*3 + (sub-1)] */
    fr *lk_eval = perm_eval + 3 * C; /* [l*5 + {product, product_next, permuted_input, permuted_input_inv, permuted_table}] */
    fr *trash_eval = lk_eval + 5 * L, *pi = trash_eval + vk->n_trash, *tmp = pi + vk->n_pi + 8;
    size_t npts = vk->n_adv_cols + 3 * L + C + vk->n_trash + 1 + vk->n_splits + 2 + 1;
    g1a *pts = (g1a *)calloc(npts, sizeof(g1a));
    g1a *adv_c = pts, *lk_pin = adv_c + vk->n_adv_cols, *lk_ptab = lk_pin + L, *perm_c = lk_ptab + L, *lk_prod = perm_c + C;
    g1a *trash_c = lk_prod + L, *vanish_rand = trash_c + vk->n_trash, *splits = vanish_rand + 1, *f_comm = splits + vk->n_splits;
    g1a *pi_pt = f_comm + 1, *ci_pt = pi_pt + 1;
    fr *expr = (fr *)calloc(ORC_MAX_EXPR + vk->n_gates + 5 * L + 3 * C + vk->n_trash + 8, sizeof(fr));
    int n_expr = 0;
    fr **set_pts = 0, **q_eval_sets = 0; int *set_sizes = 0; fr *q_evals = 0;

    /* P1 transcript init + inputs (verification_h2.hbs:26-28; aiken.rs:44-84) */
    transcript tr;
    tr_init(&tr, proof, proof_len);
    tr_common_scalar(&tr, &vk->transcript_repr);
    if (vk->n_ci) {
        if (!committed) REJECT(ORC_REJ_POINT);
        tr_common_point(&tr, committed);
        if (!g1_decompress(ci_pt, committed)) REJECT(ORC_REJ_POINT);
    }
    {
        fr cnt;
        fr_from_u64(&cnt, vk->n_pi);
        tr_common_scalar(&tr, &cnt);
        for (i = 0; i < vk->n_pi; i++) {
            /* public inputs are field elements on the reference side (Rust F, Aiken State<Scalar>): a 32-byte value
             * >= r has no counterpart there, so it is rejected like a non-canonical proof scalar */
            if (!fr_from_le32(&pi[i], instances + 32 * i)) bad_scalar = 1;
            tr_common_scalar(&tr, &pi[i]);
        }
    }
#define RD_POINT(dst) do { pb = tr_read_point(&tr); if (!pb) REJECT(ORC_REJ_SHORT); if (!g1_decompress((dst), pb)) REJECT(ORC_REJ_POINT); } while (0)
#define RD_SCALAR(dst) do { if (!tr_read_scalar(&tr, (dst), &canonical)) REJECT(ORC_REJ_SHORT); if (!canonical) bad_scalar = 1; } while (0)
    /* P2 proof read / squeeze order: extraction_steps/proof.rs:13-143 */
    fr theta, beta, gamma, trash, y, x, x1, x2, x3, x4;
    /* per phase: the advice commitments of that phase, then one squeeze per challenge of that phase (proof.rs:22-46);
     * the challenge values are not used by any expression (languages/aiken.rs:150-156: a panic) */
    for (uint32_t phase = 0; phase <= vk->max_phase; phase++) {
        for (i = 0; i < vk->n_adv_cols; i++)
            if ((vk->adv_phase ? vk->adv_phase[i] : 0) == phase) RD_POINT(&adv_c[i]);
        for (i = 0; i < vk->n_challenges; i++)
            if (vk->chal_phase[i] == phase) { fr unused; tr_squeeze(&tr, &unused); }
    }
    tr_squeeze(&tr, &theta);
    for (i = 0; i < L; i++) { RD_POINT(&lk_pin[i]); RD_POINT(&lk_ptab[i]); }
    tr_squeeze(&tr, &beta);
    tr_squeeze(&tr, &gamma);
    for (i = 0; i < C; i++) RD_POINT(&perm_c[i]);
    for (i = 0; i < L; i++) RD_POINT(&lk_prod[i]);
    tr_squeeze(&tr, &trash);
    for (i = 0; i < vk->n_trash; i++) RD_POINT(&trash_c[i]);
    RD_POINT(vanish_rand);
    tr_squeeze(&tr, &y);
    for (i = 0; i < vk->n_splits; i++) RD_POINT(&splits[i]);
    tr_squeeze(&tr, &x);
    /* xn_minus_one = x^(n-1); xn = xn_minus_one * x   (aiken.rs:124-136) */
    fr xn_minus_one, xn;
    {
        /* x^(n-1) with n = 2^k: x^n / x needs an inverse; do plain square-and-multiply on n-1 = 2^k - 1 */
        fr acc = one;
        for (uint32_t bit = 0; bit < vk->k; bit++) { fr_sqr(&acc, &acc); fr_mul(&acc, &acc, &x); }
        xn_minus_one = acc;
        fr_mul(&xn, &xn_minus_one, &x);
    }
    /* instance evals (aiken.rs:200-223) */
    for (i = 0; i < vk->n_iq; i++) {
        if ((uint32_t)vk->iq[i][0] < vk->n_ci) {
            RD_SCALAR(&instance_eval[i]);
        } else if (vk->n_pi == 0) {
            instance_eval[i] = zero;
        } else {
            /* rotate_omegas(0..n_pi) has n_pi+1 entries; inner_product zips with the n_pi inputs */
            int n = (int)vk->n_pi + 1;
            fr *rots = tmp, *basis = tmp + n;
            for (int r = 0; r < n; r++) rotate_omega(&rots[r], &vk->omega, &vk->omega_inv, &one, r);
            if (!lagrange_basis(basis, &x, &xn, &vk->bary, rots, n)) REJECT(ORC_REJ_INVERSE);
            fr acc = zero;
            for (uint32_t r = 0; r < vk->n_pi; r++) { fr t; fr_mul(&t, &basis[r], &pi[r]); fr_add(&acc, &t, &acc); }
            instance_eval[i] = acc;
        }
    }
    for (i = 0; i < vk->n_aq; i++) RD_SCALAR(&advice_eval[i]);
    for (i = 0; i < vk->n_fq; i++) RD_SCALAR(&fixed_eval[i]);
    fr random_eval;
    RD_SCALAR(&random_eval);
    for (i = 0; i < vk->n_perm_comm; i++) RD_SCALAR(&perm_common[i]);
    for (i = 0; i < C; i++) {
        RD_SCALAR(&perm_eval[3 * i + 0]);
        RD_SCALAR(&perm_eval[3 * i + 1]);
        if (i != C - 1) RD_SCALAR(&perm_eval[3 * i + 2]);
    }
    for (i = 0; i < L; i++) {
        RD_SCALAR(&lk_eval[5 * i + 0]); /* product_eval */
        RD_SCALAR(&lk_eval[5 * i + 1]); /* product_next_eval */
        RD_SCALAR(&lk_eval[5 * i + 2]); /* permuted_input_eval */
        RD_SCALAR(&lk_eval[5 * i + 3]); /* permuted_input_inv_eval */
        RD_SCALAR(&lk_eval[5 * i + 4]); /* permuted_table_eval */
    }
    for (i = 0; i < vk->n_trash; i++) RD_SCALAR(&trash_eval[i]);
    /* PCS tail: pcs/kzg.rs:55-79 */
    int S = vk->n_sets;
    q_evals = (fr *)calloc(S + 1, sizeof(fr));
    tr_squeeze(&tr, &x1);
    tr_squeeze(&tr, &x2);
    RD_POINT(f_comm);
    tr_squeeze(&tr, &x3);
    for (int s = 0; s < S; s++) RD_SCALAR(&q_evals[s]);
    tr_squeeze(&tr, &x4);
    RD_POINT(pi_pt);
    if (bad_scalar) REJECT(ORC_REJ_SCALAR);

    /* P3 evaluation points (verification_h2.hbs:32-60) */
    fr x_last;
    rotate_omega(&x_last, &vk->omega, &vk->omega_inv, &x, -((int)vk->bf + 1));
    fr l_last, l_0, sum_blind = zero, active_rows;
    {
        int n = (int)vk->bf + 2;
        fr *rots = tmp, *basis = tmp + n;
        for (int r = 0; r < n; r++) rotate_omega(&rots[r], &vk->omega, &vk->omega_inv, &one, r - ((int)vk->bf + 1));
        if (!lagrange_basis(basis, &x, &xn, &vk->bary, rots, n)) REJECT(ORC_REJ_INVERSE);
        l_last = basis[0];
        for (uint32_t r = 0; r < vk->bf; r++) fr_add(&sum_blind, &basis[1 + r], &sum_blind);
        l_0 = basis[n - 1];
        fr t;
        fr_add(&t, &l_last, &sum_blind);
        fr_sub(&active_rows, &one, &t);
    }

    /* P4 combiner.  gates: aiken.rs:248-261 */
    for (i = 0; i < vk->n_gates; i++) eval_expr(vk->pool.n, vk->gates[i], advice_eval, fixed_eval, &expr[n_expr++]);
    /* permutation terms: extraction_steps/permutation.rs:80-143 */
    {
        fr t, u;
        fr_sub(&t, &one, &perm_eval[0]); fr_mul(&expr[n_expr++], &l_0, &t);
        const fr *zl = &perm_eval[3 * (C - 1)];
        fr_mul(&t, zl, zl); fr_sub(&t, &t, zl); fr_mul(&expr[n_expr++], &l_last, &t);
        for (i = 1; i < C; i++) {
            fr_sub(&u, &perm_eval[3 * i], &perm_eval[3 * (i - 1) + 2]);
            fr_mul(&expr[n_expr++], &u, &l_0);
        }
    }
    /* permutation sets: permutation.rs:145-303, aiken.rs:345-441 */
    for (i = 0; i < C; i++) {
        fr left = perm_eval[3 * i + 1], right = perm_eval[3 * i + 0], t, u, bx;
        fr delta; fr_set(&delta, FR_DELTA);
        fr_mul(&bx, &beta, &x);
        for (uint32_t idx = 0; idx < vk->chunk_len; idx++) {
            uint32_t col = i * vk->chunk_len + idx;
            if (col >= vk->n_perm_cols) break;
            int ty = vk->perm_cols[col][0], cc = vk->perm_cols[col][1];
            const fr *ev;
            if (ty == 0) ev = &advice_eval[find_query(vk->aq, vk->n_aq, cc, 0)];
            else if (ty == 1) ev = &fixed_eval[find_query(vk->fq, vk->n_fq, cc, 0)];
            else ev = &instance_eval[find_query(vk->iq, vk->n_iq, cc, 0)];
            /* left: eval + beta * permutation_common + gamma */
            fr_mul(&t, &beta, &perm_common[col]); fr_add(&t, ev, &t); fr_add(&t, &t, &gamma);
            fr_mul(&left, &left, &t);
            /* right: eval + (beta*x) * delta^power + gamma */
            fr_pow_u64(&u, &delta, col); fr_mul(&u, &bx, &u); fr_add(&u, ev, &u); fr_add(&u, &u, &gamma);
            fr_mul(&right, &right, &u);
        }
        fr_sub(&t, &left, &right);
        fr_add(&u, &l_last, &sum_blind); fr_sub(&u, &one, &u);
        fr_mul(&expr[n_expr++], &t, &u);
    }
    /* lookups: aiken.rs:264-330; compression languages/aiken.rs:18-29 */
    for (i = 0; i < L; i++) {
        lookup_argument(&expr[n_expr], vk->pool.n, vk->lk_in[i], vk->lk_nin[i], vk->lk_tab[i], vk->lk_ntab[i], advice_eval, fixed_eval,
                        &theta, &beta, &gamma, &l_0, &l_last, &active_rows, &lk_eval[5 * i]);
        n_expr += 5;
    }
    /* trashcans: aiken.rs:444-461 */
    for (i = 0; i < vk->n_trash; i++) {
        fr acc = zero, e, sel, t;
        for (uint32_t j = 0; j < vk->tr_n[i]; j++) { eval_expr(vk->pool.n, vk->tr_exprs[i][j], advice_eval, fixed_eval, &e); fr_mul(&acc, &acc, &trash); fr_add(&acc, &acc, &e); }
        eval_expr(vk->pool.n, vk->tr_sel[i], advice_eval, fixed_eval, &sel);
        fr_sub(&t, &one, &sel); fr_mul(&t, &t, &trash_eval[i]); fr_sub(&expr[n_expr++], &acc, &t);
    }
    /* hEval: Horner in y, acc0 = 0 (aiken.rs:557-563); vanishing_s (verification_h2.hbs:90-91) */
    fr h_eval = zero, vanishing_s;
    for (int e = 0; e < n_expr; e++) { fr_mul(&h_eval, &h_eval, &y); fr_add(&h_eval, &h_eval, &expr[e]); }
    {
        fr t;
        fr_sub(&t, &xn, &one);
        if (!fr_inv(&t, &t)) REJECT(ORC_REJ_INVERSE);
        fr_mul(&vanishing_s, &h_eval, &t);
    }
    /* P5 vanishing_g: Horner in x^(n-1), highest split first (extraction_steps/vanishing.rs:6-52) */
    g1j vg;
    g1j_set_inf(&vg);
    for (int s = (int)vk->n_splits - 1; s >= 0; s--) {
        g1j t, sp;
        g1j_mul_fr(&t, &vg, &xn_minus_one);
        g1j_from_affine(&sp, &splits[s]);
        g1j_add(&vg, &t, &sp);
    }
    g1a vanishing_g;
    g1j_to_affine(&vanishing_g, &vg);

    /* P6 multi-open (halo2_kzg.ak:15-44) over the sorted point sets */
    set_pts = (fr **)calloc(S, sizeof(fr *)); q_eval_sets = (fr **)calloc(S, sizeof(fr *)); set_sizes = (int *)calloc(S, sizeof(int));
    g1j final_com;
    g1j_set_inf(&final_com);
    fr f_eval, v;
    {
        fr x4p = one;
        for (int s = 0; s < S; s++) {
            int old = vk->sort_order[s];
            const point_set *ps = &vk->sets[old];
            set_sizes[s] = ps->npts;
            set_pts[s] = (fr *)calloc(ps->npts, sizeof(fr)); q_eval_sets[s] = (fr *)calloc(ps->npts, sizeof(fr));
            for (int j = 0; j < ps->npts; j++) {
                int rv = ps->pts[j].kind == ROT_LAST ? -((int)vk->bf + 1) : ps->pts[j].kind == ROT_PREV ? -1 : ps->pts[j].kind == ROT_CUR ? 0 : ps->pts[j].kind == ROT_NEXT ? 1 : ps->pts[j].n;
                rotate_omega(&set_pts[s][j], &vk->omega, &vk->omega_inv, &x, rv);
            }
            /* compute_q_evals_and_final_comm (halo2_kzg.ak:46-89) */
            g1j q_com;
            g1j_set_inf(&q_com);
            fr x1p = one;
            for (int c = 0; c < vk->n_comm; c++) {
                const commitment_data *cd = &vk->cd[c];
                if (cd->set != old) continue;
                const g1a *P;
                switch (cd->ck) {
                case CK_INSTANCE: P = ci_pt; break;
                case CK_ADVICE: P = &adv_c[cd->cidx]; break;
                case CK_FIXED: P = &vk->fixed_comm[cd->cidx]; break;
                case CK_PERM: P = &perm_c[cd->cidx]; break;
                case CK_LOOKUP: P = &lk_prod[cd->cidx]; break;
                case CK_PERM_INPUT: P = &lk_pin[cd->cidx]; break;
                case CK_PERM_TABLE: P = &lk_ptab[cd->cidx]; break;
                case CK_COMMON: P = &vk->perm_comm[cd->cidx]; break;
                case CK_VANISH_G: P = &vanishing_g; break;
                case CK_VANISH_RAND: P = vanish_rand; break;
                default: P = &trash_c[cd->cidx]; break;
                }
                g1_scale_add(&q_com, P, &x1p);
                const fr *evs[MAX_SET_PTS];
                for (int j = 0; j < cd->npts; j++) {
                    const fr *ev;
                    switch (cd->ek[j]) {
                    case EK_INSTANCE: ev = &instance_eval[cd->eidx[j]]; break;
                    case EK_ADVICE: ev = &advice_eval[cd->eidx[j]]; break;
                    case EK_FIXED: ev = &fixed_eval[cd->eidx[j]]; break;
                    case EK_PERM: ev = &perm_eval[3 * cd->eidx[j] + cd->esub[j] - 1]; break;
                    case EK_LOOKUP: ev = &lk_eval[5 * cd->eidx[j] + 0]; break;
                    case EK_LOOKUP_NEXT: ev = &lk_eval[5 * cd->eidx[j] + 1]; break;
                    case EK_PERM_INPUT: ev = &lk_eval[5 * cd->eidx[j] + 2]; break;
                    case EK_PERM_INPUT_INV: ev = &lk_eval[5 * cd->eidx[j] + 3]; break;
                    case EK_PERM_TABLE: ev = &lk_eval[5 * cd->eidx[j] + 4]; break;
                    case EK_COMMON: ev = &perm_common[cd->eidx[j]]; break;
                    case EK_VANISH_S: ev = &vanishing_s; break;
                    case EK_RANDOM: ev = &random_eval; break;
                    default: ev = &trash_eval[cd->eidx[j]]; break;
                    }
                    evs[j] = ev;
                }
                q_eval_accumulate(q_eval_sets[s], evs, cd->npts, &x1p);
                fr_mul(&x1p, &x1p, &x1);
            }
            g1j t;
            g1j_mul_fr(&t, &q_com, &x4p);
            g1j_add(&final_com, &final_com, &t);
            fr_mul(&x4p, &x4p, &x4);
        }
        g1_scale_add(&final_com, f_comm, &x4p); /* x4^S * f_commitment */
    }
    if (!multiopen_f_v(S, set_sizes, set_pts, q_eval_sets, &x2, &x3, &x4, q_evals, &f_eval, &v)) REJECT(ORC_REJ_INVERSE);
    /* er = final_com + v*(-G1) + x3*pi ; el = pi  (halo2_kzg.ak:37-43) */
    g1a neg_g1, el = *pi_pt, er;
    fp_set(&neg_g1.x, G1_GEN_X); fp_set(&neg_g1.y, G1_GEN_Y); fp_neg(&neg_g1.y, &neg_g1.y); neg_g1.inf = 0;
    {
        g1j acc;
        g1j_set_inf(&acc);
        g1_scale_add(&acc, &neg_g1, &v);
        g1_scale_add(&acc, pi_pt, &x3);
        g1j_add(&acc, &final_com, &acc);
        g1j_to_affine(&er, &acc);
    }
    /* Recursion (IVC): fold the collapsed accumulator carried by the public inputs into (el, er) before the pairing
     * (RECURSION_ACCUMULATOR block, verification_h2.hbs:123; emitters/aiken.rs:696-757; docs/algorithms.html "IVC") */
    if (vk->has_rec) {
        const uint32_t N = vk->n_pi, F = vk->n_rec_bases;
#define PI1(k) (&pi[(k) - 1]) /* the emitter's i_k are 1-based */
        if (!fr_eq(PI1(1), &vk->transcript_repr)) REJECT(ORC_REJ_RECURSION);       /* expect transcript_rep == i_1 */
        fp two224;
        { uint64_t t[6] = {0, 0, 0, 1ull << 32, 0, 0}; fp_from_plain(&two224, t); } /* batching_coeff = (2^56)^4 */
        g1a acc_pt[2];
        for (int side = 0; side < 2; side++) {
            /* left: x from i_{N-F-8}, i_{N-F-9}; y from i_{N-F-6}, i_{N-F-7}.  right: x i_{N-F-3}, i_{N-F-4}; y i_{N-F-1}, i_{N-F-2} */
            const uint32_t xh = side == 0 ? N - F - 8 : N - F - 3, yh = side == 0 ? N - F - 6 : N - F - 1;
            fp cx, cy, one_p;
            fp_one(&one_p);
            for (int c = 0; c < 2; c++) {
                uint64_t hi[6] = {0}, lo[6] = {0};
                fr_to_plain(hi, PI1(c == 0 ? xh : yh));
                fr_to_plain(lo, PI1((c == 0 ? xh : yh) - 1));
                fp h, l, v;
                fp_from_plain(&h, hi); fp_from_plain(&l, lo);
                fp_mul(&v, &h, &two224); fp_add(&v, &v, &l); fp_add(&v, &v, &one_p); /* (1 + hi*B + lo) mod p */
                if (c == 0) cx = v; else cy = v;
            }
            /* g1_from_coords (bls_utils.ak:32-49): compressed x with the parity flag of y, then decompress */
            uint8_t raw[48];
            fp_to_be48(raw, &cx);
            raw[0] |= fp_is_lex_larger(&cy) ? 0xa0 : 0x80;
            if (!g1_decompress(&acc_pt[side], raw)) REJECT(ORC_REJ_POINT);
        }
        g1j accl, accr, t;
        g1a accl_a, accr_a;
        g1j_from_affine(&t, &acc_pt[0]); g1j_mul_fr(&accl, &t, PI1(N - F - 5));   /* scaleG1(acc_left_unscaled, i_{N-F-5}) */
        g1j_from_affine(&t, &acc_pt[1]); g1j_mul_fr(&accr, &t, PI1(N - F));       /* scaleG1(acc_right_unscaled, i_{N-F}) */
        for (uint32_t k = 0; k < F; k++) g1_scale_add(&accr, &vk->rec_bases[k], PI1(N - F + 1 + k));  /* + acc_fixed */
        g1j_to_affine(&accl_a, &accl); g1j_to_affine(&accr_a, &accr);
        uint8_t msg[192], dig[32];
        g1_compress(msg, &el); g1_compress(msg + 48, &er); g1_compress(msg + 96, &accl_a); g1_compress(msg + 144, &accr_a);
        blake2b256(dig, msg, 192);
        fr ch;
        (void)fr_from_le32(&ch, dig);                                            /* from_int(le_int(digest) % field_prime) */
        g1j e1, e2;
        g1j_from_affine(&e1, &el); g1_scale_add(&e1, &accl_a, &ch); g1j_to_affine(&el, &e1);
        g1j_from_affine(&e2, &er); g1_scale_add(&e2, &accr_a, &ch); g1j_to_affine(&er, &e2);
#undef PI1
    }
    /* P7 pairing: accept <=> e(el, s_g2) == e(er, G2) (verification_h2.hbs:125-128) */
    status = pairing_check_eq(&el, &vk->s_g2, &er, &vk->g2gen) ? ORC_ACCEPT : ORC_REJ_PAIRING;

    if (trace) {
        put_fr(trace->theta, &theta); put_fr(trace->beta, &beta); put_fr(trace->gamma, &gamma); put_fr(trace->trash, &trash);
        put_fr(trace->y, &y); put_fr(trace->x, &x); put_fr(trace->x1, &x1); put_fr(trace->x2, &x2); put_fr(trace->x3, &x3); put_fr(trace->x4, &x4);
        fr t;
        rotate_omega(&t, &vk->omega, &vk->omega_inv, &x, -1); put_fr(trace->x_prev, &t);
        rotate_omega(&t, &vk->omega, &vk->omega_inv, &x, 1); put_fr(trace->x_next, &t);
        put_fr(trace->x_last, &x_last); put_fr(trace->xn, &xn);
        put_fr(trace->l_last, &l_last); put_fr(trace->l_0, &l_0); put_fr(trace->active_rows, &active_rows);
        put_fr(trace->h_eval, &h_eval); put_fr(trace->vanishing_s, &vanishing_s); put_fr(trace->f_eval, &f_eval); put_fr(trace->v, &v);
        put_g1(trace->vanishing_g, &vanishing_g); put_g1(trace->el, &el); put_g1(trace->er, &er);
        trace->n_expressions = (uint32_t)n_expr;
        for (int e = 0; e < n_expr && e < ORC_MAX_EXPR; e++) put_fr(trace->expressions[e], &expr[e]);
    }
done:
    if (trace) trace->status = status;
    if (set_pts) for (int s = 0; s < vk->n_sets; s++) { free(set_pts[s]); free(q_eval_sets[s]); }
    free(set_pts); free(q_eval_sets); free(set_sizes); free(q_evals);
    free(mem); free(pts); free(expr);
    return status == ORC_ACCEPT;
}

typedef struct { const orc_vk *vk; size_t lo, hi; const uint8_t *proofs; const uint64_t *off; const uint8_t *inst, *ci; uint8_t *accept; } job;
static void *batch_worker(void *arg) {
    job *j = (job *)arg;
    for (size_t i = j->lo; i < j->hi; i++)
        j->accept[i] = (uint8_t)orc_verify(j->vk, j->proofs + j->off[i], (size_t)(j->off[i + 1] - j->off[i]),
                                           j->inst + (size_t)32 * j->vk->n_pi * i, j->ci ? j->ci + 48 * i : 0, 0);
    return 0;
}
int orc_verify_batch(const orc_vk *vk, size_t n, const uint8_t *proofs, const uint64_t *proof_off,
                     const uint8_t *instances, const uint8_t *committed, uint8_t *accept, int threads) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    pthread_t *th = (pthread_t *)calloc(threads, sizeof(pthread_t));
    job *jobs = (job *)calloc(threads, sizeof(job));
    for (int t = 0; t < threads; t++) {
        job jb = {vk, n * t / threads, n * (t + 1) / threads, proofs, proof_off, instances, committed, accept};
        jobs[t] = jb;
        if (threads == 1) batch_worker(&jobs[t]);
        else pthread_create(&th[t], 0, batch_worker, &jobs[t]);
    }
    if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(th[t], 0);
    free(th); free(jobs);
    return 0;
}

/* ------------------------------------------------------------------ primitive entry points (golden tests) */
void orc_blake2b256(const uint8_t *in, size_t len, uint8_t out[32]) { blake2b256(out, in, len); }

long orc_transcript_script(const uint8_t *proof, size_t proof_len, const uint8_t *ops, size_t n_ops,
                           const uint8_t *args, uint8_t *out, size_t out_cap) {
    transcript tr;
    tr_init(&tr, proof, proof_len);
    size_t w = 0;
    for (size_t i = 0; i < n_ops; i++) {
        fr s; int canonical; const uint8_t *p;
        switch (ops[i]) {
        case 0: fr_from_le32(&s, args); args += 32; tr_common_scalar(&tr, &s); break;
        case 1: tr_common_point(&tr, args); args += 48; break;
        case 2: if (!tr_read_scalar(&tr, &s, &canonical)) return -1; if (w + 32 > out_cap) return -2; fr_to_le32(out + w, &s); w += 32; break;
        case 3: p = tr_read_point(&tr); if (!p) return -1; if (w + 48 > out_cap) return -2; memcpy(out + w, p, 48); w += 48; break;
        case 4: tr_squeeze(&tr, &s); if (w + 32 > out_cap) return -2; fr_to_le32(out + w, &s); w += 32; break;
        default: return -3;
        }
    }
    return (long)w;
}
int orc_fr_inv(const uint8_t a[32], uint8_t out[32]) {
    fr x;
    fr_from_le32(&x, a);
    if (!fr_inv(&x, &x)) return 0;
    fr_to_le32(out, &x);
    return 1;
}
void orc_rotate_omegas(const uint8_t omega[32], const uint8_t omega_inv[32], int from, int to, uint8_t *out) {
    fr w, wi, one;
    fr_from_le32(&w, omega); fr_from_le32(&wi, omega_inv); fr_one(&one);
    for (int r = from; r <= to; r++) { fr t; rotate_omega(&t, &w, &wi, &one, r); fr_to_le32(out + 32 * (r - from), &t); }
}
int orc_lagrange_basis(const uint8_t x[32], const uint8_t xn[32], const uint8_t w[32], const uint8_t *rotations,
                       size_t n, uint8_t *out) {
    fr fx, fxn, fw, *rots = (fr *)malloc(sizeof(fr) * (2 * n + 1));
    fr_from_le32(&fx, x); fr_from_le32(&fxn, xn); fr_from_le32(&fw, w);
    for (size_t i = 0; i < n; i++) fr_from_le32(&rots[i], rotations + 32 * i);
    int ok = lagrange_basis(rots + n, &fx, &fxn, &fw, rots, (int)n);
    if (ok) for (size_t i = 0; i < n; i++) fr_to_le32(out + 32 * i, &rots[n + i]);
    free(rots);
    return ok;
}
int orc_lagrange_evaluation(const uint8_t *points, const uint8_t *evals, size_t n, const uint8_t x[32], uint8_t out[32]) {
    fr xs[16], ys[16], fx, r;
    if (n > 16) return 0;
    for (size_t i = 0; i < n; i++) { fr_from_le32(&xs[i], points + 32 * i); fr_from_le32(&ys[i], evals + 32 * i); }
    fr_from_le32(&fx, x);
    if (!lagrange_evaluation(&r, xs, ys, (int)n, &fx)) return 0;
    fr_to_le32(out, &r);
    return 1;
}
int orc_multiopen_scalars(size_t n_sets, const uint32_t *set_sizes, const uint8_t *points, const uint32_t *n_comms,
                          const uint8_t *evals, const uint8_t x1[32], const uint8_t x2[32], const uint8_t x3[32],
                          const uint8_t x4[32], const uint8_t *q_evals, uint8_t *q_eval_sets, uint8_t f_eval[32],
                          uint8_t v[32]) {
    fr fx1, fx2, fx3, fx4, fe, fv;
    fr_from_le32(&fx1, x1); fr_from_le32(&fx2, x2); fr_from_le32(&fx3, x3); fr_from_le32(&fx4, x4);
    for (size_t s = 0; s < n_sets; s++) if (set_sizes[s] > MAX_SET_PTS) return 0;
    fr **pts = (fr **)calloc(n_sets, sizeof(fr *)), **qs = (fr **)calloc(n_sets, sizeof(fr *));
    int *sizes = (int *)calloc(n_sets, sizeof(int));
    fr *qe = (fr *)calloc(n_sets, sizeof(fr));
    for (size_t s = 0; s < n_sets; s++) {
        sizes[s] = (int)set_sizes[s];
        pts[s] = (fr *)calloc(sizes[s], sizeof(fr)); qs[s] = (fr *)calloc(sizes[s], sizeof(fr));
        for (int j = 0; j < sizes[s]; j++) { fr_from_le32(&pts[s][j], points); points += 32; }
        fr x1p; fr_one(&x1p);
        for (uint32_t c = 0; c < n_comms[s]; c++) {
            fr e[MAX_SET_PTS];
            const fr *evs[MAX_SET_PTS];
            for (int j = 0; j < sizes[s]; j++) { fr_from_le32(&e[j], evals); evals += 32; evs[j] = &e[j]; }
            q_eval_accumulate(qs[s], evs, sizes[s], &x1p);   /* the loop body orc_verify runs */
            fr_mul(&x1p, &x1p, &fx1);
        }
        fr_from_le32(&qe[s], q_evals + 32 * s);
    }
    int ok = multiopen_f_v((int)n_sets, sizes, pts, qs, &fx2, &fx3, &fx4, qe, &fe, &fv);
    if (ok) {
        for (size_t s = 0; s < n_sets; s++) for (int j = 0; j < sizes[s]; j++) { fr_to_le32(q_eval_sets, &qs[s][j]); q_eval_sets += 32; }
        fr_to_le32(f_eval, &fe); fr_to_le32(v, &fv);
    }
    for (size_t s = 0; s < n_sets; s++) { free(pts[s]); free(qs[s]); }
    free(pts); free(qs); free(sizes); free(qe);
    return ok;
}
/* Golden-vector entry point of the lookup block orc_verify runs (lookup_argument above).
 * exprs: n_in input expressions then n_tab table expressions, concatenated expression blobs;
 * scal: theta, beta, gamma, l_0, l_last, active_rows, product, product_next, permuted_input, permuted_input_inv,
 * permuted_table (11 x 32 bytes); out: the five identities (5 x 32 bytes). */
int orc_lookup_argument(const uint8_t *exprs, size_t exprs_len, uint32_t n_in, uint32_t n_tab, const uint8_t *advice, size_t n_adv,
                        const uint8_t *fixed, size_t n_fix, const uint8_t scal[11 * 32], uint8_t out[5 * 32]) {
    if (n_in + n_tab > 64) return 0;
    rd r = {exprs, exprs_len, 0, 0};
    epool pool = {0, 0, 0};
    int ids[64];
    for (uint32_t j = 0; j < n_in + n_tab; j++) ids[j] = parse_expr(&r, &pool, 0);
    if (r.err) { free(pool.n); return 0; }
    for (int i = 0; i < pool.cnt; i++) {
        if (pool.n[i].tag == EX_FIXED && (size_t)pool.n[i].idx >= n_fix) { free(pool.n); return 0; }
        if (pool.n[i].tag == EX_ADVICE && (size_t)pool.n[i].idx >= n_adv) { free(pool.n); return 0; }
    }
    fr *a = (fr *)calloc(n_adv + 1, sizeof(fr)), *f = (fr *)calloc(n_fix + 1, sizeof(fr)), v[11], o[5];
    for (size_t i = 0; i < n_adv; i++) fr_from_le32(&a[i], advice + 32 * i);
    for (size_t i = 0; i < n_fix; i++) fr_from_le32(&f[i], fixed + 32 * i);
    for (int i = 0; i < 11; i++) fr_from_le32(&v[i], scal + 32 * i);
    lookup_argument(o, pool.n, ids, n_in, ids + n_in, n_tab, a, f, &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6]);
    for (int i = 0; i < 5; i++) fr_to_le32(out + 32 * i, &o[i]);
    free(a); free(f); free(pool.n);
    return 1;
}
/* The commitment map and point sets orc_verify walks (build_sets), for the golden commitmentMap of ProofData.hs:184-197.
 * Per commitment, in order: ck, cidx, first-seen set index, position of that set after the cardinality sort, npts, then
 * npts x (rotation kind, rotation n, ek, eidx, esub).  Returns the number of int32 written, or -1 if cap is too small. */
long orc_vk_commitment_map(const orc_vk *vk, int32_t *out, size_t cap) {
    size_t w = 0;
    for (int c = 0; c < vk->n_comm; c++) {
        const commitment_data *cd = &vk->cd[c];
        if (w + 5 + 5 * (size_t)cd->npts > cap) return -1;
        int sorted_pos = -1;
        for (int s = 0; s < vk->n_sets; s++) if (vk->sort_order[s] == cd->set) sorted_pos = s;
        out[w++] = cd->ck; out[w++] = cd->cidx; out[w++] = cd->set; out[w++] = sorted_pos; out[w++] = cd->npts;
        for (int j = 0; j < cd->npts; j++) {
            out[w++] = cd->pts[j].kind; out[w++] = cd->pts[j].n; out[w++] = cd->ek[j]; out[w++] = cd->eidx[j]; out[w++] = cd->esub[j];
        }
    }
    return (long)w;
}
static int g1a_from_xy(g1a *p, const uint8_t xy[96]) {
    int z = 1;
    for (int i = 0; i < 96; i++) if (xy[i]) z = 0;
    if (z) { p->inf = 1; fp_zero(&p->x); fp_zero(&p->y); return 1; }
    p->inf = 0;
    return fp_from_be48(&p->x, xy) && fp_from_be48(&p->y, xy + 48);
}
int orc_g1_decompress(const uint8_t in[48], uint8_t out_xy[96]) {
    g1a p;
    if (!g1_decompress(&p, in)) return 0;
    put_g1(out_xy, &p);
    return 1;
}
void orc_g1_compress(const uint8_t xy[96], uint8_t out[48]) {
    g1a p;
    g1a_from_xy(&p, xy);
    g1_compress(out, &p);
}
int orc_g1_in_subgroup(const uint8_t xy[96], int naive) {
    g1a p;
    if (!g1a_from_xy(&p, xy) || !g1a_on_curve(&p)) return -1;
    return naive ? g1a_in_subgroup_naive(&p) : g1a_in_subgroup(&p);
}
void orc_g1_msm(size_t n, const uint8_t *scalars, const uint8_t *points_xy, uint8_t out_xy[96]) {
    g1j acc;
    g1j_set_inf(&acc);
    for (size_t i = 0; i < n; i++) {
        g1a p; fr k;
        g1a_from_xy(&p, points_xy + 96 * i);
        fr_from_le32(&k, scalars + 32 * i);
        g1_scale_add(&acc, &p, &k);
    }
    g1a r;
    g1j_to_affine(&r, &acc);
    put_g1(out_xy, &r);
}
int orc_pairing_check(const uint8_t p1[96], const uint8_t q1c[96], const uint8_t p2[96], const uint8_t q2c[96]) {
    g1a a, b; g2a q1, q2; g2prep pr1, pr2;
    if (!g1a_from_xy(&a, p1) || !g1a_from_xy(&b, p2)) return -1;
    if (!g2_decompress(&q1, q1c) || !g2_decompress(&q2, q2c)) return -1;
    if (!g2_prepare(&pr1, &q1) || !g2_prepare(&pr2, &q2)) return -1;
    return pairing_check_eq(&a, &pr1, &b, &pr2);
}
void orc_g2_generator_compressed(uint8_t out[96]) {
    fp x0, x1, y0, y1;
    fp_set(&x0, G2_GEN_X0); fp_set(&x1, G2_GEN_X1); fp_set(&y0, G2_GEN_Y0); fp_set(&y1, G2_GEN_Y1);
    fp_to_be48(out, &x1); fp_to_be48(out + 48, &x0);
    out[0] |= 0x80;
    fp2 y; y.c0 = y0; y.c1 = y1;
    if (fp2_is_lex_larger(&y)) out[0] |= 0x20;
}
int orc_eval_expr(const uint8_t *blob, size_t len, const uint8_t *advice, size_t n_adv, const uint8_t *fixed,
                  size_t n_fix, uint8_t out[32]) {
    rd r = {blob, len, 0, 0};
    epool pool = {0, 0, 0};
    int id = parse_expr(&r, &pool, 0);
    if (r.err) { free(pool.n); return 0; }
    for (int i = 0; i < pool.cnt; i++) {
        if (pool.n[i].tag == EX_FIXED && (size_t)pool.n[i].idx >= n_fix) { free(pool.n); return 0; }
        if (pool.n[i].tag == EX_ADVICE && (size_t)pool.n[i].idx >= n_adv) { free(pool.n); return 0; }
    }
    fr *a = (fr *)calloc(n_adv + 1, sizeof(fr)), *f = (fr *)calloc(n_fix + 1, sizeof(fr)), res;
    for (size_t i = 0; i < n_adv; i++) fr_from_le32(&a[i], advice + 32 * i);
    for (size_t i = 0; i < n_fix; i++) fr_from_le32(&f[i], fixed + 32 * i);
    eval_expr(pool.n, id, a, f, &res);
    fr_to_le32(out, &res);
    free(a); free(f); free(pool.n);
    return 1;
}
