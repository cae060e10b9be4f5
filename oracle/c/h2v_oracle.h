/* ORACLE - test infrastructure, not product code.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product (HIP) path never does.
 *
 * CPU restatement of the reference's Halo2/KZG verifier (the algorithm of
 * /root/reference/aiken-verifier/templates/verification_h2.hbs:21-129 with the slot contents defined by
 * /root/reference/src/plutus_gen/emitters/aiken.rs:94-646 and /root/reference/src/plutus_gen/extraction/).
 * Pinned against every in-tree golden vector of the path (tests/test_oracle_golden.py).
 */
#ifndef H2V_ORACLE_H
#define H2V_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_vk orc_vk;

#define ORC_MAX_EXPR 256
/* reject reasons (status) */
enum { ORC_ACCEPT = 0, ORC_REJ_PAIRING = 1, ORC_REJ_POINT = 2, ORC_REJ_SCALAR = 3, ORC_REJ_SHORT = 4, ORC_REJ_INVERSE = 5, ORC_REJ_RECURSION = 6 };

/* The reference's own intermediate-value surface (plutus_debug TRACES, src/plutus_gen/emitters/plinth.rs:792-831)
 * plus the multi-open scalars.  Scalars 32 B little-endian canonical; points affine x||y big-endian, all-zero
 * = point at infinity. */
typedef struct {
    int32_t status;
    uint32_t n_expressions;
    uint8_t theta[32], beta[32], gamma[32], trash[32], y[32], x[32], x1[32], x2[32], x3[32], x4[32];
    uint8_t x_prev[32], x_next[32], x_last[32], xn[32];
    uint8_t l_last[32], l_0[32], active_rows[32];
    uint8_t h_eval[32], vanishing_s[32], f_eval[32], v[32];
    uint8_t vanishing_g[96], el[96], er[96];
    uint8_t expressions[ORC_MAX_EXPR][32];
} orc_trace;

/* vk description blob: see oracle/vkdesc.py */
orc_vk *orc_vk_parse(const uint8_t *desc, size_t len);
void orc_vk_free(orc_vk *vk);
int orc_vk_num_point_sets(const orc_vk *vk);
int orc_vk_num_msm_terms(const orc_vk *vk);
size_t orc_vk_proof_len(const orc_vk *vk);

/* returns 1 accept / 0 reject; instances = n_pi * 32 B LE; committed = 48 B compressed or NULL; trace optional */
int orc_verify(const orc_vk *vk, const uint8_t *proof, size_t proof_len, const uint8_t *instances,
               const uint8_t *committed, orc_trace *trace);
/* batch over independent proofs, `threads` pthreads (contiguous ranges); accept[i] in {0,1} */
int orc_verify_batch(const orc_vk *vk, size_t n, const uint8_t *proofs, const uint64_t *proof_off,
                     const uint8_t *instances, const uint8_t *committed, uint8_t *accept, int threads);

/* ---- primitive entry points for the golden-vector tests (all scalars 32 B LE, reduced mod r on input) */
void orc_blake2b256(const uint8_t *in, size_t len, uint8_t out[32]);
/* generic transcript script: ops[i] in {0: common_scalar(arg 32B), 1: common_point(arg 48B), 2: read_scalar,
 * 3: read_point, 4: squeeze}; args are consumed from `args` in order; every read/squeeze result is appended
 * to out (scalars 32 B LE reduced; points 48 raw bytes).  returns bytes written, or -1 on a short proof. */
long orc_transcript_script(const uint8_t *proof, size_t proof_len, const uint8_t *ops, size_t n_ops,
                           const uint8_t *args, uint8_t *out, size_t out_cap);
int orc_fr_inv(const uint8_t a[32], uint8_t out[32]);
void orc_rotate_omegas(const uint8_t omega[32], const uint8_t omega_inv[32], int from, int to, uint8_t *out);
int orc_lagrange_basis(const uint8_t x[32], const uint8_t xn[32], const uint8_t w[32], const uint8_t *rotations,
                       size_t n, uint8_t *out);
int orc_lagrange_evaluation(const uint8_t *points, const uint8_t *evals, size_t n, const uint8_t x[32],
                            uint8_t out[32]);
/* multi-open scalars: sets given explicitly.  set_sizes[s] = #points, points/evals laid out per set;
 * commitments: per set n_comms[s] entries, each with set_sizes[s] evals.  q_evals: S scalars.
 * outputs: q_eval_sets (sum set_sizes scalars), f_eval, v */
int orc_multiopen_scalars(size_t n_sets, const uint32_t *set_sizes, const uint8_t *points, const uint32_t *n_comms,
                          const uint8_t *evals, const uint8_t x1[32], const uint8_t x2[32], const uint8_t x3[32],
                          const uint8_t x4[32], const uint8_t *q_evals, uint8_t *q_eval_sets, uint8_t f_eval[32],
                          uint8_t v[32]);
int orc_g1_decompress(const uint8_t in[48], uint8_t out_xy[96]);
void orc_g1_compress(const uint8_t xy[96], uint8_t out[48]);
int orc_g1_in_subgroup(const uint8_t xy[96], int naive);
/* sum_i s_i * P_i with affine inputs (x||y BE, zero = infinity); naive fold (bls_utils.ak:77-86) */
void orc_g1_msm(size_t n, const uint8_t *scalars, const uint8_t *points_xy, uint8_t out_xy[96]);
/* e(p1, q1) == e(p2, q2); G1 affine 96 B, G2 compressed 96 B */
int orc_pairing_check(const uint8_t p1[96], const uint8_t q1c[96], const uint8_t p2[96], const uint8_t q2c[96]);
void orc_g2_generator_compressed(uint8_t out[96]);
/* evaluate one expression blob (vkdesc encoding) on given advice/fixed evals */
int orc_eval_expr(const uint8_t *blob, size_t len, const uint8_t *advice, size_t n_adv, const uint8_t *fixed,
                  size_t n_fix, uint8_t out[32]);

/* the lookup block of orc_verify (theta-compression + five identities) on explicit inputs: gates_test.hbs vectors */
int orc_lookup_argument(const uint8_t *exprs, size_t exprs_len, uint32_t n_in, uint32_t n_tab, const uint8_t *advice, size_t n_adv,
                        const uint8_t *fixed, size_t n_fix, const uint8_t scal[11 * 32], uint8_t out[5 * 32]);
/* commitment map + point sets as orc_verify walks them (ProofData.hs:184-197 commitmentMap) */
long orc_vk_commitment_map(const orc_vk *vk, int32_t *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
