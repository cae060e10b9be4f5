/* h2v - MI355X (gfx950) batch verifier backend for Halo2/KZG proofs over BLS12-381: the C-ABI boundary.
 *
 * This library is a drop-in for ONE hot path of input-output-hk/plutus-halo2-verifier-gen: the verification the
 * reference runs on the CPU as the two-call sequence re-exported at /root/reference/src/lib.rs:9-15
 *
 *     let guard = prepare(&vk, committed_instances, instances, &mut transcript)?;     // examples/simple_mul.rs:98
 *     guard.verify(&params.verifier_params())?;                                        // examples/simple_mul.rs:101
 *
 * (in the IVC example the guard is a DualMSM with `.check(&params) -> bool`, examples/ivc.rs:85-96,195-198), and
 * that it re-states as generated Plinth/Aiken code (aiken-verifier/templates/verification_h2.hbs:21-129).
 * The reference has no FFI for this path (Rust generics over midnight-proofs); these entry points are what a
 * `extern "C"` binding of that call pair needs - see INTEGRATION.md for the Rust-side stub.
 *
 * Conventions
 *   - plain pointers and sizes only; the library never retains caller pointers past return;
 *   - scalars: 32 bytes little-endian (the transcript wire format, transcript.ak:29-45);
 *     G1: 48 bytes zcash-compressed (bls_utils.ak:17-28);
 *   - return codes: 0 = ok, negative = API misuse / device error (H2V_E_*).  A malformed or invalid PROOF is
 *     never an API error: it is accept[i] = 0, mirroring `Err(_)` => reject on the Rust side;
 *   - no CPU fallback: every verify call runs on the GPU or fails with H2V_E_DEVICE.
 */
#ifndef H2V_H
#define H2V_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define H2V_OK 0
#define H2V_E_ARG (-1)     /* null / inconsistent argument */
#define H2V_E_PLAN (-2)    /* malformed plan blob */
#define H2V_E_DEVICE (-3)  /* HIP error (no device, out of memory, launch failure) */
#define H2V_E_LIMIT (-4)   /* plan exceeds a backend limit (e.g. more than 64 MSM terms) */

/* per-proof status bits (h2v_verify_batch_ex / trace): 0 = accepted */
#define H2V_ST_BAD_SCALAR 1u       /* non-canonical scalar encoding in the proof */
#define H2V_ST_INVERSE_OF_ZERO 2u  /* an inversion the verifier needs hit zero (recip_eea panics, bls_utils.ak:151-154) */
#define H2V_ST_SHORT_PROOF 4u      /* fewer bytes than the plan's proof layout */
#define H2V_ST_BAD_POINT 8u        /* a G1 encoding is malformed / off-curve / not in the subgroup */
#define H2V_ST_PAIRING 16u         /* e(pi, s_g2) != e(er, G2) */
#define H2V_ST_RECURSION 32u       /* IVC: the verifying-key hash in the public inputs is not this key's (aiken.rs:696) */

typedef struct h2v_plan h2v_plan;           /* replaces VerifyingKey<F,CS> + ParamsVerifierKZG for this path */
typedef struct h2v_workspace h2v_workspace; /* reusable device scratch for batches of up to max_batch proofs */

/* One batch of independent proofs for the same plan.  Replaces the arguments of `prepare`
 * (src/lib.rs:9-15; call sites examples/simple_mul.rs:98, examples/sha256.rs:131-137):
 *   proofs      concatenated proof bytes, proof i = proofs[proof_off[i] .. proof_off[i+1])   (transcript bytes)
 *   instances   n * n_public_inputs * 32 bytes  (instances: &[&[&[F]]], one public column)
 *   committed   n * 48 bytes or NULL            (committed_instances: &[&[CS::Commitment]], 0 or 1 per proof) */
typedef struct {
    uint64_t n;
    const uint8_t *proofs;
    const uint64_t *proof_off; /* n + 1 entries */
    const uint8_t *instances;
    const uint8_t *committed;
} h2v_batch;

/* optional per-stage device timings (milliseconds, HIP events on the streams the kernels run on).  A large batch is
 * cut into `launches` chunks that run as overlapping pipelines when H2V_PIPES > 1 (default 1): each *_ms is
 * the SUM of that kernel's launch durations over the chunks, total_ms the span from the first launch to the last end.
 * g1_decompress_ms is the longer of the decompression kernel's two concurrent launches (square roots + window tables
 * on one stream, subgroup tests on another).  msm_lanes_per_term is the shape the MSM launcher chose for the call:
 * 2 = one lane per GLV half, 1 = both halves on one lane (shared doublings), 3 = ladders for the per-proof terms beside a
 * fixed-base launch for the VK bases, 8 = a quad of lanes per GLV half (small launches of few terms: the four lanes share
 * the multiplications of every doubling and addition), 18 / 20 = two / four terms per lane (H2V_MSM_TPL). */
typedef struct {
    float transcript_combiner_ms, g1_decompress_ms, g1_msm_ms, pairing_ms, total_ms;
    uint32_t launches;
    uint32_t msm_lanes_per_term;
    uint32_t pairing_lanes_per_proof;   /* 32: two proofs per wave; 64: the wide engine (small launches); 16: the narrow one (four
                                         * proofs per wave); 12: the narrow engine packed five proofs to a wave; 6: ten proofs per wave, a whole Fp2 coefficient
                                         * per lane (Karatsuba terms: fewest instructions, for full chips); 1: the one-lane cross-check kernel */
    /* (round 3) msm_lanes_per_term == 3: the MSM ran as TWO kernels side by side - g1_msm_ms is the ladder launch over the
     * per-proof terms (its shape: msm_var_lanes_per_term, coded like msm_lanes_per_term), g1_msm_fixed_ms the fixed-base
     * launch over the VK-base terms; 0 otherwise */
    float g1_msm_fixed_ms;
    uint32_t msm_var_lanes_per_term;
} h2v_timings;

/* ---- plan (VerifyingKey) lifecycle -------------------------------------------------------------------------
 * h2v_plan_load: parse + validate the plan blob (plutus_halo2_verifier_gen_amd.plan.compile_plan(vk).to_bytes())
 * and upload it to `device`.  Immutable after load; may be shared by threads (each with its own workspace). */
int h2v_plan_load(const uint8_t *plan, size_t len, int device, h2v_plan **out);
/* (round 4) the same with load-time options (what the environment variable H2V_FIX_C used to select): opts = NULL or all-zero
 * fields = h2v_plan_load.  fixed_base_window_bits: window width of the all-window tables of the VK bases that the split MSM's
 * fixed-base launch reads - 0: the library's choice (12 bits: 22 additions per base and proof, 5 MB of tables per base; 8
 * when 12 would take more than 2 GB), or 4 / 8 / 12.  Results never depend on it. */
typedef struct {
    uint32_t fixed_base_window_bits;
    uint32_t reserved[7];   /* zero */
} h2v_plan_opts;
int h2v_plan_load_ex(const uint8_t *plan, size_t len, int device, const h2v_plan_opts *opts, h2v_plan **out);
/* (round 3) The plan compiler behind the boundary: a verifying-key description - the JSON form frozen as docs/vk_schema.json
 * ("h2v-vk/1": what `extract_circuit` reads out of a midnight_proofs VerifyingKey + ParamsVerifierKZG,
 * /root/reference/src/plutus_gen/extraction/mod.rs:31-232, instantiation_data.rs:26-41; the serde exporter is in
 * INTEGRATION.md) - to the blob h2v_plan_load takes.  Host-only (no GPU needed), byte-identical with
 * plutus_halo2_verifier_gen_amd.plan.compile_plan(vk).to_bytes().  *blob_out is malloc'ed: release it with h2v_blob_free.
 * A description the reference itself could not emit a verifier for (Selector / Instance / Challenge nodes, a permutation
 * column without a query at the current rotation, commitments that are not G1 points ...) is H2V_E_ARG with the reason in
 * h2v_last_error(). */
int h2v_plan_compile(const char *vk_json, size_t len, uint8_t **blob_out, size_t *blob_len);
void h2v_blob_free(uint8_t *blob);
void h2v_plan_free(h2v_plan *plan);
/* plan facts: proof length in bytes, public inputs per proof, committed instances per proof (0/1), MSM terms T */
int h2v_plan_info(const h2v_plan *plan, uint32_t *proof_len, uint32_t *n_public_inputs, uint32_t *n_committed,
                  uint32_t *n_msm_terms);

/* A workspace is sized from `plan` (MSM terms, point slots, register file, recursion / fixed-base buffers).  It may be
 * reused with ANOTHER plan only if that plan needs no more of any of these; otherwise the verify calls return H2V_E_ARG
 * (they never write past a buffer).  One workspace serves one call at a time. */
int h2v_workspace_create(const h2v_plan *plan, uint64_t max_batch, h2v_workspace **out);
void h2v_workspace_free(h2v_workspace *ws);
/* (round 3) LANES: the pipelining that callers used to build from several workspaces and streams, inside the library.
 * A laned workspace owns n_lanes sub-workspaces of `chunk` proofs each, every one with its own library-owned stream.  A verify
 * call on it (device-resident, host-buffer or RLC form, any n <= max_batch) is cut into chunks of at most `chunk` proofs
 * that go round robin through the lanes, so that the decompression / combiner kernels of one chunk run beside the MSM and
 * pairing kernels of others; accept[] / status[] are written in place, chunk by chunk, and the caller's stream waits for
 * the last one (unless joins are deferred, below).  Device memory is n_lanes x chunk proofs' worth, whatever max_batch is.
 * Verdicts never depend on n_lanes or chunk (tests/test_gpu_parity.py::test_verdicts_do_not_depend_on_the_chunking); in
 * RLC mode every chunk is its own batch check.
 *   n_lanes = 0 / chunk = 0: the library's choice (chunk = 4096 proofs whatever the plan and whatever max_batch - the largest
 *   single call - is: small calls are gathered into the lanes, COALESCING below; 16 lanes, of which the per-proof mode cycles
 *   through 8 - all 16 for launches too small to fill the chip: h2v_workspace_depth).  An explicit chunk is cut to max_batch.
 *   h2v_workspace_create(plan, max_batch) itself returns a laned workspace when max_batch >= 4 x the batch that gives every kernel
 *   of the plan one wave per SIMD (4096 proofs for 16 MSM terms, 1024 for 60; an ordinary workspace below that).  n_lanes <= 16: that is also the
 *   most host batches h2v_verify_batch_submit keeps in flight on one workspace (h2v::BatchStream / backend.BatchStream: depth
 *   <= 16).  A laned workspace has no trace buffer (h2v_trace creates its own). */
int h2v_workspace_create_lanes(const h2v_plan *plan, uint64_t max_batch, uint32_t n_lanes, uint32_t chunk, h2v_workspace **out);
/* One laned workspace for SEVERAL plans of one device - a node that verifies proofs of several circuits (BASELINE configs[2]: a
 * mixed batch of two circuits).  The lanes are sized for the largest of every dimension of the listed plans (a workspace
 * accepts any plan whose dimensions fit its buffers), so the calls of all of them - each names its plan as usual - go round
 * robin through ONE set of lanes and streams; with one laned workspace per plan the lanes of the two share the pool's sixteen
 * streams pairwise.  chunk = 0: the smallest of the plans' own chunks.  Everything else as h2v_workspace_create_lanes; options
 * (h2v_workspace_set_option, h2v_workspace_tune) hold for the workspace, i.e. for every plan on it.  The RLC mode keeps one set
 * of its buffers per plan shape on such a workspace (up to eight).  The reference has no counterpart: its verifier objects
 * are per proof (src/lib.rs:9-15) and a mixed batch is a Vec of (vk, proof) pairs. */
int h2v_workspace_create_multi(const h2v_plan *const *plans, uint32_t n_plans, uint64_t max_batch, uint32_t n_lanes, uint32_t chunk,
                               h2v_workspace **out);
/* Deferred joins (laned workspaces): with defer = 1 a device-resident verify call returns without making the caller's stream
 * wait for its chunks, so the chunks of CONSECUTIVE calls overlap in the lanes - one workspace then does what five
 * workspaces on five streams did (a stream of batches: DESIGN.md section 6).  The inputs of a call must stay untouched, and
 * its accept[] / status[] unread, until h2v_workspace_join(ws, stream) has been called and `stream` has reached that point
 * (stream = NULL: the HOST blocks until every chunk submitted so far is done; nothing is enqueued on the NULL stream).  Calls
 * still start in submission order, each after whatever was enqueued on its `stream` argument before it.
 * THE NULL STREAM (round 4).  The lanes run on library-owned streams with hardware queues of their own
 * (hipExtStreamCreateWithCUMask), and those have the default flags: they are BLOCKING streams, ordered with the legacy NULL
 * stream.  Any work a host puts on the NULL stream - PyTorch's default stream is that stream - waits for every lane and holds
 * every lane back behind it, so the overlap of consecutive calls would be lost without an error.  Therefore, on a workspace
 * with deferred joins, h2v_verify_batch_device / h2v_verify_batch_rlc_device return H2V_E_ARG for stream = NULL: give the
 * calls a stream of their own (hipStreamCreateWithFlags(.., hipStreamNonBlocking); torch.cuda.Stream()) and keep other NULL-
 * stream work (synchronous hipMemcpy included) out of the submission loop.  Without deferred joins NULL is accepted: every
 * call then waits for its own chunks anyway.  tests/test_gpu_parity.py::test_null_stream_contract_of_deferred_joins. */
/* COALESCING (round 4).  With deferred joins a device-resident call (per proof or RLC) of at most HALF the workspace's chunk is not
 * launched by itself: its proofs are gathered - at once, behind whatever its `stream` held at the time of the call - behind
 * those of the small calls before it, and the kernels run ONCE over the group: when the next call would not fit the chunk, when
 * a call of another plan or another kind (large, RLC, host-buffer, tune) arrives, or at h2v_workspace_join.  Nothing changes for
 * the caller beyond what deferred joins already say - inputs untouched and accept[] / status[] unread until the join - but a
 * stream of 64-proof calls costs a 1024-proof launch per sixteen of them instead of sixteen chains of lone waves (the per-GPU
 * shares of a batch cut over eight GPUs: DESIGN.md section 6.1).  Verdicts do not depend on it.  h2v_workspace_timings of such
 * a call reports its share of the group's launch.  RLC calls are gathered among themselves: ONE batch check over the group, with
 * the coefficients of its first call's seed; accept[] stays per proof and exact, h2v_workspace_rlc_result of each call reports the
 * GROUP's batch verdict (a rejecting proof of a neighbouring call fails the check for all of them) and the call's share of the
 * times.  An open group does not start by itself: a caller that stops submitting for a while and wants the GPU to get on with
 * what it has calls h2v_workspace_join(ws, stream) with a stream of its own - that runs the open groups and does not block the
 * host.  A workspace keeps one group open per plan and mode (at most four, and fewer than it has lanes).  Host-buffer calls
 * (h2v_verify_batch, _submit) are not gathered: a host that collects small batches concatenates them itself.
 * H2V_OPT_COALESCE = -1 switches it off for a workspace. */
int h2v_workspace_defer_joins(h2v_workspace *ws, int defer);
int h2v_workspace_join(h2v_workspace *ws, void *stream);
/* Launch-shape options of a workspace (round 3: what used to be reachable through environment variables only; results never
 * depend on them - tests/test_gpu_parity.py::test_workspace_options_change_the_shape_not_the_verdicts).  value 0 = the
 * launcher's own choice (DESIGN.md sections 4.1, 4.2, 6). */
#define H2V_OPT_MSM_TERMS_PER_LANE 1u /* per-proof MSM: 1 .. 4 terms per lane on one accumulator (shared doublings) */
#define H2V_OPT_PAIRING_ENGINE 2u     /* lanes per proof of the pairing kernel: 6 (ten proofs per wave), 12 (five), 16 (narrow: four), 32, 64 (wide), 1 (the one-lane cross-check kernel) */
#define H2V_OPT_STREAMS 3u            /* -1 auto, 0: three library streams per call, 1: everything on the caller's stream, 2: + one side stream */
/* (round 4) every remaining shape dimension, formerly H2V_* environment variables read once per process: */
#define H2V_OPT_MSM_LANES_PER_TERM 4u /* ladder launches: 1 (both GLV halves on one lane), 2 (one lane per half), 8 (a quad of lanes per half) */
#define H2V_OPT_MSM_BLOCK_SIZE 5u     /* threads per block of the MSM launches: 64 .. 512 in steps of 64 */
#define H2V_OPT_MSM_FIXED_SPLIT 6u    /* VK-base terms through the all-window tables beside the ladders: 1 .. 4 bases per lane; -1: never split */
#define H2V_OPT_COMBINER_SCHEDULE 7u  /* transcript + combiner kernel: 1 the plan's narrow bundle schedule, 2 the wide one */
#define H2V_OPT_COMBINER_PROOFS_PER_BLOCK 8u /* proofs per one-wave block of that kernel (a power of two <= what fits LDS) */
#define H2V_OPT_DECOMPRESS_FORM 9u    /* 0: one launch that takes 64-point units from a queue; 1: square roots and subgroup tests as two launches; 2: one launch of paired 128-thread blocks */
#define H2V_OPT_PIPES 10u             /* an ordinary workspace cuts a call into 2 .. 4 sub-pipelines on streams of their own (0 / 1: one) */
#define H2V_OPT_RLC_GROUP_STAGE 11u   /* RLC fall-back: -1 skips the group checks (straight to the per-proof kernels) */
#define H2V_OPT_RLC_WINDOW_BITS 12u   /* bucket MSM: window width */
#define H2V_OPT_RLC_CHAIN 13u         /* bucket MSM: most entries one lane sums */
#define H2V_OPT_RLC_ROUTE 14u         /* RLC calls are routed by the observed rate of failing groups (h2v_verify_batch_rlc below); -1: always the batch check first */
#define H2V_OPT_COALESCE 15u          /* small device-resident calls on a deferring laned workspace are gathered into one launch per kernel (h2v_workspace_defer_joins below); -1: never */
#define H2V_OPT_COUNT 16u
int h2v_workspace_set_option(h2v_workspace *ws, uint32_t option, int32_t value);
int h2v_workspace_get_option(const h2v_workspace *ws, uint32_t option, int32_t *value);
/* (round 4) MEASURED launch shapes.  The launcher's own rules are thresholds calibrated on five circuit shapes; a circuit or a
 * batch size in between gets the nearest rule, unmeasured.  h2v_workspace_tune runs `batch` (device-resident, as for
 * h2v_verify_batch_device; its verdicts are discarded) repeatedly in the way the workspace is used - a laned workspace with as
 * many calls in flight as it has lanes for that size, an ordinary one call by call -, times the launcher's choice and its
 * neighbours (pairing engine, then MSM terms per lane: <= 8 measurements of three rounds of the lanes or ~120 ms each, whichever
 * is longer - about a second in all), and leaves the fastest on
 * the workspace as H2V_OPT_PAIRING_ENGINE / H2V_OPT_MSM_TERMS_PER_LANE (0 = the launcher's rule stays: a candidate must be
 * 3 % faster to replace it).  Call it once per (plan, batch size, workspace) at start-up, on `stream` (not NULL for a laned
 * workspace); it returns when the measurements are done.  flags: 0.  Results of later calls do not depend on it. */
typedef struct {
    uint32_t n_measured;          /* configurations timed */
    uint32_t calls_in_flight;     /* how the batch was run: calls kept in flight */
    float default_ms, best_ms;    /* ms per call: the launcher's own choice / the configuration now set */
    int32_t pairing_engine;       /* H2V_OPT_PAIRING_ENGINE now set (0: the launcher's rule) */
    int32_t msm_terms_per_lane;   /* H2V_OPT_MSM_TERMS_PER_LANE now set (0: the launcher's rule) */
    uint32_t reserved[4];
} h2v_tune_report;
int h2v_workspace_tune(const h2v_plan *plan, const h2v_batch *batch, h2v_workspace *ws, void *stream, uint32_t flags,
                       h2v_tune_report *report /* or NULL */);
/* the same options for the h2v_probe_* calls of the calling thread (they have no workspace) */
int h2v_probe_set_option(uint32_t option, int32_t value);
/* lanes and chunk size of a workspace (1 lane = not laned) */
int h2v_workspace_lanes(const h2v_workspace *ws, uint32_t *n_lanes, uint32_t *chunk);
/* how many calls of n proofs each (rlc != 0: RLC mode) the workspace keeps in flight before a call has to wait for a lane:
 * the lanes that mode cycles through for chunks of that size (1 for a workspace that is not laned).  Per-proof mode: 8 lanes
 * with the decompression on a side stream, or - chunks that give the MSM at most half a wave per SIMD - all 16, each
 * chunk on its lane's stream alone; RLC mode: all of them. */
int h2v_workspace_depth(const h2v_workspace *ws, uint64_t n, int rlc, uint32_t *batches_in_flight);
/* (new) Tuning hint: the caller keeps n_in_flight batches in flight on this device (each on its own workspace).  From 4 up
 * the launcher prefers shapes that issue fewer instructions over shapes with shorter dependent chains (per-proof MSM: two
 * terms per lane on one accumulator; the narrow pairing engine from 2 x #SIMDs proofs; from 6 the whole per-proof pipeline on
 * the caller's stream; RLC mode: from 3 the one-stream form, as with H2V_RLC_ONE_STREAM; from 8 the thresholds of a chip that
 * is full whatever one launch brings: DESIGN.md section 6.1).  Results do not depend on it.  Default 1.  A LANED workspace
 * estimates it per call by itself - all its lanes when joins are deferred or host batches are streamed, otherwise the number
 * of chunks of the call - unless this function was called: then the given value holds. */
int h2v_workspace_hint_in_flight(h2v_workspace *ws, uint32_t n_in_flight);
/* Per-kernel device times of a past call that used `ws` (calls_back = 0: the most recent; up to 63 back), from HIP
 * events recorded on the streams the kernels ran on.  Synchronise the launch stream before asking. */
int h2v_workspace_timings(h2v_workspace *ws, uint32_t calls_back, h2v_timings *out);

/* ---- verification (replaces prepare + Guard::verify / DualMSM::check) --------------------------------------
 * Host-buffer form: copies the batch to the device, verifies, copies accept[] (n bytes, 1 = accept) back.
 * ws may be NULL (a temporary workspace is created for the call). */
int h2v_verify_batch(const h2v_plan *plan, const h2v_batch *batch, uint8_t *accept, h2v_workspace *ws);

/* Asynchronous host-buffer form, for callers that stream batches: _submit copies the caller's buffers into pinned staging
 * memory of `ws` (they may be reused at once), enqueues ONE upload, the verification (per proof, or the RLC batch mode with
 * H2V_SUBMIT_RLC - see below) and the download of accept[] on the workspace's own stream, and returns; _wait blocks until
 * that batch is done and copies its accept bytes out.  An ordinary workspace holds one batch at a time (alternate two and
 * the upload of batch k+1 runs beside the kernels of batch k); a LANED workspace (h2v_workspace_create_lanes) holds as many as
 * the mode has lanes - every batch gets a staging slot of its own - and _wait collects the OLDEST: one workspace is a
 * whole stream of host batches.  h2v_verify_batch is _submit + _wait. */
#define H2V_SUBMIT_RLC 1u
struct h2v_rlc_opts_s;
int h2v_verify_batch_submit(const h2v_plan *plan, const h2v_batch *batch, h2v_workspace *ws, uint32_t flags,
                            const struct h2v_rlc_opts_s *rlc_opts /* or NULL */);
int h2v_verify_batch_wait(h2v_workspace *ws, uint8_t *accept, int *fell_back /* or NULL */);

/* Device-resident form: every pointer in `batch` and `accept`/`status` are DEVICE pointers on the plan's device
 * (e.g. torch tensors' data_ptr()); work is enqueued on `stream` (a hipStream_t, NULL = default stream) and the
 * call returns after enqueueing unless `timings` is given (then it synchronises to read the events).
 * status (n x uint32, may be NULL) receives the H2V_ST_* bits. */
int h2v_verify_batch_device(const h2v_plan *plan, const h2v_batch *batch, uint8_t *accept, uint32_t *status,
                            h2v_workspace *ws, void *stream, h2v_timings *timings);

/* ---- batch-accept fast path: random linear combination (RLC) of the batch's pairing equations -------------------
 * Same inputs and the same accept[] / status[] outputs as h2v_verify_batch(_device).  Instead of one MSM and one pairing
 * per proof, the batch is folded with 128-bit coefficients r_i (blake2b-256(seed || i), seed from the OS unless given)
 * into ONE check  e(sum_i r_i pi_i, s_g2) == e(sum_i r_i er_i, G2)  - the dual-MSM form of
 * aiken-verifier/aiken_halo2/lib/halo2_kzg.ak:37-43 and the pairing equation of verification_h2.hbs:125-128 summed over
 * the batch - computed by a bucketed (Pippenger) G1 MSM over every per-proof point of the batch.  Proofs that are rejected
 * before the pairing (malformed encodings, points off the curve or outside G1, inverse of zero ...) are rejected
 * individually and take no part in the combination.  If the batch check fails, the per-proof MSM and pairing kernels run
 * for the whole batch, so accept[] is always what the per-proof mode returns, except with probability <= 2^-128 over the
 * seed (a rejecting proof hidden by the combination).  Recursive (IVC) plans have no batch form and run per proof.
 * ws must not be NULL for the device form.
 * ROUTING (round 4): a failed batch check costs more than the per-proof mode it falls back to, so a workspace keeps a running
 * estimate of the rate of FAILING GROUPS (64 proofs each) among the groups its RLC calls have met, and while that rate is
 * above 0.10 it sends RLC calls straight to the per-proof kernels (back to the batch check below 0.05; the estimate moves a
 * quarter of the way per call: one rejecting proof per batch - one group in 64 - never switches, a batch in which half the
 * groups fail switches the next call, and some ten clean calls switch back).  accept[] is the same either way;
 * fell_back = 1 / batch_accepted = 0 say that the per-proof kernels produced it.  H2V_OPT_RLC_ROUTE = -1 switches routing off. */
#define H2V_RLC_SEED_GIVEN 1u /* TEST ONLY: soundness rests on the seed being unpredictable to the provers; without this flag
                               * the library draws 32 bytes from the OS per call (getrandom) and fails closed.  With it, a
                               * process-wide call counter is still mixed into the seed, so repeated calls differ. */
#define H2V_RLC_ONE_STREAM 2u /* everything on the caller's stream (decompression before the combiner instead of beside it):
                               * 0.3 ms more per batch alone, but one stream per batch for callers that keep many batches in
                               * flight - more of them fit the hardware queues (measured: 2.47 M proofs/s with 7 in flight
                               * against 2.24 M with 5 on two streams each) */
typedef struct h2v_rlc_opts_s {
    uint8_t seed[32];   /* used when flags & H2V_RLC_SEED_GIVEN (tests, reproducible measurements - never a service) */
    uint32_t flags;
} h2v_rlc_opts;
typedef struct {
    float transcript_combiner_ms, g1_decompress_ms, prepare_ms, bucket_sort_ms, bucket_accumulate_ms, bucket_reduce_ms,
          pairing_ms, total_ms;   /* total: first phase-1 launch .. the batch verdict (a fall-back run is not included) */
    uint32_t msm_terms, window_bits, windows, max_chain;   /* shape of the right-hand bucket MSM; max_chain: entries per lane */
} h2v_rlc_timings;
int h2v_verify_batch_rlc(const h2v_plan *plan, const h2v_batch *batch, uint8_t *accept, h2v_workspace *ws,
                         const h2v_rlc_opts *opts /* or NULL */, int *fell_back /* or NULL: 1 = the per-proof kernels ran */);
int h2v_verify_batch_rlc_device(const h2v_plan *plan, const h2v_batch *batch, uint8_t *accept, uint32_t *status,
                                h2v_workspace *ws, void *stream, const h2v_rlc_opts *opts /* or NULL */);
/* after synchronising the stream of an RLC call: the batch verdict (1 = passed) and the kernel times of that call */
int h2v_workspace_rlc_result(h2v_workspace *ws, uint32_t calls_back, uint32_t *batch_accepted, h2v_rlc_timings *timings);

/* ---- parity / debugging surface ----------------------------------------------------------------------------
 * The reference's own intermediate-value trace (cargo feature plutus_debug, src/plutus_gen/emitters/plinth.rs:792-831):
 * theta, beta, gamma, x, y, hEval, vanishing_s, ..., every expression_i, plus el / er.
 * Host buffers.  scalars_out: n_trace * 32 bytes LE in the order of h2v_plan_trace_slots(); er_out / el_out:
 * 96 bytes affine x||y big-endian (all-zero = infinity). */
int h2v_plan_trace_slots(const h2v_plan *plan, uint32_t *slot_ids, uint32_t cap, uint32_t *n_out);
int h2v_trace(const h2v_plan *plan, const uint8_t *proof, size_t proof_len, const uint8_t *instances,
              const uint8_t *committed, uint8_t *scalars_out, uint8_t *msm_scalars_out /* T*32 or NULL */,
              uint8_t el_out[96], uint8_t er_out[96], uint32_t *status_out, uint8_t *accept_out);

/* primitive probes for the GPU parity tests (host buffers; canonical little-endian limbs) */
int h2v_probe_field(int device, int op, uint32_t n, const uint32_t *a, const uint32_t *b, uint32_t *out);
int h2v_probe_blake2b(int device, uint32_t n, uint32_t len, const uint8_t *msgs, uint8_t *digests);
int h2v_probe_g1_decompress(int device, uint32_t n, const uint8_t *compressed, uint8_t *xy_be, uint8_t *valid);
/* sum_t s_t * B_t per group: n groups of T terms; scalars n*T*32 B LE, bases n*T*48 B compressed; out n*96 B affine BE */
int h2v_probe_g1_msm(int device, uint32_t n, uint32_t T, const uint8_t *scalars, const uint8_t *bases_compressed,
                     uint8_t *out_xy_be);
/* the fixed-base launch of a split MSM on its own: sum over the plan's n_fix VK-base terms of s_t * B_t through the all-window
 * tables built at h2v_plan_load (k_g1_msm_fixed, bases_per_lane = 1 .. 4 as the launcher would pick); scalars n * n_fix * 32 B
 * LE in the order of the plan's VK-base terms, out n * 96 B affine BE; *n_fix_out (may be NULL) receives n_fix.  H2V_E_ARG for
 * plans without such tables (recursive plans).  Edge scalars of the signed-window recoding: tests/test_gpu_parity.py */
int h2v_probe_g1_msm_fixed(const h2v_plan *plan, uint32_t n, uint32_t bases_per_lane, const uint8_t *scalars, uint8_t *out_xy_be,
                           uint32_t *n_fix_out);
/* the bucket (Pippenger) MSM of the RLC mode on its own: sum_n s_n * B_n; scalars n*32 B LE (< r), bases n*48 B compressed
 * (encodings that do not decompress count as infinity); out 96 B affine BE (all-zero = infinity) */
int h2v_probe_g1_msm_pippenger(int device, uint32_t n, const uint8_t *scalars, const uint8_t *bases_compressed,
                               uint8_t *out_xy_be);
/* 2P + (neg ? -Q : Q) by the one-lane mixed addition and by the quad-cooperative one (the forced MSM shape H2V_MSM_LPT=8):
 * pq = x_P, y_P, x_Q, y_Q as 12 canonical LE dwords each; out = 5 x 42 dwords of lazily reduced limbs (X, Y, Z; 14 limbs of
 * 28 bits each, Montgomery form R = 2^392): the one-lane result, then lanes 0..3 of a quad.  Guards a compiler issue
 * (csrc/h2v_curve28.hpp: g1j28_madd_quad). */
int h2v_probe_quad_madd(int device, const uint32_t *pq, int neg, uint32_t *out);
/* e(p1, s_g2 of plan) == e(p2, G2) for n pairs of compressed G1 points; out[i] = 1/0 */
int h2v_probe_pairing(const h2v_plan *plan, uint32_t n, const uint8_t *p1_compressed, const uint8_t *p2_compressed,
                      uint8_t *out);
/* same with an explicit kernel (impl 0: one lane per proof, 1: cooperative - 32 lanes per proof, or the wide engine for small n,
 * as the launcher picks -, 2 / 3 / 5: its narrow (16 lanes per proof) / wide (64) / six-lane engine whatever n, -1: default) and an
 * optional dump (n * 24 * 48 bytes): the 12 Fp coefficients (flat order w^k, re/im; canonical LE) of f after the
 * Miller loop and, for the cooperative kernel, after the final exponentiation */
int h2v_probe_pairing_ex(const h2v_plan *plan, uint32_t n, const uint8_t *p1_compressed, const uint8_t *p2_compressed,
                         uint8_t *out, int impl, uint8_t *dbg);

/* ---- library lifecycle (round 4) ---------------------------------------------------------------------------------
 * h2v_shutdown(device) - device = -1: every device - quiesces and releases everything the library owns there: it waits
 * for the library's pool streams (the hardware queues the lanes run on), releases every live workspace of that device
 * (buffers, events, copy streams) and the device memory of every plan loaded on it, and destroys the pool streams.
 * Afterwards every entry point that needs that device returns H2V_E_DEVICE; workspace / plan handles stay valid as empty
 * shells (h2v_plan_info still answers) and are freed with the usual h2v_*_free.  Idempotent; not to be called while
 * another thread is inside a verify call.
 * Who calls it: a host calls it once, after its last verify call and before the process starts to exit - the Rust side in
 * the `Drop` of the object that owns the library, h2v.hpp through h2v::shutdown(), the Python binding
 * with atexit.  The reference's side is RAII throughout (the guard is consumed by value, examples/simple_mul.rs:98-104);
 * this is the one piece of process-wide state the backend adds.  As a backstop the library registers the call with
 * atexit() itself when it creates its first pool stream (i.e. after the HIP runtime has initialised, so that it runs
 * before the runtime's own exit handlers).  Why it exists: pool streams that were still alive when the HIP runtime and a
 * profiler's tool library ran their static destructors crashed the process at exit (SIGSEGV through __cxa_finalize under
 * rocprofv3, round 3). */
int h2v_shutdown(int device);

const char *h2v_last_error(void);
/* sha256 (hex) of the sources this binary was built from; "unknown" for a build outside __graft_entry__.build() */
const char *h2v_build_id(void);
int h2v_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
