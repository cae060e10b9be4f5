// C-ABI of the MI355X Halo2/KZG batch verifier (include/h2v.h).  Host side: plan parsing/validation/upload,
// workspace management, stream orchestration of the four kernels.  No arithmetic happens on the host and there
// is no CPU fallback: a missing or failing device is an error (H2V_E_DEVICE), never a silent detour.
#include "../../include/h2v.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "h2v_kernels.hip"
#include "h2v_plancc.hpp"

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    if (code == H2V_E_DEVICE && msg.find("hip: ") == std::string::npos) {      // what the runtime said last on this thread, if anything
        const hipError_t e = hipPeekAtLastError();
        if (e != hipSuccess) g_err += std::string(" [hip: ") + hipGetErrorString(e) + "]";
    }
    return code;
}
#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail(H2V_E_DEVICE, std::string(#expr) + ": hip: " + hipGetErrorString(e_));                \
    } while (0)

extern "C" const char *h2v_last_error(void) { return g_err.c_str(); }

// ---------------------------------------------------------------------------------------------- lifecycle
// Everything the library owns on a device - the pool of CU-mask streams (make_stream), every live workspace's buffers,
// streams and events, every plan's device memory - is released by h2v_shutdown(device), which a host calls before the
// process exits (backend.py: atexit; h2v.hpp: h2v::shutdown; the library itself registers it with atexit() when the first
// pool stream is created, i.e. after the HIP runtime's own initialisation, so that it runs BEFORE the runtime's exit
// handlers).  Why: a stream with a hardware queue of its own that is still alive when the runtime and a profiler's tool
// library run their static destructors was the exit-time SIGSEGV of round 3 (under rocprofv3, after "tool
// finalization", through __cxa_finalize: only the forms that had created pool streams crashed).  After the call every
// entry point that needs the device returns H2V_E_DEVICE; handles stay valid as empty shells and are freed as usual.
static std::mutex g_reg_mu;
static std::vector<h2v_workspace *> g_ws_live;
static std::vector<h2v_plan *> g_plan_live;
static bool g_shut[16] = {};
static bool dev_shut(int dev) {
    std::lock_guard<std::mutex> lock(g_reg_mu);
    return dev >= 0 && dev < 16 && g_shut[dev];
}
#define ALIVE_DEV(dev)                                                                                   \
    do {                                                                                                 \
        if (dev_shut(dev)) return fail(H2V_E_DEVICE, "h2v_shutdown has been called for this device");    \
    } while (0)
#define ALIVE(obj)                                                                                       \
    do {                                                                                                 \
        if ((obj) && (obj)->dead) return fail(H2V_E_DEVICE, "h2v_shutdown has been called: this handle is an empty shell"); \
    } while (0)
template <class T> static void reg_add(std::vector<T *> &v, T *x) { std::lock_guard<std::mutex> lock(g_reg_mu); v.push_back(x); }
template <class T> static void reg_del(std::vector<T *> &v, T *x) {
    std::lock_guard<std::mutex> lock(g_reg_mu);
    for (size_t i = 0; i < v.size(); i++) if (v[i] == x) { v[i] = v.back(); v.pop_back(); return; }
}
// content hash of csrc/ + include/h2v.h this binary was built from (__graft_entry__.build_hip compares it with the tree)
#ifndef H2V_SRC_HASH_STR
#define H2V_SRC_HASH_STR "unknown"
#endif
extern "C" const char *h2v_build_id(void) {
    static const char id[] = "H2V_SRC_HASH=" H2V_SRC_HASH_STR;
    return id + 13;
}
extern "C" int h2v_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------------------------- plan compiler
// VerifyingKey description (JSON, docs/vk_schema.json) -> plan blob, on the host: no GPU is needed for this call.
extern "C" int h2v_plan_compile(const char *vk_json, size_t len, uint8_t **blob_out, size_t *blob_len) {
    if (!vk_json || !blob_out || !blob_len) return fail(H2V_E_ARG, "null argument");
    *blob_out = nullptr; *blob_len = 0;
    try {
        const h2vplan::VK vk = h2vplan::parse_vk(vk_json, len);
        const std::vector<uint8_t> blob = h2vplan::compile_plan(vk);
        uint8_t *out = (uint8_t *)malloc(blob.size());
        if (!out) return fail(H2V_E_DEVICE, "out of memory");
        memcpy(out, blob.data(), blob.size());
        *blob_out = out; *blob_len = blob.size();
        return H2V_OK;
    } catch (const h2vplan::CompileError &e) {
        return fail(H2V_E_ARG, std::string("verifying-key description: ") + e.what());
    } catch (const std::exception &e) {
        return fail(H2V_E_ARG, std::string("plan compiler: ") + e.what());
    }
}
extern "C" void h2v_blob_free(uint8_t *blob) { free(blob); }

struct h2v_plan {
    int device = 0;
    bool dead = false;         // h2v_shutdown released the device memory: every call with this plan is H2V_E_DEVICE
    H2vDevPlan d{};            // device view
    void *blob = nullptr;      // one device allocation holding every section
    void *fold_terms = nullptr;  // recursion: the 4-entry term table of the two fold MSMs
    void *vk_tab = nullptr;      // window tables of the VK bases (k_vk_tables at load)
    void *fix_tab = nullptr;     // all-window tables of the VK bases (k_vk_fixed_tables at load; non-recursive plans)
    uint32_t n_var = 0, n_fix = 0;  // per-proof terms [0, n_var), VK-base terms [n_var, n_var + n_fix) when the list is so ordered
    uint32_t n_squeezes = 0, stream_len = 0;
    uint64_t gen = 0;            // process-wide load counter: what caches key on (a freed plan's address may be reused)
    std::vector<uint32_t> trace_slots;
};

struct h2v_workspace {
    int device = 0;
    bool dead = false;             // h2v_shutdown released everything: an empty shell until h2v_workspace_free
    uint32_t in_flight_hint = 1;   // h2v_workspace_hint_in_flight: how many batches the caller keeps in flight on this device
    bool hint_given = false;       // (laned: the caller said so itself; otherwise run_laned estimates it per call)
    uint64_t cap = 0;       // max batch
    uint32_t stride = 0;    // register-file stride (cap rounded up to 64)
    // what the buffers were sized for (the creating plan's shape): a plan fits iff each of its values is <= these
    uint32_t sz_terms = 0, sz_slots = 0, sz_regs = 0 /* 0: register file in LDS */, sz_trace = 0;
    bool sz_ivc = false, sz_fix = false;
    uint32_t *regs = nullptr, *scalars = nullptr, *pts = nullptr, *er = nullptr, *status = nullptr, *trace = nullptr, *msm_tab = nullptr;
    // recursion (IVC): acc_left / acc_right_final sums, the fold's points + scalars, and the folded el / er
    uint32_t *accl = nullptr, *accr = nullptr, *fold_pts = nullptr, *fold_scal = nullptr, *el2 = nullptr, *er2 = nullptr;
    uint32_t *pt_tab = nullptr;  // MSM window tables of every per-proof point, written by the decompression kernel
    uint32_t *er_fix = nullptr;  // sum of the VK-base terms when the MSM is split into a ladder and a fixed-base launch
    uint32_t *dec_ctr = nullptr; // work-queue counters of the decompression launch (one per pipeline chunk)
    uint8_t *valid = nullptr, *valid_sub = nullptr, *accept = nullptr;
    // Host-buffer entry points: the batch is packed into ONE pinned host block (offsets | instances | committed | proofs),
    // uploaded with one asynchronous copy on the workspace's own stream `hs`, verified there, and the accept bytes come
    // back into pinned memory the same way: h2v_verify_batch_submit returns once everything is enqueued, _wait collects.
    // in_* point into the device block.
    uint8_t *in_block = nullptr, *h_block = nullptr, *h_accept = nullptr;
    size_t in_block_cap = 0, h_accept_cap = 0;
    uint8_t *in_proofs = nullptr, *in_inst = nullptr, *in_ci = nullptr;
    uint64_t *in_off = nullptr;
    hipStream_t hs = nullptr;
    hipEvent_t ev_host = nullptr;
    bool pending = false, pending_rlc = false;
    uint64_t pending_n = 0;
    // A batch is cut into up to MAXP chunks that run as independent pipelines on their own stream pairs, so that one
    // chunk's decompression / transcript kernels (few waves) overlap another chunk's MSM / pairing kernels.
    static constexpr int MAXP = 4;
    hipStream_t pmain[MAXP] = {}, pside[MAXP] = {}, psub[MAXP] = {};
    hipEvent_t ev_fork = nullptr, ev_join[MAXP] = {}, ev_sub[MAXP] = {}, ev_fix[MAXP] = {}, ev_done[MAXP] = {};
    // ring of per-call, per-chunk event sets: [0]/[1] around the transcript+combiner kernel, [2]/[3] around the
    // decompression kernel's square-root half (side stream), [4]/[5] around the MSM, [5]/[6] around the pairing kernel,
    // [7]/[8] around the decompression kernel's subgroup half (third stream)
    static constexpr int RING = 64, NEV = 12;   // (+ [9]/[10] around the fixed-base MSM launch, [11] the end of the ladder launch beside it)
    hipEvent_t ring[RING][MAXP][NEV] = {};
    uint8_t ring_pipes[RING] = {}, ring_split[RING] = {}, ring_lpt[RING] = {}, ring_pair[RING] = {}, ring_var[RING] = {};
    uint64_t calls = 0;
    struct RlcWs *rlc = nullptr;   // buffers of the RLC batch mode, created by its first call
    std::vector<struct RlcWs *> rlc_parked;   // the same for OTHER plans this workspace has served in that mode (rlc_ensure swaps; never freed before the workspace)
    int32_t opt[H2V_OPT_COUNT] = {};   // h2v_workspace_set_option / h2v_workspace_tune: 0 = the launcher's choice
    int one_stream_mode = -1;      // lanes: 1 = the whole pipeline on the stream it is given (-1: decided from the hint)
    // ---- lanes (h2v_workspace_create_lanes): a laned workspace owns no kernel buffers of its own, only n_lanes ordinary
    // workspaces of `chunk` proofs and one library-owned stream per lane.  A verify call is cut into chunks that go round
    // robin through the lanes; chunk c of a call runs entirely on lane_st[lane of c], behind that lane's earlier chunks.
    static constexpr int MAXL = 16;
    uint32_t n_lanes = 0, chunk = 0;
    h2v_workspace *lane[MAXL] = {};
    hipStream_t lane_st[MAXL] = {};
    hipEvent_t lane_ev[MAXL] = {};         // end of the lane's most recent chunk
    bool lane_busy[MAXL] = {};             // work enqueued since the last join
    bool defer_joins = false;
    uint32_t lanes_per_proof = 0;           // lanes the per-proof mode cycles through (the RLC mode uses all n_lanes)
    uint64_t next_lane = 0;                 // round-robin position (persists across calls: consecutive calls interleave)
    H2vDevPlan lane_plan{};                 // the creating plan's shape: lanes are created when first used
    // host-buffer batches in flight on a laned workspace (h2v_verify_batch_submit / _wait): a ring of staging slots, each
    // with its own pinned block, device block and accept buffers; uploads on `hs`, downloads on `hs_down`
    struct HostSlot {
        uint8_t *in_block = nullptr, *h_block = nullptr, *h_accept = nullptr, *d_accept = nullptr;
        size_t in_cap = 0, acc_cap = 0;
        hipEvent_t ev = nullptr;
        uint64_t n = 0, call = 0;
        bool rlc = false;
    };
    static constexpr int MAXH = 16;
    HostSlot hslot[MAXH];
    uint64_t h_head = 0, h_tail = 0;
    hipStream_t hs_down = nullptr;
    bool copy_streams_owned = false;        // laned: hs / hs_down carry copies only and are plain streams of this workspace (host_stream)
    uint32_t *rlc_fail = nullptr;           // RING counters: failed batch checks among the chunks of a call (RLC mode)
    uint32_t *rlc_fail_ptr = nullptr;       // (a lane: where its batch check reports a failure; set by the parent per call)
    // routing of RLC calls by what earlier calls met (rlc_route): cumulative device counters [groups seen, groups failed], a
    // pinned host mirror refreshed behind every call, and the running estimate of the failing-group rate
    uint32_t *rlc_stats = nullptr, *rlc_stats_ptr = nullptr, *h_rlc_stats = nullptr;
    uint32_t seen_groups = 0, seen_failed = 0;
    float fail_rate = 0.0f;
    bool routed = false;                    // the most recent RLC call ran the per-proof kernels directly
    uint8_t lring_routed[64] = {};          // (per call slot)
    // per call (ring): number of chunks, first lane, and every lane's call counters when the call had been enqueued -
    // what h2v_workspace_timings / _rlc_result need to find the chunks' event sets in the lanes' own rings
    uint32_t lring_chunks[RING] = {}, lring_first[RING] = {}, lring_mod[RING] = {};
    uint64_t lring_calls[RING][MAXL] = {}, lring_rlc_calls[RING][MAXL] = {};
    const struct RlcWs *lring_rlc_obj[RING][MAXL] = {};   // (whose counter lring_rlc_calls holds: a lane serves one plan's RLC buffers at a time)
    uint8_t lring_rlc[RING] = {};
    // ---- coalescing of small device-resident calls (coalesce_call): per lane a staging area the proofs of several calls are
    // gathered into; co_open: the lanes whose groups are open, oldest first (one group per plan and mode, at most four); per call
    // slot: coalesced?, its share of the group
    struct Coalesce *co[MAXL] = {};
    std::vector<uint32_t> co_open;
    uint8_t lring_co[RING] = {};
    float lring_share[RING] = {};
};

// LDS left for the combiner's register file in a block: 160 KB minus the 8 KB hash buffer and 1 KB of slack
// (plan.py: VM_LDS_BYTES)
#define H2V_VM_LDS_BYTES ((size_t)160 * 1024 - 8192 - 1024)
static uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
// ---------------------------------------------------------------------------------------------- plan
extern "C" int h2v_plan_load(const uint8_t *blob, size_t len, int device, h2v_plan **out) { return h2v_plan_load_ex(blob, len, device, nullptr, out); }
extern "C" int h2v_plan_load_ex(const uint8_t *blob, size_t len, int device, const h2v_plan_opts *opts, h2v_plan **out) {
    if (!blob || !out) return fail(H2V_E_ARG, "null argument");
    const uint32_t opt_fix_c = opts ? opts->fixed_base_window_bits : 0u;
    if (opt_fix_c != 0 && opt_fix_c != 4 && opt_fix_c != 8 && opt_fix_c != 12) return fail(H2V_E_ARG, "fixed_base_window_bits: 0 (auto), 4, 8 or 12");
    *out = nullptr;
    const size_t hdr = 8 + 4 * H2V_PLAN_HDR_WORDS;
    if (len < hdr || memcmp(blob, H2V_PLAN_MAGIC, 8) != 0) return fail(H2V_E_PLAN, "bad magic / truncated header");
    uint32_t w[H2V_PLAN_HDR_WORDS];
    for (int i = 0; i < H2V_PLAN_HDR_WORDS; i++) w[i] = rd32(blob + 8 + 4 * i);
    if (w[H2V_HW_VERSION] != H2V_PLAN_VERSION) return fail(H2V_E_PLAN, "unsupported plan version");
    if (w[H2V_HW_TOTAL_LEN] != len) return fail(H2V_E_PLAN, "length mismatch");
    struct Sec { int off_word; uint64_t bytes; };
    const uint32_t n_instr = w[H2V_HW_N_INSTR], n_consts = w[H2V_HW_N_CONSTS], n_points = w[H2V_HW_N_POINTS],
                   n_bases = w[H2V_HW_N_VK_BASES], n_terms = w[H2V_HW_N_TERMS], n_trace = w[H2V_HW_N_TRACE],
                   n_regs = w[H2V_HW_N_REGS], proof_len = w[H2V_HW_PROOF_LEN], n_pi = w[H2V_HW_N_PI], n_ci = w[H2V_HW_N_CI];
    const Sec secs[] = {{H2V_HW_OFF_INSTR, 8ull * n_instr}, {H2V_HW_OFF_CONSTS, 32ull * n_consts}, {H2V_HW_OFF_POINTS, 4ull * n_points},
                        {H2V_HW_OFF_VK_BASES, 96ull * n_bases}, {H2V_HW_OFF_TERMS, 8ull * n_terms},
                        {H2V_HW_OFF_LINES_SG2, 192ull * H2V_MILLER_LINES}, {H2V_HW_OFF_LINES_G2, 192ull * H2V_MILLER_LINES},
                        {H2V_HW_OFF_TRACE, 8ull * n_trace},
                        {H2V_HW_OFF_LINES28_SG2, 512ull * H2V_MILLER_LINES}, {H2V_HW_OFF_LINES28_G2, 512ull * H2V_MILLER_LINES}};
    for (const Sec &s : secs) {
        const uint64_t off = w[s.off_word];
        if (off < hdr || (off & 15) || off + s.bytes > len) return fail(H2V_E_PLAN, "section out of bounds");
    }
    if (n_instr == 0 || n_instr > (1u << 20) || n_regs == 0 || n_regs > 65535 || n_points == 0 || n_points > 4096 ||
        n_terms == 0 || n_ci > 1 || n_pi > (1u << 16) || proof_len > (1u << 24))
        return fail(H2V_E_PLAN, "implausible counts");
    const uint32_t ivc = w[H2V_HW_IVC], n_main = w[H2V_HW_N_MAIN_TERMS];
    if (ivc > 1 || n_main == 0 || n_main > n_terms || (!ivc && n_main != n_terms) || (ivc && n_terms < n_main + 2))
        return fail(H2V_E_PLAN, "inconsistent recursion header");
    if (n_main > 64 || (ivc && n_terms - n_main - 1 > 64))
        return fail(H2V_E_LIMIT, "more than 64 MSM terms per sum is not supported by this backend");
    if (ivc)
        for (int k = 0; k < 8; k++)
            if (w[H2V_HW_ACC_IDX0 + k] >= n_pi) return fail(H2V_E_PLAN, "accumulator public-input index out of range");
    if (w[H2V_HW_PI_POINT] >= n_points) return fail(H2V_E_PLAN, "pi point index out of range");
    // validate the program(s): every register / constant / offset the kernels will touch is in range, and the bundle
    // discipline the multi-lane interpreter relies on holds (h2v_plan.h)
    uint32_t n_sq = 0;
    auto check_program = [&](const uint8_t *ip, uint32_t n_rec, uint32_t n_regs_v, uint32_t lanes, uint32_t *squeezes) -> int {
        if (lanes == 0 || lanes > 32 || (lanes & (lanes - 1)) || n_rec % lanes) return fail(H2V_E_PLAN, "bad lane count of the program");
        bool has_end = false;
        uint32_t sq = 0;
        for (uint32_t base_k = 0; base_k < n_rec && !has_end; base_k += lanes) {
            uint32_t wr[32], n_wr = 0;
            bool lane0_serial = false;
            for (uint32_t l = 0; l < lanes; l++) {
                const uint32_t k = base_k + l;
                const uint8_t op = ip[8 * k];
                const uint32_t dst = ip[8 * k + 2] | (ip[8 * k + 3] << 8), a = ip[8 * k + 4] | (ip[8 * k + 5] << 8), b = ip[8 * k + 6] | (ip[8 * k + 7] << 8);
                const uint32_t off = a | (b << 16);
                bool ok = true, writes = false, serial = false;
                switch (op) {
                case H2V_OP_END: has_end = true; serial = true; break;
                case H2V_OP_NOP: break;
                case H2V_OP_ABSORB_REG: ok = a < n_regs_v; serial = true; break;
                case H2V_OP_ABSORB_CI: ok = n_ci == 1; serial = true; break;
                case H2V_OP_LOAD_INSTANCE: ok = dst < n_regs_v && a < n_pi; writes = true; break;
                case H2V_OP_READ_POINT: ok = (uint64_t)off + 48 <= proof_len; serial = true; break;
                case H2V_OP_READ_SCALAR: ok = dst < n_regs_v && (uint64_t)off + 32 <= proof_len; writes = true; serial = true; break;
                case H2V_OP_SQUEEZE: ok = dst < n_regs_v; sq++; writes = true; serial = true; break;
                case H2V_OP_CONST: ok = dst < n_regs_v && a < n_consts; writes = true; break;
                case H2V_OP_ADD: case H2V_OP_SUB: case H2V_OP_MUL: ok = dst < n_regs_v && a < n_regs_v && b < n_regs_v; writes = true; break;
                case H2V_OP_NEG: case H2V_OP_INV: ok = dst < n_regs_v && a < n_regs_v; writes = true; break;
                case H2V_OP_OUT_SCALAR: ok = dst < n_terms && a < n_regs_v; break;
                case H2V_OP_ASSERT_ZERO: ok = a < n_regs_v; break;
                default: ok = false;
                }
                if (!ok) return fail(H2V_E_PLAN, "instruction " + std::to_string(k) + " out of range");
                if (serial && l != 0) return fail(H2V_E_PLAN, "transcript operation off lane 0 (record " + std::to_string(k) + ")");
                if (l == 0) lane0_serial = serial;
                else if (lane0_serial && op != H2V_OP_NOP) return fail(H2V_E_PLAN, "transcript operation shares its bundle (record " + std::to_string(k) + ")");
                if (writes) wr[n_wr++] = (l << 16) | dst;
            }
            // independence inside the bundle: no register is written twice, none is read by a lane other than... any lane
            for (uint32_t x = 0; x < n_wr; x++) {
                const uint32_t reg = wr[x] & 0xffff;
                for (uint32_t y = x + 1; y < n_wr; y++)
                    if ((wr[y] & 0xffff) == reg) return fail(H2V_E_PLAN, "two lanes of a bundle write one register");
                for (uint32_t l = 0; l < lanes; l++) {
                    const uint32_t k = base_k + l;
                    const uint8_t op = ip[8 * k];
                    const uint32_t a = ip[8 * k + 4] | (ip[8 * k + 5] << 8), b = ip[8 * k + 6] | (ip[8 * k + 7] << 8);
                    const bool reads_a = op == H2V_OP_ABSORB_REG || op == H2V_OP_ADD || op == H2V_OP_SUB || op == H2V_OP_MUL || op == H2V_OP_NEG ||
                                         op == H2V_OP_INV || op == H2V_OP_OUT_SCALAR || op == H2V_OP_ASSERT_ZERO;
                    const bool reads_b = op == H2V_OP_ADD || op == H2V_OP_SUB || op == H2V_OP_MUL;
                    if (lanes > 1 && ((reads_a && a == reg) || (reads_b && b == reg)))
                        return fail(H2V_E_PLAN, "a bundle reads a register it writes (record " + std::to_string(k) + ")");
                }
            }
        }
        if (!has_end) return fail(H2V_E_PLAN, "program has no END");
        if (squeezes) *squeezes = sq;
        return H2V_OK;
    };
    auto end_is_last = [&](const uint8_t *ip, uint32_t n_rec, uint32_t lanes) {   // the interpreter's trip count is n_rec / lanes - 1
        for (uint32_t k = 0; k + lanes < n_rec; k++) if (ip[8 * k] == H2V_OP_END) return false;
        return ip[8 * (n_rec - lanes)] == H2V_OP_END;
    };
    const uint32_t vm_lanes = w[H2V_HW_VM_LANES];
    if (int rcp = check_program(blob + w[H2V_HW_OFF_INSTR], n_instr, n_regs, vm_lanes, &n_sq)) return rcp;
    if (!end_is_last(blob + w[H2V_HW_OFF_INSTR], n_instr, vm_lanes)) return fail(H2V_E_PLAN, "END must be the last bundle of the program");
    const uint32_t wide_lanes = w[H2V_HW_VM2_LANES], wide_n_regs = w[H2V_HW_VM2_N_REGS], wide_n_instr = w[H2V_HW_VM2_N_INSTR];
    if (wide_lanes) {
        const uint64_t off = w[H2V_HW_VM2_OFF_INSTR];
        if (off < hdr || (off & 15) || off + 8ull * wide_n_instr > len || wide_n_instr == 0 || wide_n_instr > (1u << 20) ||
            wide_n_regs == 0 || wide_n_regs > 65535)
            return fail(H2V_E_PLAN, "wide program section out of bounds");
        if (wide_lanes < 2 || (size_t)wide_n_regs * 32 * (64 / wide_lanes) > H2V_VM_LDS_BYTES)
            return fail(H2V_E_PLAN, "wide program does not fit the LDS register file");
        if (int rcp = check_program(blob + off, wide_n_instr, wide_n_regs, wide_lanes, nullptr)) return rcp;
        if (!end_is_last(blob + off, wide_n_instr, wide_lanes)) return fail(H2V_E_PLAN, "END must be the last bundle of the program");
    }
    if (vm_lanes > 1 && (size_t)n_regs * 32 * (64 / vm_lanes) > H2V_VM_LDS_BYTES)
        return fail(H2V_E_PLAN, "multi-lane program does not fit the LDS register file");
    const uint8_t *pp = blob + w[H2V_HW_OFF_POINTS];
    for (uint32_t k = 0; k < n_points; k++)
        if ((uint64_t)rd32(pp + 4 * k) + 48 > proof_len) return fail(H2V_E_PLAN, "point offset out of range");
    const uint8_t *tp = blob + w[H2V_HW_OFF_TERMS];
    for (uint32_t k = 0; k < n_terms; k++) {
        const uint32_t kind = rd32(tp + 8 * k), idx = rd32(tp + 8 * k + 4);
        const bool ok = (kind == H2V_TERM_PROOF_POINT && idx < n_points) || (kind == H2V_TERM_VK_BASE && idx < n_bases) ||
                        (kind == H2V_TERM_COMMITTED_INSTANCE && n_ci == 1) || (kind == H2V_TERM_ACC_POINT && ivc && idx < 2);
        if (!ok) return fail(H2V_E_PLAN, "MSM term out of range");
    }
    const uint8_t *trp = blob + w[H2V_HW_OFF_TRACE];
    for (uint32_t k = 0; k < n_trace; k++)
        if (rd32(trp + 8 * k + 4) >= n_regs) return fail(H2V_E_PLAN, "trace register out of range");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(H2V_E_DEVICE, "no HIP device: the HIP backend is required (no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(H2V_E_ARG, "device index out of range");
    ALIVE_DEV(device);
    HIPCHK(hipSetDevice(device));
    h2v_plan *p = new h2v_plan();
    p->device = device;
    if (hipMalloc(&p->blob, len) != hipSuccess) { delete p; return fail(H2V_E_DEVICE, "hipMalloc(plan) failed"); }
    // device copy of the blob with the committed-instance MSM terms rewritten to "per-proof slot n_points"
    // (the decompression kernel stores the committed instance there), so the MSM kernel sees two term kinds only
    std::vector<uint8_t> patched(blob, blob + len);
    for (uint32_t k = 0; k < n_terms; k++) {
        uint8_t *t = patched.data() + w[H2V_HW_OFF_TERMS] + 8 * k;
        if (rd32(t) == H2V_TERM_COMMITTED_INSTANCE) {
            const uint32_t kind = H2V_TERM_PROOF_POINT, idx = n_points;
            memcpy(t, &kind, 4);
            memcpy(t + 4, &idx, 4);
        } else if (rd32(t) == H2V_TERM_ACC_POINT) {   // the decompression kernel rebuilds the accumulator points there
            const uint32_t kind = H2V_TERM_PROOF_POINT, idx = n_points + n_ci + rd32(t + 4);
            memcpy(t, &kind, 4);
            memcpy(t + 4, &idx, 4);
        }
    }
    if (ivc) {
        const uint32_t ft[8] = {H2V_TERM_PROOF_POINT, 0, H2V_TERM_PROOF_POINT, 1, H2V_TERM_PROOF_POINT, 2, H2V_TERM_PROOF_POINT, 3};
        if (hipMalloc(&p->fold_terms, sizeof ft) != hipSuccess || hipMemcpy(p->fold_terms, ft, sizeof ft, hipMemcpyHostToDevice) != hipSuccess) {
            if (p->fold_terms) (void)hipFree(p->fold_terms);
            (void)hipFree(p->blob); delete p;
            return fail(H2V_E_DEVICE, "fold term upload failed");
        }
    }
    if (hipMemcpy(p->blob, patched.data(), len, hipMemcpyHostToDevice) != hipSuccess) { if (p->fold_terms) (void)hipFree(p->fold_terms); (void)hipFree(p->blob); delete p; return fail(H2V_E_DEVICE, "plan upload failed"); }
    const uint8_t *base = (const uint8_t *)p->blob;
    H2vDevPlan &d = p->d;
    d.proof_len = proof_len; d.n_pi = n_pi; d.n_ci = n_ci; d.n_regs = n_regs; d.n_instr = n_instr; d.n_consts = n_consts;
    d.n_points = n_points; d.n_vk_bases = n_bases; d.n_terms = n_terms; d.n_trace = n_trace; d.pi_point = w[H2V_HW_PI_POINT];
    d.instr = (const H2vInstr *)(base + w[H2V_HW_OFF_INSTR]);
    d.vm_lanes = vm_lanes;
    if (wide_lanes) {
        d.wide_lanes = wide_lanes; d.wide_n_regs = wide_n_regs; d.wide_n_instr = wide_n_instr;
        d.wide_instr = (const H2vInstr *)(base + w[H2V_HW_VM2_OFF_INSTR]);
    }
    d.consts = (const uint32_t *)(base + w[H2V_HW_OFF_CONSTS]);
    d.points = (const uint32_t *)(base + w[H2V_HW_OFF_POINTS]);
    d.vk_bases = (const uint32_t *)(base + w[H2V_HW_OFF_VK_BASES]);
    d.terms = (const uint32_t *)(base + w[H2V_HW_OFF_TERMS]);
    d.lines_sg2 = (const uint32_t *)(base + w[H2V_HW_OFF_LINES_SG2]);
    d.lines_g2 = (const uint32_t *)(base + w[H2V_HW_OFF_LINES_G2]);
    d.trace = (const uint32_t *)(base + w[H2V_HW_OFF_TRACE]);
    d.lines28_sg2 = (const uint32_t *)(base + w[H2V_HW_OFF_LINES28_SG2]);
    d.lines28_g2 = (const uint32_t *)(base + w[H2V_HW_OFF_LINES28_G2]);
    d.ivc = ivc; d.n_main_terms = n_main;
    for (int k = 0; k < 8; k++) d.acc_idx[k] = w[H2V_HW_ACC_IDX0 + k];
    d.fold_terms = (const uint32_t *)p->fold_terms;
    // window tables of the VK bases for the MSM ladder: [1..8]B and [1..8]phi(B), affine, computed once per plan
    if (hipMalloc(&p->vk_tab, (size_t)n_bases * 448 * 4 + 16) != hipSuccess || hipMemset(p->vk_tab, 0, (size_t)n_bases * 448 * 4 + 16) != hipSuccess) {
        if (p->vk_tab) (void)hipFree(p->vk_tab);
        if (p->fold_terms) (void)hipFree(p->fold_terms);
        (void)hipFree(p->blob); delete p;
        return fail(H2V_E_DEVICE, "hipMalloc(vk tables) failed");
    }
    if (n_bases) {
        hipLaunchKernelGGL(k_vk_tables, dim3((n_bases + 63) / 64), dim3(64), 0, nullptr, d.vk_bases, n_bases, (uint32_t *)p->vk_tab);
        if (hipDeviceSynchronize() != hipSuccess) {
            (void)hipFree(p->vk_tab);
            if (p->fold_terms) (void)hipFree(p->fold_terms);
            (void)hipFree(p->blob); delete p;
            return fail(H2V_E_DEVICE, "VK window-table kernel failed");
        }
    }
    d.vk_tab = (const uint32_t *)p->vk_tab;
    // fixed-base launches of the MSM (non-recursive plans whose VK-base terms are the tail of the term list, as plan.py
    // orders them): every window multiple of every VK base
    if (!ivc && n_bases) {
        uint32_t nv = 0;
        while (nv < n_terms && rd32(patched.data() + w[H2V_HW_OFF_TERMS] + 8 * nv) != H2V_TERM_VK_BASE) nv++;
        bool tail_ok = nv < n_terms;
        for (uint32_t k = nv; k < n_terms; k++) tail_ok = tail_ok && rd32(patched.data() + w[H2V_HW_OFF_TERMS] + 8 * k) == H2V_TERM_VK_BASE;
        if (tail_ok) {
            // window width of the tables: 12 bits (22 additions per VK term in every proof, 5 MB per base) unless that would
            // take more than 2 GB; h2v_plan_opts.fixed_base_window_bits = 4 / 8 / 12 forces it (tests run all three)
            uint32_t fc = (size_t)n_bases * 22 * 2048 * 112 <= ((size_t)2 << 30) ? 12u : 8u;
            if (opt_fix_c) fc = opt_fix_c;
            const uint32_t fW = fc == 4 ? 65u : fc == 8 ? 33u : 22u, fE = 1u << (fc - 1);
            const size_t fix_bytes = (size_t)n_bases * fW * fE * 28 * 4;
            void *wbase = nullptr;
            bool okf = hipMalloc(&p->fix_tab, fix_bytes + 16) == hipSuccess && hipMemset(p->fix_tab, 0, fix_bytes + 16) == hipSuccess &&
                       hipMalloc(&wbase, (size_t)n_bases * fW * 96 + 16) == hipSuccess;
            if (okf) {
                hipLaunchKernelGGL(k_vk_fixed_window_bases, dim3((n_bases * fW + 63) / 64), dim3(64), 0, nullptr, d.vk_bases, n_bases, fc, fW, (uint32_t *)wbase);
                const uint64_t entries = (uint64_t)n_bases * fW * fE;
                hipLaunchKernelGGL(k_vk_fixed_tables, dim3((uint32_t)((entries + 63) / 64)), dim3(64), 0, nullptr, (const uint32_t *)wbase, n_bases * fW, fc, (uint32_t *)p->fix_tab);
                okf = hipDeviceSynchronize() == hipSuccess;
            }
            if (wbase) (void)hipFree(wbase);
            d.fix_c = fc; d.fix_W = fW;
            if (!okf) {
                if (p->fix_tab) (void)hipFree(p->fix_tab);
                (void)hipFree(p->vk_tab);
                (void)hipFree(p->blob); delete p;
                return fail(H2V_E_DEVICE, "fixed-base table setup failed");
            }
            p->n_var = nv;
            p->n_fix = n_terms - nv;
        }
    }
    d.fix_tab = (const uint32_t *)p->fix_tab; d.n_var = p->n_var; d.n_fix = p->n_fix;
    p->n_squeezes = n_sq;
    p->stream_len = w[H2V_HW_STREAM_LEN];
    for (uint32_t k = 0; k < n_trace; k++) p->trace_slots.push_back(rd32(trp + 8 * k));
    { static std::mutex mu; static uint64_t loads = 0; std::lock_guard<std::mutex> lock(mu); p->gen = ++loads; }
    reg_add(g_plan_live, p);
    *out = p;
    return H2V_OK;
}
static void plan_release(h2v_plan *p) {   // the device memory; the host-side facts (h2v_plan_info) stay
    (void)hipSetDevice(p->device);
    if (p->blob) (void)hipFree(p->blob);
    if (p->fold_terms) (void)hipFree(p->fold_terms);
    if (p->vk_tab) (void)hipFree(p->vk_tab);
    if (p->fix_tab) (void)hipFree(p->fix_tab);
    p->blob = p->fold_terms = p->vk_tab = p->fix_tab = nullptr;
}
extern "C" void h2v_plan_free(h2v_plan *p) {
    if (!p) return;
    reg_del(g_plan_live, p);
    if (!p->dead) plan_release(p);
    delete p;
}
extern "C" int h2v_plan_info(const h2v_plan *p, uint32_t *proof_len, uint32_t *n_pi, uint32_t *n_ci, uint32_t *n_terms) {
    if (!p) return fail(H2V_E_ARG, "null plan");
    if (proof_len) *proof_len = p->d.proof_len;
    if (n_pi) *n_pi = p->d.n_pi;
    if (n_ci) *n_ci = p->d.n_ci;
    if (n_terms) *n_terms = p->d.n_terms;
    return H2V_OK;
}
extern "C" int h2v_plan_trace_slots(const h2v_plan *p, uint32_t *slot_ids, uint32_t cap, uint32_t *n_out) {
    if (!p || !n_out) return fail(H2V_E_ARG, "null argument");
    *n_out = (uint32_t)p->trace_slots.size();
    for (uint32_t k = 0; k < cap && k < p->trace_slots.size(); k++) slot_ids[k] = p->trace_slots[k];
    return H2V_OK;
}

// ---------------------------------------------------------------------------------------------- workspace
static void rlc_release(struct RlcWs *r);
static hipError_t make_stream(hipStream_t *s);
static void co_release(h2v_workspace *w);
static int co_flush(h2v_workspace *w);
static int lane_streams(uint32_t l, hipStream_t *main_st, hipStream_t *side_st);
static void ws_release(h2v_workspace *w) {
    (void)hipSetDevice(w->device);
    if (w->hs_down) (void)hipStreamSynchronize(w->hs_down);
    for (auto &sl : w->hslot) {
        if (sl.in_block) (void)hipFree(sl.in_block);
        if (sl.d_accept) (void)hipFree(sl.d_accept);
        if (sl.h_block) (void)hipHostFree(sl.h_block);
        if (sl.h_accept) (void)hipHostFree(sl.h_accept);
        if (sl.ev) (void)hipEventDestroy(sl.ev);
        sl = h2v_workspace::HostSlot();
    }
    for (uint32_t l = 0; l < (uint32_t)h2v_workspace::MAXL; l++)
        if (w->lane_st[l]) { (void)hipStreamSynchronize(w->lane_st[l]); }
    co_release(w);                        // (an open group is dropped: whoever frees a workspace has joined it or no longer wants the verdicts)
    for (uint32_t l = 0; l < (uint32_t)h2v_workspace::MAXL; l++) {
        if (w->lane[l]) { ws_release(w->lane[l]); delete w->lane[l]; w->lane[l] = nullptr; }
        if (w->lane_ev[l]) (void)hipEventDestroy(w->lane_ev[l]);
    }
    w->n_lanes = 0;
    if (w->rlc) { rlc_release(w->rlc); w->rlc = nullptr; }
    for (struct RlcWs *r : w->rlc_parked) rlc_release(r);
    w->rlc_parked.clear();
    if (w->h_rlc_stats) (void)hipHostFree(w->h_rlc_stats);
    void *ptrs[] = {w->rlc_stats, w->rlc_fail, w->regs, w->scalars, w->pts, w->er, w->status, w->trace, w->valid, w->valid_sub, w->er_fix, w->dec_ctr, w->accept, w->in_block, w->msm_tab,
                    w->accl, w->accr, w->fold_pts, w->fold_scal, w->el2, w->er2, w->pt_tab};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    if (w->h_block) (void)hipHostFree(w->h_block);
    if (w->h_accept) (void)hipHostFree(w->h_accept);
    if (w->hs) (void)hipStreamSynchronize(w->hs);   // (pool streams are shared: make_stream; h2v_shutdown destroys them)
    if (w->copy_streams_owned) {
        if (w->hs) (void)hipStreamDestroy(w->hs);
        if (w->hs_down) (void)hipStreamDestroy(w->hs_down);
        w->hs = w->hs_down = nullptr; w->copy_streams_owned = false;
    }
    if (w->ev_host) (void)hipEventDestroy(w->ev_host);
    for (hipStream_t q : w->pmain) if (q) (void)hipStreamSynchronize(q);
    for (hipStream_t q : w->pside) if (q) (void)hipStreamSynchronize(q);
    for (hipStream_t q : w->psub) if (q) (void)hipStreamSynchronize(q);
    for (hipEvent_t e : w->ev_sub) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : w->ev_fix) if (e) (void)hipEventDestroy(e);
    if (w->ev_fork) (void)hipEventDestroy(w->ev_fork);
    for (hipEvent_t e : w->ev_join) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : w->ev_done) if (e) (void)hipEventDestroy(e);
    for (auto &call : w->ring) for (auto &set : call) for (hipEvent_t e : set) if (e) (void)hipEventDestroy(e);
}
static uint32_t vm_lds_slots(const H2vDevPlan &d);
static int ws_create_for(const H2vDevPlan &d, int device, uint64_t max_batch, bool with_trace, h2v_workspace **out) {
    if (max_batch == 0 || max_batch > (1ull << 24)) return fail(H2V_E_ARG, "max_batch out of range");
    ALIVE_DEV(device);
    HIPCHK(hipSetDevice(device));
    h2v_workspace *w = new h2v_workspace();
    w->device = device;
    w->cap = max_batch;
    w->stride = (uint32_t)((max_batch + 63) / 64 * 64);
    const uint64_t slots = H2V_SLOTS(d);
    w->sz_terms = d.n_terms; w->sz_slots = (uint32_t)slots; w->sz_regs = vm_lds_slots(d) == 0 ? d.n_regs : 0;
    w->sz_trace = with_trace ? d.n_trace : 0; w->sz_ivc = d.ivc != 0; w->sz_fix = d.fix_tab != nullptr;
#define WSALLOC(field, bytes)                                                                  \
    if (hipMalloc((void **)&w->field, (bytes)) != hipSuccess) { ws_release(w); delete w; return fail(H2V_E_DEVICE, "hipMalloc(" #field ") failed"); }
    if (vm_lds_slots(d) == 0) { WSALLOC(regs, (size_t)d.n_regs * 8 * w->stride * 4) }  // else the register file lives in LDS
    WSALLOC(scalars, (size_t)max_batch * d.n_terms * 32)
    WSALLOC(pts, (size_t)max_batch * slots * 96)
    WSALLOC(valid, (size_t)max_batch * slots)
    WSALLOC(valid_sub, (size_t)max_batch * slots)
    WSALLOC(dec_ctr, 64)
    if (d.fix_tab) { WSALLOC(er_fix, (size_t)max_batch * 36 * 4) }
    WSALLOC(er, (size_t)max_batch * 144)
    WSALLOC(pt_tab, (size_t)max_batch * slots * 448 * 4)             // per (proof, slot): [1..8]P and [1..8]phi(P), affine, 2 x 14 x 28-bit limbs
    if (d.ivc) { WSALLOC(msm_tab, (size_t)max_batch * 4 * 2 * 8 * 112) }  // fold MSMs build their four tables on the spot
    WSALLOC(status, (size_t)max_batch * 4)
    WSALLOC(accept, (size_t)max_batch)
    if (with_trace && d.n_trace) { WSALLOC(trace, (size_t)max_batch * d.n_trace * 32) }
    if (d.ivc) {
        WSALLOC(accl, (size_t)max_batch * 144)
        WSALLOC(accr, (size_t)max_batch * 144)
        WSALLOC(el2, (size_t)max_batch * 144)
        WSALLOC(er2, (size_t)max_batch * 144)
        WSALLOC(fold_pts, (size_t)max_batch * 4 * 96)
        WSALLOC(fold_scal, (size_t)max_batch * 4 * 32)
    }
#undef WSALLOC
    // Streams are created on first use (ws_streams): the runtime maps streams onto a few hardware queues round robin, and
    // streams that are never used would only make the ones in use collide (several batches in flight: bench.py --inflight).
    bool ok = hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < h2v_workspace::MAXP && ok; k++)
        ok = hipEventCreateWithFlags(&w->ev_sub[k], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&w->ev_fix[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&w->ev_join[k], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&w->ev_done[k], hipEventDisableTiming) == hipSuccess;
    for (auto &call : w->ring) for (auto &set : call) for (hipEvent_t &e : set)
        if (ok) ok = hipEventCreate(&e) == hipSuccess;
    if (!ok) { ws_release(w); delete w; return fail(H2V_E_DEVICE, "stream/event creation failed"); }
    *out = w;
    return H2V_OK;
}
// Chunk size of the lanes: the batch that gives the kernels of the per-proof pipeline about one wave per SIMD - the merged
// MSM runs one lane per (proof, term): 65536 / T proofs for 1024 waves (T = 16: 4096, the BASELINE batch; T = 60: 1092) -
// as a power of two between 1024 and 4096.
static uint32_t default_chunk(const H2vDevPlan &d) {
    const double want = 65536.0 / (double)(d.n_main_terms ? d.n_main_terms : 1);
    uint32_t c = 1024;
    while (c < 4096 && (double)c * 1.4142 < want) c <<= 1;
    return c;
}
// What a LANE holds (h2v_workspace_create_lanes with chunk = 0): 4096 proofs whatever the plan.  With several batches in flight
// the launch that pays is the large one - the six-lane pairing engine, ladders that share their doublings, fewer and longer
// launches - and since small calls are gathered (coalesce_call) a lane's capacity is no longer the size of the calls.  Measured,
// ms per call of the BASELINE size with the lanes at default_chunk -> 4096 (-> 8192): lookup_table x 2048 3.17 -> 2.92 (2.86),
// atms x 2048 3.30 -> 2.78 (2.70), sha256 shape x 1024 1.96 -> 1.86 (1.87), secp256k1 x 512 1.01 -> 0.87, their shares x 128 / x 64
// 0.32 / 0.17 -> 0.27 / 0.15; simple_mul x 4096 is at 4096 already (8192: the same).  A call that waits for its own chunks is
// still cut down to default_chunk / 2 per lane (run_laned).
static uint32_t lane_chunk(const H2vDevPlan &d) { (void)d; return 4096u; }
// Lanes: measured on MI355X, simple_mul, a stream of 4096-proof batches through one laned workspace (deferred joins; ms per
// batch): per-proof mode, each lane's pipeline on the lane's stream with the decompression beside it on a side stream:
// 4 lanes 4.60, 5: 4.35, 6: 4.31, 11: 4.72; everything on the lane's stream: 5: 4.64, 8: 4.26, 11: 4.43 (five caller-owned
// workspaces on fifteen streams, round 2's bench: 4.50); RLC mode (one stream per lane): 8 lanes 1.65, 11: 1.54, 16: 1.47.
// With the split MSM, the 12-bit fixed-base tables and the six-lane pairing engine the per-proof optimum moved to 8 lanes
// (= the whole pool of 16 streams; 9 lanes share streams: 4.27): simple_mul x 4096 6: 3.88, 7: 3.86, 8: 3.76; atms x 2048
// 3.92 -> 3.57; sha256 shape x 1024 2.36 -> 2.26, x 128 1.16 -> 0.93; secp256k1 shape x 64 1.08 -> 0.87 ms per batch.
#define H2V_DEFAULT_LANES 16u
#define H2V_PER_PROOF_LANES 8u
static int ensure_lane(h2v_workspace *w, uint32_t l) {
    if (w->lane[l]) return H2V_OK;
    // the lane's stream and event first, the lane itself last: a lane that exists has both (a failure here leaves lane[l]
    // NULL, and the next call tries again instead of enqueueing on a NULL stream)
    hipStream_t side_st = nullptr;
    if (int rcs = lane_streams(l, &w->lane_st[l], &side_st)) { w->lane_st[l] = nullptr; return rcs; }
    if (!w->lane_ev[l] && hipEventCreateWithFlags(&w->lane_ev[l], hipEventDisableTiming) != hipSuccess) { w->lane_ev[l] = nullptr; return fail(H2V_E_DEVICE, "lane event creation failed"); }
    h2v_workspace *lw = nullptr;
    int rc = ws_create_for(w->lane_plan, w->device, w->chunk, false, &lw);
    if (rc) return rc;
    lw->one_stream_mode = 2;            // (set per call: laned_depth)
    lw->pside[0] = side_st;             // (the decompression's stream in the two-stream form: the other half of the lane's pair)
    lw->in_flight_hint = w->in_flight_hint > w->n_lanes ? w->in_flight_hint : w->n_lanes;
    memcpy(lw->opt, w->opt, sizeof lw->opt);
    w->lane[l] = lw;
    return H2V_OK;
}
static int create_lanes_for(const H2vDevPlan &d, int device, uint64_t max_batch, uint32_t n_lanes, uint32_t chunk, h2v_workspace **out) {
    if (max_batch == 0 || max_batch > (1ull << 24)) return fail(H2V_E_ARG, "max_batch out of range");
    if (n_lanes > (uint32_t)h2v_workspace::MAXL) return fail(H2V_E_ARG, "at most 16 lanes");
    // chunk = 0, the library's choice: the plan's own chunk even when max_batch - the largest single call - is smaller: the lanes
    // are what small calls are gathered into (coalesce_call).  An explicit chunk is the caller's, cut to max_batch.
    if (chunk == 0) chunk = lane_chunk(d);
    else if ((uint64_t)chunk > max_batch) chunk = (uint32_t)max_batch;
    HIPCHK(hipSetDevice(device));
    h2v_workspace *w = new h2v_workspace();
    w->device = device; w->cap = max_batch; w->chunk = chunk;
    w->stride = (uint32_t)((max_batch + 63) / 64 * 64);
    w->lane_plan = d;
    w->sz_terms = d.n_terms; w->sz_slots = (uint32_t)H2V_SLOTS(d); w->sz_regs = vm_lds_slots(d) == 0 ? d.n_regs : 0; w->sz_trace = 0; w->sz_ivc = d.ivc != 0; w->sz_fix = d.fix_tab != nullptr;
    w->lanes_per_proof = n_lanes ? n_lanes : H2V_PER_PROOF_LANES;     // an explicit lane count holds for both modes
    w->n_lanes = n_lanes ? n_lanes : H2V_DEFAULT_LANES;
    w->in_flight_hint = w->n_lanes;
    bool ok = hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming) == hipSuccess && hipMalloc((void **)&w->accept, (size_t)max_batch) == hipSuccess &&
              hipMalloc((void **)&w->rlc_fail, h2v_workspace::RING * 4) == hipSuccess && hipMemset(w->rlc_fail, 0, h2v_workspace::RING * 4) == hipSuccess;
    // the first lane now (an allocation failure surfaces here, not in the middle of a verify call); the others on first use
    if (ok) ok = ensure_lane(w, 0) == H2V_OK;
    if (!ok) { const std::string e = g_err; ws_release(w); delete w; return fail(H2V_E_DEVICE, "lane creation failed: " + e); }
    reg_add(g_ws_live, w);
    *out = w;
    return H2V_OK;
}
extern "C" int h2v_workspace_create_lanes(const h2v_plan *p, uint64_t max_batch, uint32_t n_lanes, uint32_t chunk, h2v_workspace **out) {
    if (!p || !out) return fail(H2V_E_ARG, "null argument");
    ALIVE(p);
    return create_lanes_for(p->d, p->device, max_batch, n_lanes, chunk, out);
}
// One set of lanes for SEVERAL plans of one device (a node that verifies proofs of several circuits; BASELINE configs[2] names a
// mixed batch): the lane workspaces are sized for the largest of every dimension - MSM terms, point slots, the combiner's
// global register file where a plan needs one, the recursion and fixed-base buffers where a plan has them - so every listed plan
// passes ws_fits, and the calls of all of them go round robin through the SAME lanes and streams.  (One laned workspace per plan
// also works, but their lanes share the pool's sixteen streams pairwise: two plans x 2048 proofs, 6.70 ms per step on two
// workspaces.)  The shape below is a synthetic H2vDevPlan that only ws_create_for / default_chunk / laned_depth ever read.
extern "C" int h2v_workspace_create_multi(const h2v_plan *const *plans, uint32_t n_plans, uint64_t max_batch, uint32_t n_lanes, uint32_t chunk, h2v_workspace **out) {
    if (!plans || !out || n_plans == 0) return fail(H2V_E_ARG, "null argument");
    for (uint32_t k = 0; k < n_plans; k++) {
        if (!plans[k]) return fail(H2V_E_ARG, "null plan");
        ALIVE(plans[k]);
        if (plans[k]->device != plans[0]->device) return fail(H2V_E_ARG, "the plans of one workspace must live on one device");
    }
    H2vDevPlan u = plans[0]->d;
    uint32_t slots = 0, regs_global = 0, chunk_min = 0;
    for (uint32_t k = 0; k < n_plans; k++) {
        const H2vDevPlan &d = plans[k]->d;
        u.n_terms = d.n_terms > u.n_terms ? d.n_terms : u.n_terms;
        u.n_main_terms = d.n_main_terms > u.n_main_terms ? d.n_main_terms : u.n_main_terms;
        slots = (uint32_t)H2V_SLOTS(d) > slots ? (uint32_t)H2V_SLOTS(d) : slots;
        if (vm_lds_slots(d) == 0 && d.n_regs > regs_global) regs_global = d.n_regs;
        if (d.ivc) u.ivc = d.ivc;
        if (d.fix_tab) u.fix_tab = d.fix_tab;
        const uint32_t c = lane_chunk(d);
        chunk_min = chunk_min == 0 || c < chunk_min ? c : chunk_min;
    }
    u.n_ci = 0;
    u.n_points = slots - 2u * (u.ivc ? 1u : 0u);                  // H2V_SLOTS(u) == the largest plan's
    if (regs_global) { u.vm_lanes = 1; u.n_regs = regs_global; }   // vm_lds_slots(u) == 0: a global register file of that size
    else if (vm_lds_slots(u) == 0) return fail(H2V_E_ARG, "internal: union shape");
    (void)chunk_min;
    return create_lanes_for(u, plans[0]->device, max_batch, n_lanes, chunk, out);      // (chunk = 0: the library's lane size, whatever max_batch is)
}
extern "C" int h2v_workspace_create(const h2v_plan *p, uint64_t max_batch, h2v_workspace **out) {
    if (!p || !out) return fail(H2V_E_ARG, "null argument");
    // a workspace for batches of at least four chunks is laned: the call is pipelined inside the library (measured,
    // simple_mul: 8192 proofs in one launch per kernel 5.6 ms per 4096 against 5.9 in two chunks; 20480: 5.3 against 4.7)
    ALIVE(p);
    if (max_batch >= 4ull * default_chunk(p->d)) return h2v_workspace_create_lanes(p, max_batch, 0, 0, out);
    const int rc = ws_create_for(p->d, p->device, max_batch, false, out);
    if (rc == H2V_OK) reg_add(g_ws_live, *out);
    return rc;
}
static int option_check(uint32_t option, int32_t value) {
    auto bad = [](const char *m) { return fail(H2V_E_ARG, m); };
    switch (option) {
    case H2V_OPT_MSM_TERMS_PER_LANE: if (value < 0 || value > 4) return bad("terms per lane: 0 (auto) .. 4"); break;
    case H2V_OPT_PAIRING_ENGINE:
        if (value != 0 && value != 1 && value != 6 && value != 12 && value != 16 && value != 32 && value != 64)
            return bad("pairing engine: 0 (auto), 6, 12, 16, 32, 64 lanes per proof, or 1 (the one-lane cross-check kernel)");
        break;
    case H2V_OPT_STREAMS: if (value < -1 || value > 2) return bad("streams: -1 (auto), 0 (three), 1 (the caller's), 2 (the caller's + one for the decompression)"); break;
    case H2V_OPT_MSM_LANES_PER_TERM: if (value != 0 && value != 1 && value != 2 && value != 8) return bad("MSM lanes per term: 0 (auto), 1, 2, 8"); break;
    case H2V_OPT_MSM_BLOCK_SIZE: if (value < 0 || value > 512 || value % 64) return bad("MSM block size: 0 (auto), 64 .. 512 in steps of 64"); break;
    case H2V_OPT_MSM_FIXED_SPLIT: if (value < -1 || value > 4) return bad("fixed-base split: 0 (auto), 1 .. 4 bases per lane, -1 (never)"); break;
    case H2V_OPT_COMBINER_SCHEDULE: if (value < 0 || value > 2) return bad("combiner schedule: 0 (auto), 1 (narrow), 2 (wide)"); break;
    case H2V_OPT_COMBINER_PROOFS_PER_BLOCK: if (value < 0 || value > 64 || (value & (value - 1))) return bad("combiner proofs per block: 0 (auto) or a power of two <= 64"); break;
    case H2V_OPT_DECOMPRESS_FORM: if (value < 0 || value > 2) return bad("decompression: 0 (auto: one queue launch), 1 (two launches), 2 (one launch of paired blocks)"); break;
    case H2V_OPT_PIPES: if (value < 0 || value > h2v_workspace::MAXP) return bad("pipes: 0 / 1 (one pipeline) .. 4"); break;
    case H2V_OPT_RLC_GROUP_STAGE: if (value < -1 || value > 0) return bad("RLC group stage: 0 (auto), -1 (off)"); break;
    case H2V_OPT_RLC_WINDOW_BITS: if (value != 0 && (value < 3 || value > (int32_t)PIP_MAX_C)) return bad("RLC window bits: 0 (auto), 3 .. the bucket MSM's maximum"); break;
    case H2V_OPT_RLC_CHAIN: if (value != 0 && (value < 2 || value > 1024)) return bad("RLC entries per lane: 0 (auto), 2 .. 1024"); break;
    case H2V_OPT_RLC_ROUTE: if (value < -1 || value > 0) return bad("RLC routing: 0 (by the observed rate of failing groups), -1 (never)"); break;
    case H2V_OPT_COALESCE: if (value < -1 || value > 0) return bad("coalescing of small calls: 0 (auto: deferred joins, calls of at most half a chunk), -1 (never)"); break;
    default: return bad("unknown option");
    }
    return H2V_OK;
}
extern "C" int h2v_workspace_set_option(h2v_workspace *ws, uint32_t option, int32_t value) {
    if (!ws) return fail(H2V_E_ARG, "null argument");
    ALIVE(ws);
    if (int rc = option_check(option, value)) return rc;
    auto apply = [&](h2v_workspace *w) {
        if (option == H2V_OPT_STREAMS) w->one_stream_mode = value;
        else w->opt[option] = value;
    };
    apply(ws);
    for (uint32_t l = 0; l < (uint32_t)h2v_workspace::MAXL; l++) if (ws->lane[l] && option != H2V_OPT_STREAMS) apply(ws->lane[l]);
    return H2V_OK;
}
extern "C" int h2v_workspace_get_option(const h2v_workspace *ws, uint32_t option, int32_t *value) {
    if (!ws || !value || option == 0 || option >= H2V_OPT_COUNT) return fail(H2V_E_ARG, "bad argument");
    *value = option == H2V_OPT_STREAMS ? ws->one_stream_mode : ws->opt[option];
    return H2V_OK;
}
extern "C" int h2v_workspace_lanes(const h2v_workspace *ws, uint32_t *n_lanes, uint32_t *chunk) {
    if (!ws) return fail(H2V_E_ARG, "null argument");
    if (n_lanes) *n_lanes = ws->n_lanes ? ws->n_lanes : 1;
    if (chunk) *chunk = ws->n_lanes ? ws->chunk : (uint32_t)ws->cap;
    return H2V_OK;
}
static uint32_t laned_depth(const h2v_workspace *w, uint64_t n, bool rlc, int *stream_mode);
extern "C" int h2v_workspace_depth(const h2v_workspace *ws, uint64_t n, int rlc, uint32_t *batches_in_flight) {
    if (!ws || !batches_in_flight) return fail(H2V_E_ARG, "null argument");
    ALIVE(ws);
    *batches_in_flight = ws->n_lanes ? laned_depth(ws, n, rlc != 0, nullptr) : 1;
    return H2V_OK;
}
extern "C" int h2v_workspace_defer_joins(h2v_workspace *ws, int defer) {
    if (!ws) return fail(H2V_E_ARG, "null argument");
    ALIVE(ws);
    if (!ws->n_lanes) return fail(H2V_E_ARG, "only a laned workspace can defer its joins (h2v_workspace_create_lanes)");
    if (!defer) { HIPCHK(hipSetDevice(ws->device)); if (int rcf = co_flush(ws)) return rcf; }   // (an open group of coalesced calls runs now)
    ws->defer_joins = defer != 0;
    return H2V_OK;
}
// Deferred joins and the legacy NULL stream do not go together: the pool streams are created with the default flags
// (hipExtStreamCreateWithCUMask has no non-blocking form), so every operation on the NULL stream - PyTorch's default stream
// IS that stream - waits for all lanes and holds them back behind it: the pipelining would be lost without a word.  A
// deferring workspace therefore refuses stream = NULL (H2V_STREAM_CUMASK=0 streams are non-blocking: allowed).
static int null_stream_check(const h2v_workspace *ws, const void *stream) {
    static const bool blocking_pool = []() { const char *e = getenv("H2V_STREAM_CUMASK"); return !e || atoi(e) != 0; }();
    if (ws && ws->n_lanes && ws->defer_joins && stream == nullptr && blocking_pool)
        return fail(H2V_E_ARG, "a workspace with deferred joins needs a stream of its own: the legacy NULL stream (PyTorch's default stream) is ordered with "
                               "every library stream and would serialise the lanes - pass a non-default stream, or h2v_workspace_defer_joins(ws, 0)");
    return H2V_OK;
}
// every lane that has work enqueued since the last join: `st` waits for its last chunk
static int lanes_join(h2v_workspace *w, hipStream_t st, bool host_block = false) {
    if (int rcf = co_flush(w)) return rcf;          // (the open group of coalesced calls runs now)
    for (uint32_t l = 0; l < w->n_lanes; l++)
        if (w->lane_busy[l]) {
            if (host_block) HIPCHK(hipEventSynchronize(w->lane_ev[l]));
            else HIPCHK(hipStreamWaitEvent(st, w->lane_ev[l], 0));
            w->lane_busy[l] = false;
        }
    return H2V_OK;
}
extern "C" int h2v_workspace_join(h2v_workspace *ws, void *stream) {
    if (!ws) return fail(H2V_E_ARG, "null argument");
    ALIVE(ws);
    if (!ws->n_lanes) return H2V_OK;   // an ordinary workspace's calls are already ordered on the caller's stream
    HIPCHK(hipSetDevice(ws->device));
    // stream = NULL: the HOST waits (the legacy NULL stream is never touched: every pool stream is a blocking stream, i.e.
    // ordered with it, and a wait enqueued there would serialise all lanes)
    return lanes_join(ws, (hipStream_t)stream, stream == nullptr);
}
extern "C" void h2v_workspace_free(h2v_workspace *w) {
    if (!w) return;
    reg_del(g_ws_live, w);
    if (!w->dead) ws_release(w);
    delete w;
}
extern "C" int h2v_workspace_hint_in_flight(h2v_workspace *ws, uint32_t n_in_flight) {
    if (!ws || n_in_flight == 0) return fail(H2V_E_ARG, "bad argument");
    ALIVE(ws);
    ws->in_flight_hint = n_in_flight;
    ws->hint_given = true;
    return H2V_OK;
}

// A workspace is sized from the plan it was created for; another plan may use it iff every buffer is large enough for
// it (terms, point slots, a global register file if it needs one, recursion buffers, the fixed-base sum).
static int ws_fits(const h2v_workspace *w, const h2v_plan *p, uint64_t n, bool want_trace) {
    const H2vDevPlan &d = p->d;
    if (w->device != p->device) return fail(H2V_E_ARG, "workspace belongs to another device");
    if (w->cap < n) return fail(H2V_E_ARG, "workspace too small for this batch");
    if (w->n_lanes) {
        if (want_trace) return fail(H2V_E_ARG, "a laned workspace has no trace buffer");
        return ws_fits(w->lane[0], p, n < w->chunk ? n : w->chunk, false);
    }
    if (d.n_terms > w->sz_terms || H2V_SLOTS(d) > w->sz_slots) return fail(H2V_E_ARG, "workspace was created for a smaller plan (MSM terms / point slots)");
    if (vm_lds_slots(d) == 0 && d.n_regs > w->sz_regs) return fail(H2V_E_ARG, "workspace has no (or too small a) global register file for this plan");
    if (d.ivc && !w->sz_ivc) return fail(H2V_E_ARG, "workspace was created for a non-recursive plan");
    if (d.fix_tab && !w->sz_fix) return fail(H2V_E_ARG, "workspace lacks the fixed-base sum buffer of this plan");
    if (want_trace && d.n_trace > w->sz_trace) return fail(H2V_E_ARG, "workspace has no trace buffer for this plan");
    return H2V_OK;
}

// Library-owned streams.  The runtime multiplexes ordinary streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues per
// process, and kernels of two streams that share a queue run one after the other - with five batches in flight on three
// streams each that is the difference between 5.7 and 4.3 ms per 4096-proof batch (round 2 needed GPU_MAX_HW_QUEUES=16 in
// the caller's environment for it).  A stream created with an explicit CU mask gets a hardware queue of its own, whatever
// the process-wide setting is (measured, default environment: 5 lanes 5.79 -> 4.34 ms per batch).  Every queue also gets
// its own scratch arena, sized for the kernel with the largest private segment (a process that had created 47 of them
// died with HSA_STATUS_ERROR_OUT_OF_RESOURCES at the next launch), so the library owns a fixed POOL of such streams per
// device - H2V_QUEUE_POOL, default 16 (24 measured slower: 4.97 against 4.30 ms per batch) - and hands them out round robin: workspaces share them (a stream is an ordering
// domain, sharing one only adds order), they live until h2v_shutdown, and the mask names every CU.  Pool streams have the
// default flags, i.e. they are ordered with the legacy NULL stream: callers that defer joins should not submit on the
// NULL stream.  H2V_STREAM_CUMASK=0: plain non-blocking streams of the runtime's own pool instead.
static std::vector<hipStream_t> g_pool[16];
static size_t g_pool_next[16] = {};
extern "C" int h2v_shutdown(int device);
static const size_t g_pool_cap = []() { const char *e = getenv("H2V_QUEUE_POOL"); const int v = e ? atoi(e) : 16; return (size_t)(v < 1 ? 1 : v > 32 ? 32 : v); }();
static hipError_t make_stream(hipStream_t *s) {
    static const int cumask = []() { const char *e = getenv("H2V_STREAM_CUMASK"); return e ? atoi(e) : 1; }();
    const size_t pool_cap = g_pool_cap;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_reg_mu);
    if (g_shut[dev]) return hipErrorContextIsDestroyed;
    if (g_pool[dev].size() < pool_cap) {
        hipStream_t ns = nullptr;
        if (cumask) {
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            std::vector<uint32_t> mask((cus + 31) / 32, 0xffffffffu);
            if (cus % 32) mask.back() = (1u << (cus % 32)) - 1;
            e = hipExtStreamCreateWithCUMask(&ns, (uint32_t)mask.size(), mask.data());
        } else {
            e = hipStreamCreateWithFlags(&ns, hipStreamNonBlocking);
        }
        if (e != hipSuccess) return e;
        // registered now, i.e. after the HIP runtime has initialised itself: exit handlers run in reverse order of
        // registration, so the pool is destroyed while the runtime (and a profiler's tool) are still whole
        static const int registered = atexit([]() { (void)h2v_shutdown(-1); });
        (void)registered;
        g_pool[dev].push_back(ns);
        *s = ns;
        return hipSuccess;
    }
    *s = g_pool[dev][g_pool_next[dev]++ % g_pool[dev].size()];
    return hipSuccess;
}
// The two streams of lane l of ANY laned workspace: a fixed place in the pool, so that the lanes of one workspace never share a
// stream among themselves whatever else was created in between (handing streams out in creation order, two workspaces whose
// lanes were created alternately ended up with a lane whose side stream WAS its main stream, and with two lanes of one
// workspace on one pair: 3.7 / 4.1 ms per call where 3.3 / 3.4 are normal - bench.py's mixed workload, round 4).  Lanes 0..7
// (the per-proof mode's eight two-stream lanes): main 2 l, side 2 l + 1; lanes 8..15 (only the sixteen one-stream lanes of
// small chunks and of the RLC mode reach them): the odd streams as main.  Lanes of different workspaces with the same index
// share their pair - with sixteen hardware queues and two plans in flight something has to.  (main l, side l + 8 instead: 1.7 % slower,
// 3.30-3.32 against 3.24-3.26 ms per simple_mul x 4096 batch; pairs spread over the pool in steps of four: the same.)
static int lane_streams(uint32_t l, hipStream_t *main_st, hipStream_t *side_st) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    for (;;) {                                     // fill the pool (make_stream creates while it is below its capacity)
        size_t have, cap;
        { std::lock_guard<std::mutex> lock(g_reg_mu); have = g_pool[dev].size(); cap = g_pool_cap; }
        if (have >= cap && have > 0) break;
        hipStream_t q;
        if (make_stream(&q) != hipSuccess) return fail(H2V_E_DEVICE, "lane stream creation failed");
        { std::lock_guard<std::mutex> lock(g_reg_mu); if (g_pool[dev].size() == have) break; }   // (H2V_QUEUE_POOL smaller than asked: the pool is full)
    }
    std::lock_guard<std::mutex> lock(g_reg_mu);
    const size_t n = g_pool[dev].size();
    if (n == 0) return fail(H2V_E_DEVICE, "no pool stream");
    const uint32_t m = l < 8 ? 2 * l : 2 * (l - 8) + 1;
    *main_st = g_pool[dev][m % n];
    *side_st = g_pool[dev][(m ^ 1u) % n];
    return H2V_OK;
}
// Releases everything the library owns on `device` (-1: on every device): waits for the pool streams, releases every live
// workspace (they become empty shells: dead) and every plan's device memory, destroys the pool streams.  Idempotent.
extern "C" int h2v_shutdown(int device) {
    if (device < -1 || device >= 16) return fail(H2V_E_ARG, "device index out of range");
    std::vector<h2v_workspace *> wss;
    std::vector<h2v_plan *> plans;
    std::vector<hipStream_t> streams[16];
    {
        std::lock_guard<std::mutex> lock(g_reg_mu);
        for (h2v_workspace *w : g_ws_live) if (!w->dead && (device < 0 || w->device == device)) wss.push_back(w);
        for (h2v_plan *p : g_plan_live) if (!p->dead && (device < 0 || p->device == device)) plans.push_back(p);
        for (int d = 0; d < 16; d++)
            if (device < 0 || d == device) { g_shut[d] = true; streams[d].swap(g_pool[d]); }
    }
    int rc = H2V_OK;
    for (int d = 0; d < 16; d++) {
        if (streams[d].empty()) continue;
        if (hipSetDevice(d) != hipSuccess) { rc = fail(H2V_E_DEVICE, "hipSetDevice failed during shutdown"); continue; }
        for (hipStream_t q : streams[d]) (void)hipStreamSynchronize(q);
    }
    for (h2v_workspace *w : wss) {
        const int dev = w->device;
        ws_release(w);
        *w = h2v_workspace();
        w->device = dev; w->dead = true;
    }
    for (h2v_plan *p : plans) { plan_release(p); p->dead = true; }
    for (int d = 0; d < 16; d++) {
        if (streams[d].empty()) continue;
        if (hipSetDevice(d) != hipSuccess) continue;
        for (hipStream_t q : streams[d])
            if (hipStreamDestroy(q) != hipSuccess) rc = fail(H2V_E_DEVICE, "hipStreamDestroy(pool stream) failed");
    }
    return rc;
}
// the main / side / third stream of pipeline chunk k, created when first asked for
static int ws_streams(h2v_workspace *w, int k, bool need_main, bool need_side, bool need_sub) {
    if (need_main && !w->pmain[k]) HIPCHK(make_stream(&w->pmain[k]));
    if (need_side && !w->pside[k]) HIPCHK(make_stream(&w->pside[k]));
    if (need_sub && !w->psub[k]) HIPCHK(make_stream(&w->psub[k]));
    return H2V_OK;
}

// launch-shape options of the workspace a call runs on (h2v_workspace_set_option): the launchers below run on the calling
// host thread, inside run_pipeline / run_rlc, which set this for their duration
struct LaunchOptions { int32_t v[H2V_OPT_COUNT] = {}; uint32_t in_flight = 1; };
static thread_local LaunchOptions g_opts;
static thread_local LaunchOptions g_probe_opts;     // h2v_probe_set_option: the shape options the probes of this thread launch with
static void opts_from(const h2v_workspace *w) { memcpy(g_opts.v, w->opt, sizeof g_opts.v); g_opts.in_flight = w->in_flight_hint; }
struct ProbeOpts { ProbeOpts() { g_opts = g_probe_opts; } ~ProbeOpts() { g_opts = LaunchOptions(); } };
extern "C" int h2v_probe_set_option(uint32_t option, int32_t value) {
    if (option == H2V_OPT_STREAMS) return fail(H2V_E_ARG, "the probes run on the NULL stream");
    if (int rc = option_check(option, value)) return rc;
    g_probe_opts.v[option] = value;
    return H2V_OK;
}
// Transcript + combiner launch.  A block is one wave; its 64 lanes are P proofs x L lanes per proof (the plan's
// bundles have L records), the Fr register file of the P proofs lives in LDS (160 KB per CU).  A single-lane plan whose
// register file does not fit 64 proofs runs P = 32 / 16 / 8 proofs per block with the other lanes idle, or - below 8 -
// with the register file in the workspace's global buffer.
static uint32_t vm_lds_slots(const H2vDevPlan &d) {
    if (d.vm_lanes > 1) return 64 / d.vm_lanes;   // validated at load: fits
    uint32_t P = 64;
    while (P >= 8 && (size_t)d.n_regs * 32 * P > H2V_VM_LDS_BYTES) P >>= 1;
    return P >= 8 ? P : 0;
}
static int launch_vm(const H2vDevPlan &d0, uint32_t n, uint32_t stride, const uint8_t *proofs, const uint64_t *off, const uint8_t *inst,
                     const uint8_t *ci, uint32_t *regs, uint32_t *scalars, uint32_t *status, uint32_t *trace, hipStream_t st) {
    // The wide schedule (more lanes per proof: shorter chain, more waves) when the launch would leave most of the chip
    // idle anyway - at most a quarter of the SIMDs get a wave - and nobody asked for the trace (its register numbers are
    // the narrow schedule's).  H2V_OPT_COMBINER_SCHEDULE forces the choice.
    const int env_wide = g_opts.v[H2V_OPT_COMBINER_SCHEDULE] - 1;   // -1 auto, 0 narrow, 1 wide
    H2vDevPlan d = d0;
    const bool wide_ok = d.wide_lanes && !trace;
    // (eight or more batches in flight: the chip is full anyway and the wide schedule's extra instructions count - only launches
    //  of at most 32 waves take it.  ms per batch, wide -> narrow schedule: sha256 shape x 128 0.66 -> 0.69, x 256 0.88 -> 0.84,
    //  x 512 1.22 -> 1.18, x 1024 2.16 -> 2.00; secp256k1 shape x 64 0.54 -> 0.56, x 512 1.32 -> 1.16)
    const uint64_t wide_waves = ((uint64_t)n * d.wide_lanes + 63) / 64;
    const bool wide = wide_ok && (env_wide >= 0 ? env_wide != 0 : wide_waves <= (g_opts.in_flight >= 8 ? 32u : 256u));
    if (wide) { d.vm_lanes = d.wide_lanes; d.n_regs = d.wide_n_regs; d.n_instr = d.wide_n_instr; d.instr = d.wide_instr; }
    uint32_t P = vm_lds_slots(d);
    if (P == 0) {
        hipLaunchKernelGGL(k_transcript_combiner, dim3((n + 63) / 64), dim3(64), 0, st, d, n, stride, proofs, off, inst, ci, regs, scalars, status, trace);
        return H2V_OK;
    }
    // Fewer proofs per block than the wave could serve (the spare lanes shadow): the register file of P proofs is what
    // decides how many blocks a CU holds (simple_mul, P = 32: 79 + 8 KB = ONE block per CU = one wave on one of its four
    // SIMDs), and a wave's chain is as long with 16 proofs as with 32.  H2V_OPT_COMBINER_PROOFS_PER_BLOCK forces P.
    const int env_p = g_opts.v[H2V_OPT_COMBINER_PROOFS_PER_BLOCK];
    if (env_p >= 1 && (env_p & (env_p - 1)) == 0 && (uint32_t)env_p <= P) P = (uint32_t)env_p;
    const size_t lds = (size_t)d.n_regs * 32 * P;
    HIPCHK(hipFuncSetAttribute((const void *)k_transcript_combiner_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_transcript_combiner_lds, dim3((n + P - 1) / P), dim3(64), lds, st, d, n, P, proofs, off, inst, ci, scalars, status, trace);
    return H2V_OK;
}

// MSM launch geometry: 2 lanes per (proof, term); LDS 172 B per thread (42 limbs + the infinity flag).
// MSM launch shape from a cost model fitted to MI355X measurements:
//  * lanes per term: 2 = one lane per GLV half, chain of ~1250 multiplications; 1 = both halves on one accumulator,
//    chain ~1600 but 36 % less work per proof and half the waves;
//  * a wave alone on its SIMD runs ~1.7x faster than two sharing one (2048 proofs: 1.50 ms, 4096: 2.6 ms), and the
//    dispatcher only spreads one wave per SIMD for 64- and 256-thread blocks (128 / 192 / 320 / 384 / 448 / 512
//    put two waves of a block on the same SIMD: 2.5 ms where 64 / 256 take 1.5 ms at 1024 waves).
// cost = chain x (1 if every wave can sit alone, else 1.7 x whole rounds of two waves per SIMD: the waves of a
// launch all take the same time, so a partly filled round costs a full one).
static double msm_n_simd() {
    static const double v = []() {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return 4.0 * cus;
    }();
    return v;
}
struct MsmShape { uint32_t lpt, bs; double cost, waves; };
// lanes_per_proof lanes of chain length `chain` per proof; other_waves: waves of a launch running beside this one
static void msm_try_shape(MsmShape &best, uint32_t lpt, uint32_t lpp, double chain, uint32_t n, double other_waves, uint32_t force_bs) {
    for (uint32_t cand = 64; cand <= 512; cand += 64) {
        if (cand < lpp || (force_bs && cand != force_bs)) continue;
        const uint32_t pb = cand / lpp;
        const double waves = (double)((n + pb - 1) / pb) * (cand / 64), rho = (waves + other_waves) / msm_n_simd();
        const bool spreads = cand == 64 || cand == 256;
        const double rounds = rho > 2.0 ? (double)(uint64_t)((rho + 1.999) / 2.0) : 1.0;
        double cost = chain * ((spreads && rho <= 1.0) ? 1.0 : 1.7 * rounds);
        // ties: 256-thread blocks first (four waves, one per SIMD of a CU whatever the dispatcher's state: after a
        // launch of 128-thread blocks, 1024 one-wave blocks of this kernel measured 2.47 ms instead of 1.86, 256-thread
        // blocks 1.87), then one-wave blocks, then fewer idle lanes
        cost *= 1.0 + (cand == 256 ? 0.0 : cand == 64 ? 0.004 : 0.01) + 0.005 * (double)(cand - pb * lpp) / cand;
        if (cost < best.cost) { best.cost = cost; best.lpt = lpt; best.bs = cand; best.waves = waves; }
    }
}
static MsmShape msm_ladder_shape(uint32_t n_terms, uint32_t n, double other_waves, bool quad_ok = false) {
    const int env_lpt = g_opts.v[H2V_OPT_MSM_LANES_PER_TERM];
    const uint32_t env_bs = (uint32_t)g_opts.v[H2V_OPT_MSM_BLOCK_SIZE];
    MsmShape best = {2, 512, 1e300, 0};
    for (uint32_t cl = 2; cl >= 1; cl--) {
        if (env_lpt && (uint32_t)env_lpt != cl) continue;
        msm_try_shape(best, cl, cl * n_terms, cl == 2 ? 1250.0 : 1600.0, n, other_waves, env_bs);
    }
    // a quad per GLV half (h2v_kernels.hip: msm_body, LPT = 8)
    // Only on request (H2V_OPT_MSM_LANES_PER_TERM = 8): alone it shortens a T = 16 launch of 64-512 proofs from 1.35 to 1.17-1.27 ms, but it issues
    // four times the instructions, and with four steps in flight - how small batches are run for throughput - the step got
    // slower at 64 and 512 proofs (1.30 -> 1.40, 1.66 -> 1.90 ms) and faster only at 256 (1.57 -> 1.47).
    if (quad_ok && 8 * n_terms <= 512 && env_lpt == 8) {
        MsmShape q = {8, 512, 1e300, 0};
        msm_try_shape(q, 8, 8 * n_terms, 1080.0, n, other_waves, env_bs);
        if (q.cost < 1e300) best = q;
    }
    if (best.cost == 1e300) {   // forced shape that does not fit: fall back to the widest block
        const uint32_t pb = 512 / (2 * n_terms) ? 512 / (2 * n_terms) : 1;
        best.lpt = 2; best.bs = 512; best.waves = (double)((n + pb - 1) / pb) * 8;
    }
    return best;
}
static uint32_t launch_msm_ladders(const H2vDevPlan &d, const H2vMsmArgs &ma, uint32_t n, const MsmShape &sh, const uint32_t *scalars,
                                   const uint32_t *pts, uint32_t *tabws, hipStream_t st) {
    const uint32_t lpp = sh.lpt * ma.n_terms;
    const uint32_t per_block = sh.bs / lpp;
    const uint32_t blocks = (n + per_block - 1) / per_block;
    if (ma.skip) {   // fall-back of the RLC batch mode: a grid of at most one block per CU walks the logical blocks
        const uint32_t cg = (uint32_t)(msm_n_simd() / 4.0), blocks_c = blocks < cg ? blocks : cg;
        if (sh.lpt == 1) hipLaunchKernelGGL(k_g1_msm_merged_cond, dim3(blocks_c), dim3(sh.bs), (size_t)sh.bs * 172, st, d, ma, n, per_block, scalars, pts, tabws);
        else hipLaunchKernelGGL(k_g1_msm_cond, dim3(blocks_c), dim3(sh.bs), (size_t)sh.bs * 172, st, d, ma, n, per_block, scalars, pts, tabws);
    } else if (sh.lpt == 8) hipLaunchKernelGGL(k_g1_msm_quad, dim3(blocks), dim3(sh.bs), (size_t)sh.bs * 172, st, d, ma, n, per_block, scalars, pts, tabws);
    else if (sh.lpt == 1) hipLaunchKernelGGL(k_g1_msm_merged, dim3(blocks), dim3(sh.bs), (size_t)sh.bs * 172, st, d, ma, n, per_block, scalars, pts, tabws);
    else hipLaunchKernelGGL(k_g1_msm, dim3(blocks), dim3(sh.bs), (size_t)sh.bs * 172, st, d, ma, n, per_block, scalars, pts, tabws);
    return sh.lpt;
}
// H2V_OPT_MSM_TERMS_PER_LANE = 2 .. 4: k_g1_msm_multi (several terms per lane share the doublings: less work, fewer and longer waves;
// for callers that keep several batches in flight).  Single-group launches with prebuilt tables only.
// Without the option the caller's hint decides (h2v_workspace_hint_in_flight): a caller that keeps >= 4 batches in flight is
// bound by the instructions issued, not by chain length, and two terms per lane issue 26 % fewer multiply-adds per proof
// (measured, simple_mul x 4096: 5 in flight 5.12 -> 4.66 ms per step; with 3 in flight 5.04 -> 5.01).
// Terms per lane of the ladder kernel for callers that keep the chip full: more terms per lane share more doublings and make
// fewer, longer waves - as many as still leave the launch a quarter of a wave per SIMD (two at least).  ms per batch, eight
// batches in flight, terms per lane 2 / 3 / 4: simple_mul x 4096 (10 terms: 320 / 256 / 192 waves) 3.73 / 3.64 / 3.84;
// lookup_table x 2048 (25 terms: 416 / 288 / 224) 3.42 / 3.36 / -; x 4096 6.82 / 6.71 / 6.59; atms x 2048 (20 terms: 320 / 224 /
// 160) 3.55 / 3.72 / 4.05; sha256 shape in chunks of 1024 (25 terms: 208 / 144) 2.26 / 2.35.
static int msm_terms_per_lane(uint32_t in_flight_hint, uint32_t n = 0, uint32_t n_terms = 0) {
    if (g_opts.v[H2V_OPT_MSM_TERMS_PER_LANE] >= 1) return g_opts.v[H2V_OPT_MSM_TERMS_PER_LANE];
    if (in_flight_hint < 4) return 1;
    for (int t = 4; t > 2; t--)
        if ((double)n * ((n_terms + t - 1) / t) / 64.0 >= msm_n_simd() / 4.0) return t;
    return 2;
}
static uint32_t launch_msm_range(const H2vDevPlan &d, const H2vMsmArgs &ma, uint32_t n, const uint32_t *scalars, const uint32_t *pts,
                                 uint32_t *tabws, hipStream_t st, uint32_t in_flight_hint = 1) {
    const int tpl = msm_terms_per_lane(in_flight_hint, n, ma.n_terms);
    // (only launches of at least a quarter of a wave per SIMD at one lane per term: below that the launch is a chain of lone
    //  waves whatever else is in flight, and the two-lanes-per-term ladder is the shortest chain - sha256 shape x 128 with six
    //  shares in flight: MSM 2.5 ms -> 1.4 ms alone)
    // (eight or more batches in flight: from an eighth - sha256 / secp256k1 shape x 256, sixteen shares in flight: 1.05 / 1.08 ->
    //  0.89 / 0.91 ms per share with two terms per lane; x 128 and x 64 stay with the shortest chain: 0.66 -> 0.74, 0.54 -> 0.64)
    const bool fills = (double)n * ma.n_terms / 64.0 >= msm_n_simd() / (in_flight_hint >= 8 ? 8.0 : 4.0);
    const bool tpl_forced = g_opts.v[H2V_OPT_MSM_TERMS_PER_LANE] > 0;
    if (tpl > 1 && (fills || tpl_forced) && ma.pt_tab && !ma.skip && ma.grp_end[0] == ma.n_terms && ma.n_terms >= (uint32_t)tpl && ma.n_terms <= 256u * tpl) {
        // lanes per proof as for `tpl` whole terms per lane, then the proof's 2 T GLV halves dealt out evenly over them: ten terms
        // on four lanes are 5 + 5 + 5 + 5 halves, not 6 + 6 + 6 + 2 (a forced tpl keeps its 2 tpl halves per lane)
        const uint32_t lpp0 = (ma.n_terms + tpl - 1) / tpl, bs = 256;
        const uint32_t hpl = tpl_forced ? 2u * (uint32_t)tpl : (2 * ma.n_terms + lpp0 - 1) / lpp0;
        const uint32_t lpp = (2 * ma.n_terms + hpl - 1) / hpl;
        const uint32_t per_block = bs / lpp, blocks = (n + per_block - 1) / per_block;
        hipLaunchKernelGGL(k_g1_msm_multi, dim3(blocks), dim3(bs), (size_t)bs * 172, st, d, ma, n, per_block, hpl, scalars, pts, tabws);
        return 16 + (hpl + 1) / 2;   // reported as msm_lanes_per_term: 18 / 19 / 20 = up to two / three / four terms' halves per lane
    }
    return launch_msm_ladders(d, ma, n, msm_ladder_shape(ma.n_terms, n, 0.0, ma.pt_tab != nullptr && !ma.skip), scalars, pts, tabws, st);
}
// Fixed-base split of the plan's own MSM (non-recursive plans, tables present): the per-proof terms [0, n_var) as ladders
// and, beside them on another stream, the VK-base terms as one lane per term that walks the all-window table of its base
// (65 mixed additions, no doubling: 0.88 ms alone).  Measured (2048 proofs): T = 50 with 30 VK bases 3.37 -> 2.52 ms,
// T = 34 with 9 VK bases 3.42 -> 2.68 ms; but where the single launch already has every SIMD to itself the split is
// slower (simple_mul x 4096: 1.87 -> 2.50 ms, sha256 shape x 1024: 1.99 -> 2.56 ms) - waves of two concurrent launches
// pair up on SIMDs even when there would be room for all of them alone.  So the rule is: split (one base per lane) only
// when the single launch cannot have one wave per SIMD and most terms are VK bases.  H2V_OPT_MSM_FIXED_SPLIT = k forces a split with k bases per lane, -1 forbids it.
struct MsmSplit { bool on; MsmShape var, fix; uint32_t k; };
static MsmSplit msm_split_shape(const H2vDevPlan &d, uint32_t n, const MsmShape &single, uint32_t in_flight_hint = 1) {
    const int opt_fix = g_opts.v[H2V_OPT_MSM_FIXED_SPLIT];                  // 0 auto, 1 .. 4 bases per lane, -1 never
    const int env_fix = opt_fix == 0 ? -1 : opt_fix < 0 ? 0 : opt_fix;       // (-1 auto, 0 never, k forced: the form the rule below is written in)
    const uint32_t env_bs = (uint32_t)g_opts.v[H2V_OPT_MSM_BLOCK_SIZE];
    MsmSplit out = {false, {}, {}, 0};
    if (!d.fix_tab || !d.n_fix || !d.n_var || d.ivc || env_fix == 0) return out;
    // (and only when the VK bases are the majority of the terms: with 9 of 34 the MSM gained 0.8 ms and the pairing kernel
    // that followed the three launches lost as much of its own placement; with 6 of 16 at 8192 proofs the split was slower)
    // A caller that keeps the chip full (hint >= 4: the lanes) is bound by the instructions issued: the VK-base terms then
    // ALWAYS go through the all-window tables, two bases per lane (65 mixed additions each and no doubling, where a ladder
    // lane shares 128 doublings between two terms) - measured with six batches in flight, ms per batch: simple_mul x 4096
    // 4.39 -> 4.30, sha256 shape x 1024 2.67 -> 2.56, atms x 2048 4.43 -> 4.21, lookup_table x 2048 and secp256k1 x 512
    // unchanged (+-1 %); launches below a quarter of a wave per SIMD keep the single ladder launch (sha256 x 128: 1.17 -> 1.26).
    const bool in_flight = in_flight_hint >= 4 && d.n_fix >= 2 && (double)n * d.n_main_terms / 64.0 >= msm_n_simd() / 4.0;
    if (env_fix < 0 && !in_flight && (single.waves <= msm_n_simd() || d.n_fix < d.n_var)) return out;
    const uint32_t k = env_fix > 0 ? (uint32_t)(env_fix > 4 ? 4 : env_fix) : in_flight ? (d.n_fix >= 16 ? 4u : 2u) : 1u;   // (bases per lane: +-1 % either way)
    const uint32_t lanes = (d.n_fix + k - 1) / k;
    MsmShape fx = {1, 512, 1e300, 0};
    msm_try_shape(fx, 1, lanes, k * 800.0, n, 0.0, env_bs);
    if (fx.cost == 1e300) return out;
    out.on = true;
    out.k = k;
    out.fix = fx;
    out.var = msm_ladder_shape(d.n_var, n, fx.waves);
    return out;
}
// the proof's own MSM: terms [0, n_main_terms) of the plan's table, scalars from the combiner, points from decompression.
// A recursive plan sums acc_left and acc_right + fixed bases in the same launch (three groups, three outputs).
static uint32_t launch_msm(const H2vDevPlan &d, uint32_t n, const uint32_t *scalars, const uint32_t *pts, const uint32_t *pt_tab,
                       uint32_t *er, uint32_t *accl, uint32_t *accr, hipStream_t st, uint32_t in_flight_hint = 1) {
    H2vMsmArgs ma = {d.terms, 0, d.n_main_terms, d.n_terms, 0, H2V_SLOTS(d), {d.n_main_terms, d.n_main_terms, d.n_main_terms}, {er, nullptr, nullptr},
                     pt_tab, d.vk_tab, nullptr, 0, 0};
    if (d.ivc) {
        ma.n_terms = d.n_terms;
        ma.grp_end[0] = d.n_main_terms; ma.grp_end[1] = d.n_main_terms + 1; ma.grp_end[2] = d.n_terms;
        ma.out[1] = accl; ma.out[2] = accr;
    }
    return launch_msm_range(d, ma, n, scalars, pts, nullptr, st, in_flight_hint);
}
// Recursion (IVC) fold between the MSM and the pairing (emitters/aiken.rs:696-757): the batching challenge from
// (el, er, acc_left, acc_right_final), then el' = el + c acc_left and er' = er + c acc_right_final in one two-group
// launch over the fold's own point / scalar buffers.  Pointers are the chunk's.
struct IvcBufs { uint32_t *accl, *accr, *fold_pts, *fold_scal, *el2, *er2; };
static void launch_ivc_fold(const H2vDevPlan &d, uint32_t n, const uint32_t *pts, const uint32_t *er,
                            const IvcBufs &b, uint32_t *tabws, hipStream_t st) {
    hipLaunchKernelGGL(k_ivc_challenge, dim3((n + 63) / 64), dim3(64), 0, st, d, n, pts, er, b.accl, b.accr, b.fold_pts, b.fold_scal);
    const H2vMsmArgs fold = {d.fold_terms, 0, 4, 4, 0, 4, {2, 4, 4}, {b.el2, b.er2, nullptr}, nullptr, nullptr, nullptr, 0, 0};
    launch_msm_range(d, fold, n, b.fold_scal, b.fold_pts, tabws, st);
}
// Pairing kernel selection: the cooperative 16-lanes-per-proof kernel is the product path; the one-lane-per-proof
// kernel stays as a cross-check (H2V_OPT_PAIRING_ENGINE = 1, or impl = 0 in the probe).
static uint32_t launch_pairing_impl(int impl, const H2vDevPlan &d, uint32_t n, const uint32_t *pts, const uint8_t *valid, const uint8_t *valid_sub, const uint32_t *er,
                                const uint32_t *el_jac, uint32_t *status, uint8_t *accept, uint32_t *dbg, hipStream_t st, const uint32_t *skip = nullptr,
                                bool prefer_narrow = false, double wide_up_to = -1.0, bool prefer_six = false, bool prefer_twelve = false) {
    if (wide_up_to < 0) wide_up_to = msm_n_simd();
    // The WIDE engine (one proof per wave, four lanes per coefficient: 3 / 2 / 1 terms per lane and call instead of 6 / 4 / 2)
    // when even one wave per proof leaves SIMDs free: n <= #SIMDs.  Above that the two-proofs-per-wave kernel does less
    // total work.  H2V_OPT_PAIRING_ENGINE forces an engine; the conditional (RLC fall-back) launch is never wide.
    // impl 2 / 3 / 5 / 6 (probe): the narrow / the wide / the six-lane / the twelve-lane kernel whatever n
    const int forced = g_opts.v[H2V_OPT_PAIRING_ENGINE];
    if (impl == 1 && forced) impl = forced == 1 ? 0 : forced == 16 ? 2 : forced == 64 ? 3 : forced == 6 ? 5 : forced == 12 ? 6 : 4;   // 4: the two-proofs-per-wave engine
    if (impl == 5 || (impl == 1 && !skip && prefer_six)) {
        hipLaunchKernelGGL(k_pairing_six, dim3((n + SIX_GROUPS - 1) / SIX_GROUPS), dim3(64), SIX_LDS_BYTES, st, d, n, pts, valid, valid_sub, er, el_jac, status, accept, dbg);
        return 6u;
    }
    const bool wide = !skip && impl != 2 && impl != 4 && impl != 6 && (impl == 3 || (double)n <= wide_up_to);
    const bool narrow = !skip && !wide && impl != 4 && (impl == 2 || prefer_narrow);
    // the narrow engine packed five proofs to a wave (impl 6, option 12): wherever the narrow one would run
    if (impl == 6 || (narrow && impl != 2 && prefer_twelve)) {
        hipLaunchKernelGGL(k_pairing_coop_twelve, dim3((n + COOP_GROUPS_TWELVE - 1) / COOP_GROUPS_TWELVE), dim3(64), COOP_LDS_BYTES(COOP_GROUPS_TWELVE), st, d, n, pts, valid, valid_sub, er, el_jac, status, accept, dbg);
        return 12u;
    }
    if (impl == 0) { hipLaunchKernelGGL(k_pairing_check, dim3((n + 63) / 64), dim3(64), 0, st, d, n, pts, valid, valid_sub, er, el_jac, status, accept, dbg); return 1; }
    else if (wide) hipLaunchKernelGGL(k_pairing_coop_wide, dim3(n), dim3(64), COOP_LDS_BYTES(1), st, d, n, pts, valid, valid_sub, er, el_jac, status, accept, dbg);
    else if (narrow) hipLaunchKernelGGL(k_pairing_coop_narrow, dim3((n + 3) / 4), dim3(64), COOP_LDS_BYTES(COOP_GROUPS_NARROW), st, d, n, pts, valid, valid_sub, er, el_jac, status, accept, dbg);
    else hipLaunchKernelGGL(k_pairing_coop, dim3((n + 1) / 2), dim3(64), COOP_LDS_BYTES(COOP_GROUPS_PER_WAVE), st, d, n, pts, valid, valid_sub, er, el_jac, status, accept, dbg, skip);
    return wide ? 64u : narrow ? 16u : 32u;
}
// The NARROW engine (four proofs per wave, one lane per coefficient: 21 % fewer instructions per pairing, twice the chain).
// Measured (simple_mul, pairing kernel alone / step with five batches in flight):
//   n = 1536: normal 1.94 ms / 2.32 ms, narrow 2.60 / 2.40      n = 2048: 1.96 / 2.71, 2.61 / 2.64
//   n = 3072: 2.99 / 3.71, 2.63 / 3.46     n = 4096: 3.00 / 4.81, 2.66 / 4.50     n = 8192: 5.65 / 9.59, 6.59 / 8.91
// (a narrow block needs 22.5 KB of LDS: seven per CU, so its second wave per SIMD does not fit everywhere).  Rule: a caller
// that keeps the chip full (hint >= 4) takes it from 2 * #SIMDs proofs up; one step at a time takes it exactly where the
// normal kernel needs a second wave per SIMD and the narrow one does not.  H2V_OPT_PAIRING_ENGINE forces the choice.
static uint32_t launch_pairing(const H2vDevPlan &d, uint32_t n, const uint32_t *pts, const uint8_t *valid, const uint8_t *valid_sub, const uint32_t *er,
                           const uint32_t *el_jac, uint32_t *status, uint8_t *accept, uint32_t *dbg, hipStream_t st, uint32_t in_flight_hint = 1) {
    const int impl = 1;
    const double S = msm_n_simd();
    // Eight or more batches in flight (the library's lanes): the chip is full whatever one launch brings, so the engine with the
    // fewest instructions that still has waves to spread wins much earlier - wide (2 x the normal engine's instructions) only up
    // to #SIMDs / 16 proofs, narrow from 3/8 #SIMDs.  ms per batch, wide / normal / narrow: secp256k1 shape x 64 0.55 / 0.58 / 0.63;
    // sha256 shape x 128 0.67 / 0.66 / 0.72; x 256 1.15 / 1.08 / 1.06; x 512 1.61 / 1.43 / 1.47; secp256k1 x 512 1.67 / 1.57 / 1.55;
    // simple_mul x 512 1.02 / 0.88 / 0.84; x 1024 - / 1.47 / 1.39; sha256 x 1024 - / 2.19 / 2.16.
    // ... and where the narrow engine would run, its twelve-lane packing (five proofs per wave instead of four with four idle lanes
    // each; the line products in a pass of their own): ms per batch, narrow -> twelve: sha256 shape x 1024 1.99 -> 1.96, secp256k1 x 512
    // 1.21 -> 1.17, lookup_table x 2048 3.26 -> 3.16, atms x 2048 3.40 -> 3.34, simple_mul x 1024 1.24 -> 1.17.
    const bool many = in_flight_hint >= 8;
    // (round 4, from the tuner's measurements: at #SIMDs / 4 proofs - sha256 / secp256k1 shape x 256 - the normal engine beats the
    //  twelve-lane one by 8-12 % in every run, 0.77-0.80 against 0.85-0.90 ms per batch; at x 512 neither wins by 3 %: the border is 3/8)
    const bool prefer_narrow = many ? (double)n >= 3.0 * S / 8.0 : in_flight_hint >= 4 ? (double)n >= 2.0 * S : ((double)n > 2.0 * S && (double)n <= 4.0 * S);
    // (the wide engine issues twice the instructions of the normal one: a caller that keeps the chip full takes it only up to
    // #SIMDs / 2 proofs - sha256 shape x 1024, four in flight: wide 3.18, normal 2.98, narrow 3.12 ms per step; secp256k1 x 512:
    // wide 2.37, normal 2.47, narrow 2.64)
    // The SIX-lane engine (ten proofs per wave, h2v_pairing_six.hpp: a quarter fewer instructions per pairing than the narrow
    // one, a chain almost twice as long) wherever a caller that keeps the chip full would take the narrow one
    // from 4 * #SIMDs proofs up (410 waves; with eight or more batches in flight from 3 * #SIMDs: simple_mul x 3072 2.85 -> 2.69 ms per
    // batch against the twelve-lane engine, x 2048 the same either way).  ms per step, narrow -> six: simple_mul x 4096
    // 4.06 -> 3.88 with six batches in flight, 3.76 with eight; below that size its few long waves lose: lookup_table x 2048
    // 3.48 -> 3.65, atms x 2048 3.92 -> 5.10.
    return launch_pairing_impl(impl, d, n, pts, valid, valid_sub, er, el_jac, status, accept, dbg, st, nullptr, prefer_narrow,
                               many ? S / 16.0 : in_flight_hint >= 4 ? S / 2.0 : S, in_flight_hint >= 4 && (double)n >= (many ? 3.0 : 4.0) * S, many);
}

// ---------------------------------------------------------------------------------------------- pipeline
// Enqueues the four kernels.  Without timings the transcript/combiner kernel (few, long waves) and the
// decompression kernel (many short ones) run concurrently on two streams and join before the MSM.
extern "C" int h2v_workspace_timings(h2v_workspace *w, uint32_t calls_back, h2v_timings *tm);
static int run_pipeline(const H2vDevPlan &d, uint32_t n, const uint8_t *proofs, const uint64_t *off, const uint8_t *inst,
                        const uint8_t *ci, uint8_t *accept, uint32_t *status_out, h2v_workspace *w, hipStream_t st,
                        h2v_timings *tm, bool want_trace) {
    opts_from(w);
    struct Reset { ~Reset() { g_opts = LaunchOptions(); } } reset_opts;
    const uint32_t slots = H2V_SLOTS(d);
    const uint32_t vm_blocks = (n + 63) / 64;
    const uint32_t dec_blocks = (n * slots + 63) / 64;
    (void)vm_blocks; (void)dec_blocks;
    uint32_t *trace = want_trace ? w->trace : nullptr;
    uint32_t *status = w->status;
    static const bool dbg = getenv("H2V_DEBUG_SYNC") != nullptr;  // serialise + sync after every kernel, say which one ran
    if (dbg) {
#define DBG_STAGE(name, launch)                                                                 \
        fprintf(stderr, "[h2v] launching %s (n=%u)\n", name, n); fflush(stderr);               \
        launch;                                                                                 \
        HIPCHK(hipGetLastError());                                                              \
        HIPCHK(hipDeviceSynchronize());                                                         \
        fprintf(stderr, "[h2v] %s done\n", name); fflush(stderr);
        DBG_STAGE("k_g1_decompress", hipLaunchKernelGGL(k_g1_decompress, dim3(dec_blocks), dim3(128), 0, st, d, n, proofs, off, ci, inst, w->pts, w->valid, w->pt_tab, 0u, (uint8_t *)nullptr))
        DBG_STAGE("k_transcript_combiner", { int rcv = launch_vm(d, n, w->stride, proofs, off, inst, ci, w->regs, w->scalars, status, trace, st); if (rcv) return rcv; })
        DBG_STAGE("k_g1_msm", launch_msm(d, n, w->scalars, w->pts, w->pt_tab, w->er, w->accl, w->accr, st))
        if (d.ivc) {
            const IvcBufs ib = {w->accl, w->accr, w->fold_pts, w->fold_scal, w->el2, w->er2};
            DBG_STAGE("ivc fold", launch_ivc_fold(d, n, w->pts, w->er, ib, w->msm_tab, st))
        }
        DBG_STAGE("k_pairing_check", launch_pairing(d, n, w->pts, w->valid, nullptr, d.ivc ? w->er2 : w->er, d.ivc ? w->el2 : nullptr, status, accept, nullptr, st))
#undef DBG_STAGE
        if (status_out) HIPCHK(hipMemcpyAsync(status_out, status, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        return H2V_OK;
    }
    // number of chunks (H2V_OPT_PIPES).  Default 1: measured on MI355X, 2/3/4 concurrent pipelines of a 4096-proof batch
    // took 14.2 / 20.7 / 29.1 ms against 13.5 ms for one (kernels with different private-segment sizes alternating on
    // several queues cost more than the idle SIMDs they fill), so the split is kept as an experiment knob only.
    const int dec_form = w->opt[H2V_OPT_DECOMPRESS_FORM];     // 0 / 1: two roles (queue launch / two launches); 2: one launch of paired blocks
    const bool split_dec = dec_form != 2;
    const int env_pipes = w->opt[H2V_OPT_PIPES];
    int pipes = env_pipes > 0 ? env_pipes : 1;
    if (pipes > h2v_workspace::MAXP) pipes = h2v_workspace::MAXP;
    if (want_trace || (uint32_t)pipes > n) pipes = 1;
    const int slot = (int)(w->calls % h2v_workspace::RING);
    w->ring_pipes[slot] = (uint8_t)pipes;
    w->ring_split[slot] = split_dec ? 1 : 0;
    w->calls++;
    HIPCHK(hipEventRecord(w->ev_fork, st));
    for (int k = 0; k < pipes; k++) {
        const uint32_t lo = (uint32_t)((uint64_t)n * k / pipes), hi = (uint32_t)((uint64_t)n * (k + 1) / pipes), m = hi - lo;
        hipEvent_t *ev = w->ring[slot][k];
        const bool dec_queue_on = dec_form == 0;
        // A caller that keeps >= 6 batches in flight (h2v_workspace_hint_in_flight) gets the whole pipeline on ITS stream: the
        // other batches fill the chip, and one stream per batch keeps many batches within the 16 hardware queues (three
        // streams per batch collide from the sixth batch on).  Measured, 40 steps of simple_mul x 4096: three streams 5 / 8 / 11
        // in flight 4.46 / 4.66 / 4.75 ms per step, one stream 4.80 / 4.39 / 4.50.  H2V_OPT_STREAMS forces the choice.
        const bool one_stream = pipes == 1 && (w->one_stream_mode >= 0 ? w->one_stream_mode == 1 : w->in_flight_hint >= 6);
        // one_stream_mode 2 (lanes): the pipeline on the stream it is given, only the decompression beside it on a side stream
        const bool two_stream = pipes == 1 && w->one_stream_mode == 2;
        if (!one_stream || two_stream)
            if (int rcs = ws_streams(w, k, !two_stream, true, !two_stream && split_dec && !dec_queue_on)) return rcs;
        hipStream_t pm = (one_stream || two_stream) ? st : w->pmain[k], ps = (one_stream && !two_stream) ? st : w->pside[k];
        const uint64_t *off_k = off + lo;
        const uint8_t *inst_k = inst ? inst + (size_t)lo * d.n_pi * 32 : nullptr;
        const uint8_t *ci_k = ci ? ci + (size_t)lo * 48 : nullptr;
        uint32_t *regs_k = w->regs + lo, *scal_k = w->scalars + (size_t)lo * d.n_terms * 8, *pts_k = w->pts + (size_t)lo * slots * 24;
        uint32_t *er_k = w->er + (size_t)lo * 36, *status_k = status + lo;
        uint8_t *valid_k = w->valid + (size_t)lo * slots, *accept_k = accept + lo;
        HIPCHK(hipStreamWaitEvent(pm, w->ev_fork, 0));
        HIPCHK(hipStreamWaitEvent(ps, w->ev_fork, 0));
        // decompression (many short waves) runs beside the transcript+combiner kernel (few long waves)
        uint32_t *pt_tab_k = w->pt_tab + (size_t)lo * slots * 448;
        uint8_t *vsub_k = split_dec ? w->valid_sub + (size_t)lo * slots : nullptr;
        const uint32_t dec_grid = (m * slots + 63) / 64;
        int rcv = 0;
        auto vm = [&]() {
            HIPCHK(hipEventRecord(ev[0], pm));
            rcv = launch_vm(d, m, w->stride, proofs, off_k, inst_k, ci_k, regs_k, scal_k, status_k, trace, pm);
            HIPCHK(hipEventRecord(ev[1], pm));
            return 0;
        };
        // Decompression: by default ONE launch whose waves (at most one per SIMD) take 64-point units from a queue - all
        // subgroup tests, then all square roots (k_g1_decompress_queue); H2V_OPT_DECOMPRESS_FORM = 1: the two halves as two
        // launches of 64-thread blocks on two streams; = 2: one launch of 128-thread blocks, a root wave and a subgroup
        // wave per block.
        const bool dec_queue = dec_form == 0;
        auto sqrt_half = [&]() {
            HIPCHK(hipEventRecord(ev[2], ps));
            if (split_dec && dec_queue) {
                uint32_t *ctr = w->dec_ctr + k;
                HIPCHK(hipMemsetAsync(ctr, 0, 4, ps));
                const uint32_t units = 2 * dec_grid, max_blocks = (uint32_t)(msm_n_simd() / 4.0);
                uint32_t blocks = (units + 3) / 4;
                if (blocks > max_blocks) blocks = max_blocks;
                hipLaunchKernelGGL(k_g1_decompress_queue, dim3(blocks), dim3(256), 0, ps, d, m, proofs, off_k, ci_k, inst_k, pts_k, valid_k, pt_tab_k, vsub_k, ctr, dec_grid);
            } else if (split_dec) hipLaunchKernelGGL(k_g1_decompress, dim3(dec_grid), dim3(64), 0, ps, d, m, proofs, off_k, ci_k, inst_k, pts_k, valid_k, pt_tab_k, 1u, (uint8_t *)nullptr);
            else hipLaunchKernelGGL(k_g1_decompress, dim3(dec_grid), dim3(128), 0, ps, d, m, proofs, off_k, ci_k, inst_k, pts_k, valid_k, pt_tab_k, 0u, (uint8_t *)nullptr);
            HIPCHK(hipEventRecord(ev[3], ps));
            HIPCHK(hipEventRecord(w->ev_join[k], ps));
            return 0;
        };
        auto sub_half = [&]() {
            // the subgroup tests as a launch of their own (only without the queue)
            if (!split_dec) return 0;
            hipStream_t pb = (dec_queue || one_stream || two_stream) ? ps : w->psub[k];   // (with the queue there is no third launch: only the events are recorded)
            HIPCHK(hipStreamWaitEvent(pb, w->ev_fork, 0));
            HIPCHK(hipEventRecord(ev[7], pb));
            if (!dec_queue) hipLaunchKernelGGL(k_g1_decompress, dim3(dec_grid), dim3(64), 0, pb, d, m, proofs, off_k, ci_k, inst_k, pts_k, valid_k, (uint32_t *)nullptr, 2u, vsub_k);
            HIPCHK(hipEventRecord(ev[8], pb));
            HIPCHK(hipEventRecord(w->ev_sub[k], pb));
            return 0;
        };
        const int rco = sqrt_half() || sub_half() || vm();
        if (rco) return rco;
        if (rcv) return rcv;
        HIPCHK(hipStreamWaitEvent(pm, w->ev_join[k], 0));
        // The MSM also waits for the subgroup launch although it does not read its result: a merged-halves MSM relies on
        // finding every SIMD empty (one wave each: 1.9 ms; a SIMD shared with a leftover wave: 2.2 ms for the launch)
        if (split_dec) HIPCHK(hipStreamWaitEvent(pm, w->ev_sub[k], 0));
        HIPCHK(hipEventRecord(ev[4], pm));
        uint32_t *tab_k = d.ivc ? w->msm_tab + (size_t)lo * 4 * 2 * 8 * 28 : nullptr;   // fold MSMs only
        const IvcBufs ib = {d.ivc ? w->accl + (size_t)lo * 36 : nullptr, d.ivc ? w->accr + (size_t)lo * 36 : nullptr,
                            d.ivc ? w->fold_pts + (size_t)lo * 96 : nullptr, d.ivc ? w->fold_scal + (size_t)lo * 32 : nullptr,
                            d.ivc ? w->el2 + (size_t)lo * 36 : nullptr, d.ivc ? w->er2 + (size_t)lo * 36 : nullptr};
        const MsmShape single = msm_ladder_shape(d.ivc ? d.n_terms : d.n_main_terms, m, 0.0, true);
        const MsmSplit split = w->er_fix ? msm_split_shape(d, m, single, w->in_flight_hint) : MsmSplit{false, {}, {}, 0};
        if (split.on) {
            // per-proof terms as ladders on the main stream; the VK-base terms beside them on the side stream (free since
            // the square roots finished), which first waits for the combiner's scalars; a one-lane-per-proof kernel adds
            // the two sums
            uint32_t *erf_k = w->er_fix + (size_t)lo * 36;
            H2vMsmArgs mv = {d.terms, 0, d.n_var, d.n_terms, 0, slots, {d.n_var, d.n_var, d.n_var}, {er_k, nullptr, nullptr}, pt_tab_k, d.vk_tab, nullptr, 0, 0};
            H2vMsmArgs mf = {d.terms, d.n_var, d.n_fix, d.n_terms, d.n_var, slots, {d.n_fix, d.n_fix, d.n_fix}, {erf_k, nullptr, nullptr}, pt_tab_k, d.vk_tab,
                             d.fix_tab, split.k, (d.n_fix + split.k - 1) / split.k};
            HIPCHK(hipStreamWaitEvent(ps, ev[1], 0));
            const uint32_t pbf = split.fix.bs / mf.n_fixl;
            HIPCHK(hipEventRecord(ev[9], ps));
            hipLaunchKernelGGL(k_g1_msm_fixed, dim3((m + pbf - 1) / pbf), dim3(split.fix.bs), (size_t)split.fix.bs * 172, ps, d, mf, m, pbf, scal_k, pts_k, (uint32_t *)nullptr);
            HIPCHK(hipEventRecord(ev[10], ps));
            HIPCHK(hipEventRecord(w->ev_fix[k], ps));
            if (ps == pm) HIPCHK(hipEventRecord(ev[4], pm));   // (one stream: the ladder launch starts behind the fixed-base one)
            uint32_t var_code;
            if (msm_terms_per_lane(w->in_flight_hint) > 1) var_code = launch_msm_range(d, mv, m, scal_k, pts_k, nullptr, pm, w->in_flight_hint);   // (several terms per lane)
            else var_code = launch_msm_ladders(d, mv, m, split.var, scal_k, pts_k, nullptr, pm);
            HIPCHK(hipEventRecord(ev[11], pm));
            w->ring_var[slot] = (uint8_t)var_code;
            HIPCHK(hipStreamWaitEvent(pm, w->ev_fix[k], 0));
            hipLaunchKernelGGL(k_g1_sum_pairs, dim3((m + 63) / 64), dim3(64), 0, pm, m, er_k, erf_k);
            w->ring_lpt[slot] = 3;
        } else {
            w->ring_lpt[slot] = (uint8_t)launch_msm(d, m, scal_k, pts_k, pt_tab_k, er_k, ib.accl, ib.accr, pm, w->in_flight_hint);
        }
        const uint32_t *er_in = er_k, *el_in = nullptr;
        if (d.ivc) {   // (timed with the MSM: the challenge hash and one more pass of the same kernel)
            launch_ivc_fold(d, m, pts_k, er_k, ib, tab_k, pm);
            er_in = ib.er2; el_in = ib.el2;
        }
        HIPCHK(hipEventRecord(ev[5], pm));
        w->ring_pair[slot] = (uint8_t)launch_pairing(d, m, pts_k, valid_k, vsub_k, er_in, el_in, status_k, accept_k, nullptr, pm, w->in_flight_hint);
        HIPCHK(hipEventRecord(ev[6], pm));
        HIPCHK(hipEventRecord(w->ev_done[k], pm));
        HIPCHK(hipStreamWaitEvent(st, w->ev_done[k], 0));
    }
    HIPCHK(hipGetLastError());
    if (status_out) HIPCHK(hipMemcpyAsync(status_out, status, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    if (tm) {
        HIPCHK(hipStreamSynchronize(st));
        int rc = h2v_workspace_timings(w, 0, tm);
        if (rc) return rc;
    }
    return H2V_OK;
}

static int run_rlc(const h2v_plan *p, uint32_t n, const uint8_t *proofs, const uint64_t *off, const uint8_t *inst, const uint8_t *ci,
                   uint8_t *accept, uint32_t *status_out, h2v_workspace *w, hipStream_t st, const uint32_t seed[8], bool one_stream_opt);
// ROUTING of RLC calls (round 4).  The batch-accept mode pays for every batch whose check fails: with 1 % rejecting proofs
// nearly half of the groups of 64 fail, and the call costs 1.3 x the per-proof mode it falls back to.  A workspace therefore
// keeps a running estimate of the rate of FAILING GROUPS among the groups its RLC calls have seen - cumulative device
// counters written by the batch / group checks (or, for a routed call, by k_rlc_count_groups from the status words), mirrored
// into pinned host memory behind every chunk and read without synchronisation at the head of the next call - and sends a
// call straight to the per-proof kernels while that rate is above 0.10 (back to the batch check below 0.05).  Break-even,
// measured (simple_mul x 4096, sixteen calls in flight, ms per call): honest 1.37; one failing group (1 reject) 2.74; 17 % of the
// groups failing (0.3 % rejects) 3.72; the per-proof mode 3.25 whatever it meets: about 2.6 + 6.4 x rate, equal at a rate of 0.1.
// Same accept[] either way; fell_back / h2v_workspace_rlc_result report which path ran.  H2V_OPT_RLC_ROUTE = -1: never route.
static int rlc_stats_ensure(h2v_workspace *w) {
    if (w->rlc_stats) return H2V_OK;
    if (hipMalloc((void **)&w->rlc_stats, 8) != hipSuccess || hipMemset(w->rlc_stats, 0, 8) != hipSuccess ||
        hipHostMalloc((void **)&w->h_rlc_stats, 8, hipHostMallocDefault) != hipSuccess)
        return fail(H2V_E_DEVICE, "allocation of the routing counters failed");
    w->h_rlc_stats[0] = w->h_rlc_stats[1] = 0;
    return H2V_OK;
}
static bool rlc_route(h2v_workspace *w) {
    if (w->opt[H2V_OPT_RLC_ROUTE] < 0 || !w->h_rlc_stats) return false;
    const volatile uint32_t *h = w->h_rlc_stats;
    const uint32_t g = h[0], f = h[1];
    const uint32_t dg = g - w->seen_groups, df = f - w->seen_failed;
    if (dg) {
        const float rate = df >= dg ? 1.0f : (float)df / (float)dg;
        w->fail_rate = 0.75f * w->fail_rate + 0.25f * rate;
        w->seen_groups = g; w->seen_failed = f;
    }
    w->routed = w->routed ? w->fail_rate > 0.05f : w->fail_rate > 0.10f;
    return w->routed;
}
// a routed chunk / call: the per-proof pipeline, then the counters from its status words
static int run_routed(const h2v_plan *p, uint32_t n, const uint8_t *proofs, const uint64_t *off, const uint8_t *inst, const uint8_t *ci,
                      uint8_t *accept, uint32_t *status_out, h2v_workspace *w, hipStream_t st, uint32_t *stats) {
    int rc = run_pipeline(p->d, n, proofs, off, inst, ci, accept, status_out, w, st, nullptr, false);
    if (rc) return rc;
    hipLaunchKernelGGL(k_rlc_count_groups, dim3(((n + 63) / 64 + 63) / 64), dim3(64), 0, st, n, w->status, stats);
    HIPCHK(hipGetLastError());
    return H2V_OK;
}
static uint64_t rlc_calls_of(const h2v_workspace *w);
static const uint32_t *rlc_flags_of(const h2v_workspace *w);
// A call on a laned workspace: chunks of at most w->chunk proofs, round robin through the lanes (continuing where the
// previous call stopped).  Chunk c runs on its lane's own stream(s) behind everything the caller had enqueued on `st`
// before the call and behind the lane's earlier chunks.  Nothing else orders the chunks: the kernels of neighbouring
// chunks drift apart by themselves (a decompression launch wants every SIMD, so the second one queues behind the first),
// and an explicit stagger - chunk c waiting for phase 1 of chunk c - 1 - measured 10-20 % slower.  rlc: every chunk is
// its own batch check.
// How many lanes a call of n proofs cycles through, and (per-proof mode) the streams of a chunk.  Chunks that give the MSM
// at most half a wave per SIMD are chains of lone waves - latency, not issue slots - and want as many of them in flight
// as there are streams: every lane, each chunk on its lane's stream alone.  Larger chunks: H2V_PER_PROOF_LANES lanes, the
// decompression on a side stream.  Measured (ms per batch; 8 lanes two streams each -> 16 lanes one stream each): secp256k1
// shape x 64 0.87 -> 0.57, x 128 1.12 -> 0.74; sha256 shape x 128 0.94 -> 0.74, x 256 1.29 -> 0.90, x 512 1.45 -> 1.24, secp256k1
// x 512 1.51 -> 1.34, simple_mul x 1024 1.48 -> 1.23; simple_mul x 2048 and sha256 x 1024: the same either way (2.16 / 2.18).
// An explicit lane count (h2v_workspace_create_lanes) is kept; H2V_OPT_STREAMS forces the streams.
static uint32_t laned_depth(const h2v_workspace *w, uint64_t n, bool rlc, int *stream_mode) {
    if (stream_mode) *stream_mode = 1;
    if (rlc) return w->n_lanes;
    const uint64_t m = n < w->chunk ? n : w->chunk;
    const bool small = (double)m * w->lane_plan.n_terms / 64.0 <= msm_n_simd() / 2.0;
    if (stream_mode) {
        *stream_mode = w->one_stream_mode >= 0 ? w->one_stream_mode : small ? 1 : 2;
    }
    return (small || w->lanes_per_proof >= w->n_lanes) ? w->n_lanes : w->lanes_per_proof;
}
static int run_laned(const h2v_plan *p, uint32_t n, const uint8_t *proofs, const uint64_t *off, const uint8_t *inst, const uint8_t *ci,
                     uint8_t *accept, uint32_t *status_out, h2v_workspace *w, hipStream_t st, bool rlc, const uint32_t *seed, bool force_join,
                     bool never_join = false) {
    const H2vDevPlan &d = p->d;
    if (int rcf = co_flush(w)) return rcf;          // (calls start in submission order: an open group of coalesced calls first)
    int stream_mode = 1;
    const uint32_t L = laned_depth(w, n, rlc, &stream_mode);
    // A per-proof call that would leave lanes empty at the workspace's chunk size is cut finer, one chunk per lane (in 512s,
    // not below 2048: there the chains' latency costs more than the extra lanes win).  One call on 20 480 simple_mul proofs,
    // eight lanes: chunks of 4096 24.0 ms, 2560 22.5 ms, 2048 27.7 ms.  (RLC mode: every chunk is a batch check of its own and
    // 4096 measured best - 45 056 proofs in 4096s 19.3 ms, in 2816s 20.6 ms.)
    uint32_t chunk = w->chunk;
    // (only for a call that waits for its own chunks: with deferred joins or streamed host batches the NEXT call fills the lanes,
    //  and whole chunks keep the full-chip launch shapes - a stream of 16 384-proof calls: 15.9 ms per call cut in 2048s, ~13 in 4096s)
    if (!rlc && !w->defer_joins && !never_join && n > chunk && (uint64_t)n < (uint64_t)L * chunk) {
        const uint32_t c2 = (uint32_t)(((n + L - 1) / L + 511) / 512 * 512);
        if (c2 >= default_chunk(d) / 2 && c2 < chunk) chunk = c2;       // (simple_mul: 2048; T = 60: 512)
    }
    const uint32_t nch = (n + chunk - 1) / chunk;
    // What the chunks' launch shapes may assume about the chip: with deferred joins (and the host-buffer stream of batches)
    // consecutive calls overlap and every lane is busy; a call that waits for its own chunks has only THOSE in flight - one
    // 4096-proof chunk on a laned workspace is a lone batch (with the full-chip shapes it took 11.6 ms instead of 5.8).
    const uint32_t call_hint = w->hint_given ? w->in_flight_hint : (w->defer_joins || never_join) ? w->n_lanes : (nch < L ? nch : L);
    const int slot = (int)(w->calls % h2v_workspace::RING);
    w->calls++;
    w->lring_chunks[slot] = nch; w->lring_first[slot] = (uint32_t)(w->next_lane % L); w->lring_mod[slot] = L; w->lring_rlc[slot] = rlc ? 1 : 0;
    w->lring_co[slot] = 0;
    bool routed = false;
    if (rlc) {
        if (int rcs = rlc_stats_ensure(w)) return rcs;
        routed = rlc_route(w);
        w->lring_routed[slot] = routed ? 1 : 0;
        HIPCHK(hipMemsetAsync(w->rlc_fail + slot, routed ? 1 : 0, 4, st));      // (routed: "a batch check failed" = the per-proof kernels produce accept[])
    }
    HIPCHK(hipEventRecord(w->ev_fork, st));
    for (uint32_t c = 0; c < nch; c++) {
        const uint32_t l = (uint32_t)(w->next_lane++ % L);
        int rc = ensure_lane(w, l);
        if (rc) return rc;
        h2v_workspace *lw = w->lane[l];
        hipStream_t ls = w->lane_st[l];
        const uint32_t lo = c * chunk, m = (n - lo) < chunk ? (n - lo) : chunk;
        HIPCHK(hipStreamWaitEvent(ls, w->ev_fork, 0));
        const uint8_t *inst_c = inst ? inst + (size_t)lo * d.n_pi * 32 : nullptr, *ci_c = ci ? ci + (size_t)lo * 48 : nullptr;
        if (rlc) {
            uint32_t sd[8];
            for (int k = 0; k < 8; k++) sd[k] = seed[k];
            sd[7] ^= 0x9e3779b9u * (c + 1);      // (a chunk is its own batch check: its own coefficients)
            lw->rlc_fail_ptr = w->rlc_fail + slot;
            lw->rlc_stats_ptr = w->rlc_stats;
            lw->in_flight_hint = call_hint;
            lw->opt[H2V_OPT_RLC_GROUP_STAGE] = w->opt[H2V_OPT_RLC_GROUP_STAGE];
            if (routed) {
                lw->one_stream_mode = 1;
                rc = run_routed(p, m, proofs, off + lo, inst_c, ci_c, accept + lo, status_out ? status_out + lo : nullptr, lw, ls, w->rlc_stats);
            } else {
                rc = run_rlc(p, m, proofs, off + lo, inst_c, ci_c, accept + lo, status_out ? status_out + lo : nullptr, lw, ls, sd, true);
            }
            if (rc == H2V_OK && hipMemcpyAsync(w->h_rlc_stats, w->rlc_stats, 8, hipMemcpyDeviceToHost, ls) != hipSuccess) rc = fail(H2V_E_DEVICE, "routing counters: copy failed");
        } else {
            lw->one_stream_mode = stream_mode;
            lw->in_flight_hint = call_hint;
            rc = run_pipeline(d, m, proofs, off + lo, inst_c, ci_c, accept + lo, status_out ? status_out + lo : nullptr, lw, ls, nullptr, false);
        }
        if (rc) {
            // earlier chunks of this call (and of calls before it) are still in flight on other lanes: nothing of the caller's may
            // be reused before they have drained, so an error return waits for every lane
            const std::string e = g_err;
            for (uint32_t q = 0; q < w->n_lanes; q++) if (w->lane_st[q]) (void)hipStreamSynchronize(w->lane_st[q]);
            for (uint32_t q = 0; q < w->n_lanes; q++) w->lane_busy[q] = false;
            return fail(rc, e);
        }
        HIPCHK(hipEventRecord(w->lane_ev[l], ls));
        w->lane_busy[l] = true;
    }
    for (uint32_t l = 0; l < w->n_lanes; l++) {
        w->lring_calls[slot][l] = w->lane[l] ? w->lane[l]->calls : 0;
        w->lring_rlc_calls[slot][l] = w->lane[l] ? rlc_calls_of(w->lane[l]) : 0;
        w->lring_rlc_obj[slot][l] = w->lane[l] ? w->lane[l]->rlc : nullptr;
    }
    if (never_join) return H2V_OK;
    if (!w->defer_joins || force_join) return lanes_join(w, st);
    return H2V_OK;
}
// ---- coalescing of small calls (round 4; VERDICT r3 #6: "let lanes pack units of several small calls of the same plan into one
// launch per kernel").  A device-resident call (per proof, or RLC among RLC calls) of at most HALF a chunk on a laned workspace with deferred joins is not
// launched by itself: its proofs are gathered (h2v_coalesce.hpp, on a lane's stream, behind whatever the caller's stream held at
// the time of the call - so the caller's buffers are read at once, as always) behind those of the calls before it, and the
// pipeline runs ONCE over the group - when the next call would not fit, when a call of another plan or kind arrives, or at
// h2v_workspace_join - after which every call's accept[] / status[] are copied to where the caller wanted them.  That is within
// the contract of deferred joins as it stood (results are the caller's after h2v_workspace_join); what changes is how the work
// is cut: sixteen 64-proof calls are one 1024-proof launch per kernel instead of sixteen chains of lone waves.  Verdicts cannot
// depend on it (a proof's verdict depends on its own bytes; tests/test_gpu_parity.py::test_small_calls_are_coalesced).
// H2V_OPT_COALESCE = -1 on the workspace switches it off.
struct Coalesce {
    uint8_t *proofs = nullptr, *inst = nullptr, *ci = nullptr, *accept = nullptr;
    uint64_t *off = nullptr;
    uint32_t *status = nullptr;
    size_t cap_proof_bytes = 0, cap_inst = 0;
    uint32_t cap = 0;                    // proofs
    const h2v_plan *plan = nullptr;      // of the open group
    uint64_t plan_gen = 0;
    uint32_t count = 0;
    bool rlc = false;                    // the open group's mode: per proof, or ONE batch check over the whole group
    uint32_t seed[8] = {};               // (RLC: the first call's - one fresh seed makes every coefficient unpredictable)
    struct Part { uint8_t *accept; uint32_t *status; uint32_t base, n; int slot; };
    std::vector<Part> parts;
};
static void co_release(h2v_workspace *w) {
    for (auto &c : w->co) {
        if (!c) continue;
        void *ptrs[] = {c->proofs, c->inst, c->ci, c->accept, c->off, c->status};
        for (void *q : ptrs) if (q) (void)hipFree(q);
        delete c;
        c = nullptr;
    }
    w->co_open.clear();
}
// the staging buffers of lane l for groups of plan p (grown when a plan with longer proofs / more public inputs comes along)
static int co_ensure(h2v_workspace *w, uint32_t l, const h2v_plan *p) {
    if (!w->co[l]) w->co[l] = new Coalesce();
    Coalesce &c = *w->co[l];
    const uint32_t cap = w->chunk;
    const size_t need_p = (size_t)cap * p->d.proof_len + 64, need_i = (size_t)cap * (p->d.n_pi ? p->d.n_pi : 1) * 32;
    if (c.cap == cap && c.cap_proof_bytes >= need_p && c.cap_inst >= need_i) return H2V_OK;
    HIPCHK(hipStreamSynchronize(w->lane_st[l]));              // (an earlier group may still be running out of the old buffers)
    void *ptrs[] = {c.proofs, c.inst, c.ci, c.accept, c.off, c.status};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    c.proofs = c.inst = c.ci = c.accept = nullptr; c.off = nullptr; c.status = nullptr; c.cap = 0;
    const size_t bp = need_p > c.cap_proof_bytes ? need_p : c.cap_proof_bytes, bi = need_i > c.cap_inst ? need_i : c.cap_inst;
    bool ok = hipMalloc((void **)&c.proofs, bp) == hipSuccess && hipMalloc((void **)&c.inst, bi) == hipSuccess &&
              hipMalloc((void **)&c.ci, (size_t)cap * 48) == hipSuccess && hipMalloc((void **)&c.accept, cap) == hipSuccess &&
              hipMalloc((void **)&c.off, ((size_t)cap + 1) * 8) == hipSuccess && hipMalloc((void **)&c.status, (size_t)cap * 4) == hipSuccess;
    if (!ok) return fail(H2V_E_DEVICE, "hipMalloc(coalescing buffers) failed");
    c.cap = cap; c.cap_proof_bytes = bp; c.cap_inst = bi;
    return H2V_OK;
}
// runs every open group, oldest first
static int co_flush_lane(h2v_workspace *w, uint32_t l);
static int co_flush(h2v_workspace *w) {
    while (!w->co_open.empty())
        if (int rc = co_flush_lane(w, w->co_open.front())) return rc;
    return H2V_OK;
}
// runs the open group of lane l: ONE pass of the pipeline over everything gathered so far, then every call's verdicts to the caller's buffers
static int co_flush_lane(h2v_workspace *w, uint32_t l) {
    for (size_t k = 0; k < w->co_open.size(); k++)
        if (w->co_open[k] == l) { w->co_open.erase(w->co_open.begin() + (long)k); break; }
    Coalesce &c = *w->co[l];
    if (c.count == 0) return H2V_OK;
    h2v_workspace *lw = w->lane[l];
    hipStream_t ls = w->lane_st[l];
    int stream_mode = 1;
    (void)laned_depth(w, c.count, false, &stream_mode);
    lw->one_stream_mode = stream_mode;
    lw->in_flight_hint = w->hint_given ? w->in_flight_hint : w->n_lanes;
    const h2v_plan *p = c.plan;
    const uint8_t *inst_g = p->d.n_pi ? c.inst : nullptr, *ci_g = p->d.n_ci ? c.ci : nullptr;
    int rc = H2V_OK;
    bool routed = false;
    const int slot0 = c.parts.empty() ? 0 : c.parts.front().slot;
    if (c.rlc) {
        // one batch check over the group (as a chunk of run_laned): the verdict counter of the group's FIRST call takes the
        // kernel's report and is copied to the other calls' counters behind it
        if ((rc = rlc_stats_ensure(w)) == H2V_OK) {
            routed = rlc_route(w);
            if (hipMemsetAsync(w->rlc_fail + slot0, routed ? 1 : 0, 4, ls) != hipSuccess) rc = fail(H2V_E_DEVICE, "memset failed");
        }
        if (rc == H2V_OK) {
            lw->rlc_fail_ptr = w->rlc_fail + slot0;
            lw->rlc_stats_ptr = w->rlc_stats;
            lw->opt[H2V_OPT_RLC_GROUP_STAGE] = w->opt[H2V_OPT_RLC_GROUP_STAGE];
            if (routed) { lw->one_stream_mode = 1; rc = run_routed(p, c.count, c.proofs, c.off, inst_g, ci_g, c.accept, c.status, lw, ls, w->rlc_stats); }
            else rc = run_rlc(p, c.count, c.proofs, c.off, inst_g, ci_g, c.accept, c.status, lw, ls, c.seed, true);
            if (rc == H2V_OK && hipMemcpyAsync(w->h_rlc_stats, w->rlc_stats, 8, hipMemcpyDeviceToHost, ls) != hipSuccess) rc = fail(H2V_E_DEVICE, "routing counters: copy failed");
        }
    } else {
        rc = run_pipeline(p->d, c.count, c.proofs, c.off, inst_g, ci_g, c.accept, c.status, lw, ls, nullptr, false);
    }
    for (const Coalesce::Part &q : c.parts) {
        if (rc) break;
        if (hipMemcpyAsync(q.accept, c.accept + q.base, q.n, hipMemcpyDeviceToDevice, ls) != hipSuccess ||
            (q.status && hipMemcpyAsync(q.status, c.status + q.base, (size_t)q.n * 4, hipMemcpyDeviceToDevice, ls) != hipSuccess) ||
            (c.rlc && q.slot != slot0 && hipMemcpyAsync(w->rlc_fail + q.slot, w->rlc_fail + slot0, 4, hipMemcpyDeviceToDevice, ls) != hipSuccess))
            rc = fail(H2V_E_DEVICE, "coalesced calls: copying the verdicts out failed");
        // the call's record: its group ran as this lane's most recent call; its share of that launch
        w->lring_share[q.slot] = (float)q.n / (float)c.count;
        w->lring_routed[q.slot] = routed ? 1 : 0;
        for (uint32_t k = 0; k < w->n_lanes; k++) {
            w->lring_calls[q.slot][k] = w->lane[k] ? w->lane[k]->calls : 0;
            w->lring_rlc_calls[q.slot][k] = w->lane[k] ? rlc_calls_of(w->lane[k]) : 0;
            w->lring_rlc_obj[q.slot][k] = w->lane[k] ? w->lane[k]->rlc : nullptr;
        }
    }
    c.parts.clear();
    c.count = 0;
    c.plan = nullptr;
    if (rc) {
        const std::string e = g_err;
        for (uint32_t q = 0; q < w->n_lanes; q++) if (w->lane_st[q]) (void)hipStreamSynchronize(w->lane_st[q]);
        for (uint32_t q = 0; q < w->n_lanes; q++) w->lane_busy[q] = false;
        return fail(rc, e);
    }
    HIPCHK(hipEventRecord(w->lane_ev[l], ls));
    w->lane_busy[l] = true;
    return H2V_OK;
}
static bool co_wanted(const h2v_workspace *w, const h2v_plan *p, uint64_t n) {
    return w->n_lanes && w->defer_joins && w->opt[H2V_OPT_COALESCE] >= 0 && n * 2 <= w->chunk && !w->pending;
    (void)p;
}
// one small call: gathered into the open group (opening one, or running the open one first when this call does not fit it)
static int coalesce_call(const h2v_plan *p, const h2v_batch *b, uint8_t *accept, uint32_t *status, h2v_workspace *w, hipStream_t st, bool rlc = false,
                         const uint32_t *seed = nullptr) {
    const uint32_t n = (uint32_t)b->n;
    // the open group of this plan and mode (a workspace that serves several plans keeps one group per plan open: calls that
    // alternate between two plans would otherwise run every group at one call)
    int found = -1;
    for (uint32_t q : w->co_open) {
        const Coalesce &o = *w->co[q];
        if (o.plan == p && o.plan_gen == p->gen && o.rlc == rlc) { found = (int)q; break; }
    }
    if (found >= 0 && w->co[found]->count + n > w->co[found]->cap) {
        if (int rcf = co_flush_lane(w, (uint32_t)found)) return rcf;
        found = -1;
    }
    if (found < 0) {
        int sm = 1;
        const uint32_t L = laned_depth(w, w->chunk, false, &sm);
        // at most four groups open, and never as many as there are lanes to put them on (a workspace of three lanes that served
        // five plans opened a group on a lane that still held one: its calls were never run - found by tests/soak.py)
        const size_t max_open = L < 4 ? L : 4;
        while (w->co_open.size() >= max_open)
            if (int rcf = co_flush_lane(w, w->co_open.front())) return rcf;
        uint32_t l = (uint32_t)(w->next_lane++ % L);
        for (uint32_t tries = 0; tries < L; tries++) {          // (a lane whose group is still open is not a place for another)
            bool taken = false;
            for (uint32_t q : w->co_open) taken = taken || q == l;
            if (!taken) break;
            l = (uint32_t)(w->next_lane++ % L);
        }
        int rc = ensure_lane(w, l);
        if (rc) return rc;
        if ((rc = co_ensure(w, l, p))) return rc;
        Coalesce &c = *w->co[l];
        c.plan = p; c.plan_gen = p->gen; c.count = 0; c.parts.clear();
        c.rlc = rlc;
        for (int k = 0; k < 8; k++) c.seed[k] = rlc && seed ? seed[k] : 0;
        HIPCHK(hipMemsetAsync(c.off, 0, 8, w->lane_st[l]));     // off[0] = 0: behind the previous group's pipeline on this stream
        w->co_open.push_back(l);
        found = (int)l;
    }
    const uint32_t l = (uint32_t)found;
    Coalesce &c = *w->co[l];
    hipStream_t ls = w->lane_st[l];
    HIPCHK(hipEventRecord(w->ev_fork, st));
    HIPCHK(hipStreamWaitEvent(ls, w->ev_fork, 0));
    hipLaunchKernelGGL(k_coalesce_offsets, dim3(1), dim3(256), 0, ls, b->proof_off, n, p->d.proof_len, c.off + c.count);
    hipLaunchKernelGGL(k_coalesce_copy, dim3(n), dim3(256), 0, ls, b->proofs, b->proof_off, c.proofs, c.off + c.count, n);
    HIPCHK(hipGetLastError());
    if (p->d.n_pi) HIPCHK(hipMemcpyAsync(c.inst + (size_t)c.count * p->d.n_pi * 32, b->instances, (size_t)n * p->d.n_pi * 32, hipMemcpyDeviceToDevice, ls));
    if (p->d.n_ci) HIPCHK(hipMemcpyAsync(c.ci + (size_t)c.count * 48, b->committed, (size_t)n * 48, hipMemcpyDeviceToDevice, ls));
    // the call's record in the ring: one "chunk" on lane l, that lane's NEXT pipeline call (nothing else runs there before the flush)
    const int slot = (int)(w->calls % h2v_workspace::RING);
    w->calls++;
    w->lring_chunks[slot] = 1; w->lring_first[slot] = l; w->lring_mod[slot] = w->n_lanes; w->lring_rlc[slot] = rlc ? 1 : 0; w->lring_routed[slot] = 0;
    w->lring_co[slot] = 1; w->lring_share[slot] = 0.0f;      // (0: its group has not run yet)
    for (uint32_t q = 0; q < w->n_lanes; q++) { w->lring_calls[slot][q] = 0; w->lring_rlc_calls[slot][q] = 0; w->lring_rlc_obj[slot][q] = nullptr; }   // (filled in by the flush)
    c.parts.push_back({accept, status, c.count, n, slot});
    c.count += n;
    w->lane_busy[l] = true;              // (h2v_workspace_join must look at this lane; its event is recorded by the flush)
    if (c.count >= c.cap) return co_flush_lane(w, l);
    return H2V_OK;
}

// chunk c of the call in ring slot `slot` ran on lane *l as that lane's call number (0-based, absolute) *idx
static void laned_chunk_pos(const h2v_workspace *w, int slot, uint32_t c, bool rlc, uint32_t *l, uint64_t *idx) {
    const uint32_t L = w->lring_mod[slot], nch = w->lring_chunks[slot], f = w->lring_first[slot];
    *l = (f + c) % L;
    uint32_t uses = 0;                       // chunks of this call on that lane
    for (uint32_t q = 0; q < nch; q++) if ((f + q) % L == *l) uses++;
    const uint64_t end = rlc ? w->lring_rlc_calls[slot][*l] : w->lring_calls[slot][*l];
    *idx = end - uses + c / L;
}

extern "C" int h2v_verify_batch_device(const h2v_plan *p, const h2v_batch *b, uint8_t *accept, uint32_t *status,
                                       h2v_workspace *ws, void *stream, h2v_timings *timings) {
    if (!p || !b || !accept) return fail(H2V_E_ARG, "null argument");
    if (b->n == 0) return H2V_OK;
    if (!b->proofs || !b->proof_off) return fail(H2V_E_ARG, "null proofs / offsets");
    if (p->d.n_pi && !b->instances) return fail(H2V_E_ARG, "plan has public inputs but instances == NULL");
    if (p->d.n_ci && !b->committed) return fail(H2V_E_ARG, "plan has a committed instance but committed == NULL");
    ALIVE(p); ALIVE(ws);
    HIPCHK(hipSetDevice(p->device));
    h2v_workspace *tmp = nullptr;
    if (!ws) {
        int rc = ws_create_for(p->d, p->device, b->n, false, &tmp);
        if (rc) return rc;
        ws = tmp;
    }
    if (int rcf = ws_fits(ws, p, b->n, false)) { if (tmp) h2v_workspace_free(tmp); return rcf; }
    if (ws->pending) return fail(H2V_E_ARG, "the workspace has a host batch in flight: call h2v_verify_batch_wait first");
    if (int rcn = null_stream_check(ws, stream)) { if (tmp) h2v_workspace_free(tmp); return rcn; }
    if (ws->n_lanes) {
        if (!timings && co_wanted(ws, p, b->n)) return coalesce_call(p, b, accept, status, ws, (hipStream_t)stream);
        int rcl = run_laned(p, (uint32_t)b->n, b->proofs, b->proof_off, b->instances, b->committed, accept, status, ws, (hipStream_t)stream, false, nullptr, timings != nullptr);
        if (rcl == H2V_OK && timings) {
            HIPCHK(hipStreamSynchronize((hipStream_t)stream));
            rcl = h2v_workspace_timings(ws, 0, timings);
        }
        return rcl;
    }
    int rc = run_pipeline(p->d, (uint32_t)b->n, b->proofs, b->proof_off, b->instances, b->committed, accept, status, ws,
                          (hipStream_t)stream, timings, false);
    if (tmp) {
        if (rc == H2V_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = fail(H2V_E_DEVICE, "stream synchronize failed");
        h2v_workspace_free(tmp);
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------- measured launch shapes
// h2v_workspace_tune: instead of trusting the launcher's thresholds (calibrated on five shapes on one chip), MEASURE the
// candidate shapes for THIS plan, batch size and workspace regime once and leave the winner on the workspace.  The batch is
// run repeatedly in the way the workspace will be used - a laned workspace with as many calls in flight as it has lanes for
// that size (deferred joins), an ordinary one call by call - because the shapes interact: the six-lane pairing engine is the
// slowest engine alone and the fastest one in a full pipeline.  Coordinate descent over the two dimensions that matter per
// call (pairing engine, MSM terms per lane): the launcher's own choice first, then each neighbouring engine, then each
// terms-per-lane value with the best engine.  A candidate replaces the incumbent only if it is more than 3 % faster.
// accept[] goes to a scratch buffer: verdicts do not depend on shapes (tests/test_gpu_parity.py).
extern "C" int h2v_workspace_tune(const h2v_plan *p, const h2v_batch *b, h2v_workspace *ws, void *stream, uint32_t flags, h2v_tune_report *rep) {
    if (!p || !b || !ws) return fail(H2V_E_ARG, "null argument");
    ALIVE(p); ALIVE(ws);
    if (flags != 0) return fail(H2V_E_ARG, "flags must be 0 (the per-proof mode is what is tuned)");
    if (b->n == 0 || !b->proofs || !b->proof_off) return fail(H2V_E_ARG, "tuning needs a representative device-resident batch");
    if (p->d.n_pi && !b->instances) return fail(H2V_E_ARG, "plan has public inputs but instances == NULL");
    if (p->d.n_ci && !b->committed) return fail(H2V_E_ARG, "plan has a committed instance but committed == NULL");
    HIPCHK(hipSetDevice(p->device));
    if (int rcf = ws_fits(ws, p, b->n, false)) return rcf;
    if (ws->pending) return fail(H2V_E_ARG, "the workspace has a host batch in flight: call h2v_verify_batch_wait first");
    hipStream_t st = (hipStream_t)stream;
    const bool laned = ws->n_lanes != 0;
    if (laned && !st) return fail(H2V_E_ARG, "tuning a laned workspace keeps calls in flight: it needs a stream of its own, not the legacy NULL stream");
    const uint32_t n = (uint32_t)b->n;
    const bool co = laned && ws->opt[H2V_OPT_COALESCE] >= 0 && (uint64_t)n * 2 <= ws->chunk;     // (small calls run gathered: that is the regime to measure)
    const uint32_t depth = !laned ? 1u : co ? (ws->chunk / n) * laned_depth(ws, ws->chunk, false, nullptr) : laned_depth(ws, n, false, nullptr);   // (co: a group on every lane)
    const uint32_t m = laned && n > ws->chunk ? ws->chunk : co ? ws->chunk / n * n : n;         // proofs per launch
    const double S = msm_n_simd();
    uint8_t *acc_tmp = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipMalloc((void **)&acc_tmp, n) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (acc_tmp) (void)hipFree(acc_tmp);
        if (e0) (void)hipEventDestroy(e0);
        return fail(H2V_E_DEVICE, "tuning: allocation failed");
    }
    const bool saved_defer = ws->defer_joins;
    const int32_t saved_pair = ws->opt[H2V_OPT_PAIRING_ENGINE], saved_tpl = ws->opt[H2V_OPT_MSM_TERMS_PER_LANE];
    int rc = H2V_OK;
    auto set2 = [&](int32_t pairing, int32_t tpl) {
        (void)h2v_workspace_set_option(ws, H2V_OPT_PAIRING_ENGINE, pairing);
        (void)h2v_workspace_set_option(ws, H2V_OPT_MSM_TERMS_PER_LANE, tpl);
    };
    auto round_of_calls = [&](uint32_t calls) -> int {
        for (uint32_t k = 0; k < calls; k++) {
            int r = co ? coalesce_call(p, b, acc_tmp, nullptr, ws, st)
                  : laned ? run_laned(p, n, b->proofs, b->proof_off, b->instances, b->committed, acc_tmp, nullptr, ws, st, false, nullptr, false)
                          : run_pipeline(p->d, n, b->proofs, b->proof_off, b->instances, b->committed, acc_tmp, nullptr, ws, st, nullptr, false);
            if (r) return r;
        }
        return laned ? lanes_join(ws, st) : H2V_OK;
    };
    uint32_t n_meas = 0;
    float per_call_ms = 0;                    // (the first measurement's result)
    auto measure = [&](int32_t pairing, int32_t tpl, float *ms) -> int {
        set2(pairing, tpl);
        if (laned) ws->defer_joins = true;
        int r = round_of_calls(depth);                                  // fills the lanes (and creates them)
        if (r) return r;
        // three rounds of the lanes, and at least ~120 ms once the first measurement has told what a call takes: the candidates of a
        // small batch differ by a few per cent of a fraction of a millisecond (secp256k1 shape x 256: the 38 ms measurements of
        // round 4's first tuner picked the narrow engine in one run and the normal one, 13 % faster, in the next)
        uint32_t calls = laned ? 3 * depth : 4;
        if (per_call_ms > 0) {
            const uint32_t want = (uint32_t)(120.0f / per_call_ms) + 1;
            calls = want > calls ? (want < 40 * depth ? want : 40 * depth) : calls;
            if (co) calls = (calls + depth - 1) / depth * depth;     // (whole rounds of full groups)
        }
        if (hipEventRecord(e0, st) != hipSuccess) return fail(H2V_E_DEVICE, "event record failed");
        if ((r = round_of_calls(calls))) return r;
        if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return fail(H2V_E_DEVICE, "tuning: the measured calls failed");
        float t = 0;
        if (hipEventElapsedTime(&t, e0, e1) != hipSuccess) return fail(H2V_E_DEVICE, "event timing failed");
        *ms = t / (float)calls;
        if (per_call_ms == 0) per_call_ms = *ms;
        n_meas++;
        return H2V_OK;
    };
    float best_ms = 0, def_ms = 0;
    int32_t best_pair = 0, best_tpl = 0;
    do {
        if ((rc = measure(0, 0, &def_ms))) break;
        if (laned && 120.0f / def_ms > 3.0f * (float)depth && (rc = measure(0, 0, &def_ms))) break;   // (the baseline at the candidates' length)
        best_ms = def_ms;
        const int32_t full[] = {6, 12, 16}, mid[] = {12, 16, 32}, low[] = {32, 64};
        const int32_t *eng = (double)m >= 2.0 * S ? full : (double)m >= S / 4.0 ? mid : low;
        const int n_eng = (double)m >= S / 4.0 ? 3 : 2;
        // (the fastest candidate replaces the launcher's rule when it is 3 % faster than THAT - not each candidate against the best so
        //  far, which favoured whichever engine happened to be measured first)
        float cand_ms = 0;
        int32_t cand = 0;
        for (int k = 0; k < n_eng && !rc; k++) {
            float t;
            if ((rc = measure(eng[k], 0, &t))) break;
            if (cand == 0 || t < cand_ms) { cand_ms = t; cand = eng[k]; }
        }
        if (rc) break;
        if (cand && cand_ms < best_ms * 0.97f) { best_ms = cand_ms; best_pair = cand; }
        const bool ladders_share = !p->d.ivc && (double)m * p->d.n_main_terms / 64.0 >= S / 8.0;   // (k_g1_msm_multi's own preconditions)
        float cand_t_ms = 0;
        int32_t cand_tpl = 0;
        for (int32_t tpl = 2; tpl <= 4 && ladders_share && !rc; tpl++) {
            float t;
            if ((rc = measure(best_pair, tpl, &t))) break;
            if (cand_tpl == 0 || t < cand_t_ms) { cand_t_ms = t; cand_tpl = tpl; }
        }
        if (!rc && cand_tpl && cand_t_ms < best_ms * 0.97f) { best_ms = cand_t_ms; best_tpl = cand_tpl; }
    } while (0);
    ws->defer_joins = saved_defer;
    if (rc == H2V_OK) set2(best_pair, best_tpl); else set2(saved_pair, saved_tpl);
    (void)hipStreamSynchronize(st);
    (void)hipFree(acc_tmp);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc == H2V_OK && rep) {
        memset(rep, 0, sizeof *rep);
        rep->n_measured = n_meas; rep->default_ms = def_ms; rep->best_ms = best_ms; rep->calls_in_flight = depth;
        rep->pairing_engine = best_pair; rep->msm_terms_per_lane = best_tpl;
    }
    return rc;
}

// Per-kernel device times of a PAST call on this workspace (0 = the most recent, up to 63 back).  The events were
// recorded on the streams the kernels ran on; the caller must have synchronised the launch stream first.
extern "C" int h2v_workspace_timings(h2v_workspace *w, uint32_t calls_back, h2v_timings *tm) {
    if (!w || !tm) return fail(H2V_E_ARG, "null argument");
    ALIVE(w);
    if (calls_back >= h2v_workspace::RING || calls_back >= w->calls) return fail(H2V_E_ARG, "no such call in the event ring");
    HIPCHK(hipSetDevice(w->device));
    const int slot = (int)((w->calls - 1 - calls_back) % h2v_workspace::RING);
    if (w->n_lanes) {
        // sums over the call's chunks (h2v_timings: `launches` = chunks); total_ms = first chunk's start .. the last end
        if (w->lring_co[slot] && w->lring_share[slot] == 0.0f) return fail(H2V_E_ARG, "that call is in a group of coalesced calls that has not run yet: h2v_workspace_join first");
        if (w->lring_rlc[slot] && !w->lring_routed[slot]) return fail(H2V_E_ARG, "that call ran in RLC mode: h2v_workspace_rlc_result");
        memset(tm, 0, sizeof *tm);
        const uint32_t nch = w->lring_chunks[slot];
        hipEvent_t first = nullptr;
        for (uint32_t c = 0; c < nch; c++) {
            uint32_t l; uint64_t idx;
            laned_chunk_pos(w, slot, c, false, &l, &idx);
            h2v_workspace *lw = w->lane[l];
            if (lw->calls - 1 - idx >= (uint64_t)h2v_workspace::RING) return fail(H2V_E_ARG, "the lanes' event rings have wrapped since that call");
            h2v_timings t1;
            int rc = h2v_workspace_timings(lw, (uint32_t)(lw->calls - 1 - idx), &t1);
            if (rc) return rc;
            tm->transcript_combiner_ms += t1.transcript_combiner_ms; tm->g1_decompress_ms += t1.g1_decompress_ms;
            tm->g1_msm_ms += t1.g1_msm_ms; tm->pairing_ms += t1.pairing_ms; tm->g1_msm_fixed_ms += t1.g1_msm_fixed_ms;
            tm->msm_var_lanes_per_term = t1.msm_var_lanes_per_term;
            tm->msm_lanes_per_term = t1.msm_lanes_per_term; tm->pairing_lanes_per_proof = t1.pairing_lanes_per_proof;
            hipEvent_t *ev = lw->ring[idx % h2v_workspace::RING][0];
            if (!first) first = ev[2];
            float span = 0;
            HIPCHK(hipEventElapsedTime(&span, first, ev[6]));
            if (span > tm->total_ms) tm->total_ms = span;
        }
        tm->launches = nch;
        if (w->lring_co[slot]) {     // a coalesced call: its share of the one launch that served its group
            const float f = w->lring_share[slot];
            tm->transcript_combiner_ms *= f; tm->g1_decompress_ms *= f; tm->g1_msm_ms *= f; tm->pairing_ms *= f; tm->g1_msm_fixed_ms *= f;
        }
        return H2V_OK;
    }
    const int pipes = w->ring_pipes[slot];
    memset(tm, 0, sizeof *tm);
    tm->launches = (uint32_t)pipes;
    tm->msm_lanes_per_term = w->ring_lpt[slot];
    tm->pairing_lanes_per_proof = w->ring_pair[slot];
    float first_start = 0, last_end = 0;
    for (int k = 0; k < pipes; k++) {
        hipEvent_t *ev = w->ring[slot][k];
        HIPCHK(hipEventSynchronize(ev[6]));
        float a, b, c, e, t0 = 0, t1 = 0;
        HIPCHK(hipEventElapsedTime(&a, ev[0], ev[1]));
        HIPCHK(hipEventElapsedTime(&b, ev[2], ev[3]));
        if (getenv("H2V_TIMELINE")) {
            const int ne = w->ring_split[slot] ? 9 : 7;
            fprintf(stderr, "[h2v timeline]");
            for (int q = 0; q < ne; q++) { float t; HIPCHK(hipEventElapsedTime(&t, ev[0], ev[q])); fprintf(stderr, " e%d=%.3f", q, t); }
            fprintf(stderr, "\n");
        }
        if (w->ring_split[slot]) {   // two concurrent launches: report the longer one
            float b2;
            HIPCHK(hipEventElapsedTime(&b2, ev[7], ev[8]));
            if (b2 > b) b = b2;
        }
        if (w->ring_lpt[slot] == 3) {
            // split MSM: the ladder launch over the per-proof terms ([4] .. [11]) and, beside it on another stream, the
            // fixed-base launch over the VK-base terms ([9] .. [10]) - two kernels, two durations
            float fx;
            HIPCHK(hipEventElapsedTime(&c, ev[4], ev[11]));
            HIPCHK(hipEventElapsedTime(&fx, ev[9], ev[10]));
            tm->g1_msm_fixed_ms += fx;
            tm->msm_var_lanes_per_term = w->ring_var[slot];
        } else {
            HIPCHK(hipEventElapsedTime(&c, ev[4], ev[5]));
        }
        HIPCHK(hipEventElapsedTime(&e, ev[5], ev[6]));
        tm->transcript_combiner_ms += a; tm->g1_decompress_ms += b; tm->g1_msm_ms += c; tm->pairing_ms += e;
        if (k > 0) { HIPCHK(hipEventElapsedTime(&t0, w->ring[slot][0][2], ev[2])); }
        HIPCHK(hipEventElapsedTime(&t1, w->ring[slot][0][2], ev[6]));
        if (k == 0 || t0 < first_start) first_start = t0;
        if (t1 > last_end) last_end = t1;
    }
    tm->total_ms = last_end - first_start;
    return H2V_OK;
}

// The stream of the host-buffer entry points.  Not laned: it runs the kernels too - a pool stream (a hardware queue of its
// own).  Laned: it carries the uploads and the fork event only (and hs_down the downloads), the kernels run on the lanes'
// pool streams: plain non-blocking streams of the workspace, so that the copies never queue behind a lane's kernels - with
// eight two-stream lanes the pool's sixteen streams are all taken, and a pool stream shared with a lane held every upload
// back behind that lane's pairing launch (host path, 8 lanes: 877 k proofs/s; with its own copy streams: see DESIGN.md 6.1).
static int host_stream(h2v_workspace *ws) {
    if (!ws->hs) {
        if (ws->n_lanes) {
            HIPCHK(hipStreamCreateWithFlags(&ws->hs, hipStreamNonBlocking));
            ws->copy_streams_owned = true;
            if (!ws->hs_down) HIPCHK(hipStreamCreateWithFlags(&ws->hs_down, hipStreamNonBlocking));
        } else {
            HIPCHK(make_stream(&ws->hs));
        }
    }
    if (!ws->ev_host) HIPCHK(hipEventCreateWithFlags(&ws->ev_host, hipEventDisableTiming));
    return H2V_OK;
}
// bytes a host-buffer call's verdicts take in its pinned download block: accept[n], then the RLC verdict word at the next multiple of 8
static inline size_t accept_extent(uint64_t n) { return (size_t)((n + 7) & ~(uint64_t)7) + 8; }
// Packs the caller's host buffers into the pinned block and enqueues ONE upload on ws->hs.
static int stage_inputs(const h2v_plan *p, const h2v_batch *b, h2v_workspace *ws) {
    const uint64_t n = b->n;
    const uint64_t total = b->proof_off[n];
    for (uint64_t i = 0; i < n; i++)
        if (b->proof_off[i + 1] < b->proof_off[i]) return fail(H2V_E_ARG, "proof offsets must be non-decreasing");
    int rc = host_stream(ws);
    if (rc) return rc;
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_off = 0, o_inst = up16((n + 1) * 8), o_ci = o_inst + up16(n * p->d.n_pi * 32), o_proofs = o_ci + up16(n * p->d.n_ci * 48),
                 need = o_proofs + up16(total + 64);
    if (need > ws->in_block_cap) {
        HIPCHK(hipStreamSynchronize(ws->hs));
        if (ws->in_block) (void)hipFree(ws->in_block);
        if (ws->h_block) (void)hipHostFree(ws->h_block);
        ws->in_block = nullptr; ws->h_block = nullptr; ws->in_block_cap = 0;
        const size_t cap = need + need / 4;
        if (hipMalloc((void **)&ws->in_block, cap) != hipSuccess || hipHostMalloc((void **)&ws->h_block, cap, hipHostMallocDefault) != hipSuccess)
            return fail(H2V_E_DEVICE, "staging allocation failed");
        ws->in_block_cap = cap;
    }
    // (accept bytes, then - RLC mode - the verdict word at the next multiple of 8: the capacity is compared with THAT extent;
    //  comparing it with n let a batch within 8 proofs of an earlier, smaller batch's slack put the word past the end of the
    //  pinned block: hipMemcpyAsync "invalid argument", found by tools/soak.py)
    if (accept_extent(n) > ws->h_accept_cap) {
        if (ws->h_accept) (void)hipHostFree(ws->h_accept);
        ws->h_accept = nullptr; ws->h_accept_cap = 0;
        const size_t cap = accept_extent(n) + n / 4 + 64;
        if (hipHostMalloc((void **)&ws->h_accept, cap, hipHostMallocDefault) != hipSuccess) return fail(H2V_E_DEVICE, "staging allocation failed");
        ws->h_accept_cap = cap;
    }
    memcpy(ws->h_block + o_off, b->proof_off, (n + 1) * 8);
    if (p->d.n_pi) memcpy(ws->h_block + o_inst, b->instances, n * p->d.n_pi * 32);
    if (p->d.n_ci) memcpy(ws->h_block + o_ci, b->committed, n * 48);
    memcpy(ws->h_block + o_proofs, b->proofs, total);
    memset(ws->h_block + o_proofs + total, 0, 64);
    HIPCHK(hipMemcpyAsync(ws->in_block, ws->h_block, need, hipMemcpyHostToDevice, ws->hs));
    ws->in_off = (uint64_t *)(ws->in_block + o_off);
    ws->in_inst = ws->in_block + o_inst;
    ws->in_ci = ws->in_block + o_ci;
    ws->in_proofs = ws->in_block + o_proofs;
    return H2V_OK;
}
static int run_rlc(const h2v_plan *p, uint32_t n, const uint8_t *proofs, const uint64_t *off, const uint8_t *inst, const uint8_t *ci,
                   uint8_t *accept, uint32_t *status_out, h2v_workspace *w, hipStream_t st, const uint32_t seed[8], bool one_stream_opt);
static bool rlc_supported(const h2v_plan *p);
static int rlc_seed(const h2v_rlc_opts *o, uint32_t seed[8]);
static int run_rlc_or_routed(const h2v_plan *p, uint32_t n, const uint8_t *proofs, const uint64_t *off, const uint8_t *inst, const uint8_t *ci,
                             uint8_t *accept, uint32_t *status_out, h2v_workspace *w, hipStream_t st, const uint32_t seed[8], bool one_stream_opt);

extern "C" int h2v_workspace_rlc_result(h2v_workspace *w, uint32_t calls_back, uint32_t *batch_accepted, h2v_rlc_timings *tm);
// Host-buffer batches on a LANED workspace: as many in flight as the mode has lanes, on ONE workspace.  Every batch gets a
// staging slot of its own (pinned block + device block + accept buffers); uploads run in order on `hs`, the chunks on
// the lanes, and the downloads on `hs_down`, each behind the lanes its batch ran on.  h2v_verify_batch_wait collects the
// OLDEST batch.
static int submit_laned(const h2v_plan *p, const h2v_batch *b, h2v_workspace *ws, bool rlc, const uint32_t *seed) {
    const uint32_t depth = laned_depth(ws, b->n, rlc, nullptr);
    if (ws->h_head - ws->h_tail >= depth || ws->h_head - ws->h_tail >= (uint64_t)h2v_workspace::MAXH)
        return fail(H2V_E_ARG, "as many batches in flight as this workspace has lanes: call h2v_verify_batch_wait first");
    if (!ws->hs_down) HIPCHK(make_stream(&ws->hs_down));
    h2v_workspace::HostSlot &sl = ws->hslot[ws->h_head % h2v_workspace::MAXH];
    if (!sl.ev) HIPCHK(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    const uint64_t n = b->n;
    sl.n = n; sl.rlc = rlc;
    if (n) {
        const uint64_t total = b->proof_off[n];
        for (uint64_t i = 0; i < n; i++)
            if (b->proof_off[i + 1] < b->proof_off[i]) return fail(H2V_E_ARG, "proof offsets must be non-decreasing");
        auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
        const size_t o_inst = up16((n + 1) * 8), o_ci = o_inst + up16(n * p->d.n_pi * 32), o_proofs = o_ci + up16(n * p->d.n_ci * 48),
                     need = o_proofs + up16(total + 64);
        if (need > sl.in_cap) {     // (the slot is free: its previous batch has been collected)
            if (sl.in_block) (void)hipFree(sl.in_block);
            if (sl.h_block) (void)hipHostFree(sl.h_block);
            sl.in_block = nullptr; sl.h_block = nullptr; sl.in_cap = 0;
            const size_t cap = need + need / 4;
            if (hipMalloc((void **)&sl.in_block, cap) != hipSuccess || hipHostMalloc((void **)&sl.h_block, cap, hipHostMallocDefault) != hipSuccess)
                return fail(H2V_E_DEVICE, "staging allocation failed");
            sl.in_cap = cap;
        }
        if (accept_extent(n) > sl.acc_cap) {
            if (sl.h_accept) (void)hipHostFree(sl.h_accept);
            if (sl.d_accept) (void)hipFree(sl.d_accept);
            sl.h_accept = nullptr; sl.d_accept = nullptr; sl.acc_cap = 0;
            const size_t cap = accept_extent(n) + n / 4 + 64;
            if (hipHostMalloc((void **)&sl.h_accept, cap, hipHostMallocDefault) != hipSuccess || hipMalloc((void **)&sl.d_accept, cap) != hipSuccess)
                return fail(H2V_E_DEVICE, "staging allocation failed");
            sl.acc_cap = cap;
        }
        memcpy(sl.h_block, b->proof_off, (n + 1) * 8);
        if (p->d.n_pi) memcpy(sl.h_block + o_inst, b->instances, n * p->d.n_pi * 32);
        if (p->d.n_ci) memcpy(sl.h_block + o_ci, b->committed, n * 48);
        memcpy(sl.h_block + o_proofs, b->proofs, total);
        memset(sl.h_block + o_proofs + total, 0, 64);
        HIPCHK(hipMemcpyAsync(sl.in_block, sl.h_block, need, hipMemcpyHostToDevice, ws->hs));
        int rc = run_laned(p, (uint32_t)n, sl.in_block + o_proofs, (const uint64_t *)sl.in_block, sl.in_block + o_inst, sl.in_block + o_ci, sl.d_accept, nullptr,
                           ws, ws->hs, rlc, seed, false, true);
        if (rc == H2V_OK) {
            // the download waits for the lanes this call's chunks ran on (their events as recorded just now)
            const int slot = (int)((ws->calls - 1) % h2v_workspace::RING);
            sl.call = ws->calls - 1;
            const uint32_t nch = ws->lring_chunks[slot], mod = ws->lring_mod[slot], first = ws->lring_first[slot];
            for (uint32_t c = 0; c < nch && c < mod; c++)
                if (hipStreamWaitEvent(ws->hs_down, ws->lane_ev[(first + c) % mod], 0) != hipSuccess) rc = fail(H2V_E_DEVICE, "stream wait failed");
            if (rc == H2V_OK && hipMemcpyAsync(sl.h_accept, sl.d_accept, n, hipMemcpyDeviceToHost, ws->hs_down) != hipSuccess) rc = fail(H2V_E_DEVICE, "download of accept[] failed");
            // (RLC: how many of the call's batch checks failed comes back the same way - a synchronous copy in _wait would go
            //  through the legacy NULL stream, which every pool stream is ordered with: it would drain all batches in flight)
            if (rc == H2V_OK && rlc && hipMemcpyAsync(sl.h_accept + ((n + 7) & ~(uint64_t)7), ws->rlc_fail + slot, 4, hipMemcpyDeviceToHost, ws->hs_down) != hipSuccess)
                rc = fail(H2V_E_DEVICE, "download of the batch verdict failed");
        }
        if (rc) {
            const std::string e = g_err;
            (void)hipStreamSynchronize(ws->hs);
            for (uint32_t l = 0; l < ws->n_lanes; l++) if (ws->lane_st[l]) (void)hipStreamSynchronize(ws->lane_st[l]);
            (void)hipStreamSynchronize(ws->hs_down);
            return fail(rc, e);
        }
    }
    HIPCHK(hipEventRecord(sl.ev, ws->hs_down));
    ws->h_head++;
    ws->pending = true;
    return H2V_OK;
}
static int wait_laned(h2v_workspace *ws, uint8_t *accept, int *fell_back) {
    if (ws->h_head == ws->h_tail) return fail(H2V_E_ARG, "no batch in flight on this workspace");
    h2v_workspace::HostSlot &sl = ws->hslot[ws->h_tail % h2v_workspace::MAXH];
    HIPCHK(hipEventSynchronize(sl.ev));
    if (sl.n) memcpy(accept, sl.h_accept, sl.n);
    if (fell_back) {
        *fell_back = 0;
        if (sl.rlc && sl.n) {
            uint32_t failed = 0;
            memcpy(&failed, sl.h_accept + ((sl.n + 7) & ~(uint64_t)7), 4);
            *fell_back = failed ? 1 : 0;
        }
    }
    ws->h_tail++;
    ws->pending = ws->h_head != ws->h_tail;
    return H2V_OK;
}

// Asynchronous host-buffer form: packs + uploads the batch, enqueues the verification (per-proof, or the RLC batch mode
// with H2V_SUBMIT_RLC) and the download of accept[] on the workspace's stream, and returns.  The caller's buffers may be
// reused as soon as this returns (they were copied into pinned memory).  One batch at a time per workspace: alternate two
// workspaces to overlap the upload of batch k+1 with the kernels of batch k.
extern "C" int h2v_verify_batch_submit(const h2v_plan *p, const h2v_batch *b, h2v_workspace *ws, uint32_t flags, const h2v_rlc_opts *opts) {
    if (!p || !b || !ws) return fail(H2V_E_ARG, "null argument");
    ALIVE(p); ALIVE(ws);
    if (ws->pending && !ws->n_lanes) return fail(H2V_E_ARG, "the workspace already has a batch in flight: call h2v_verify_batch_wait first");
    if (b->n && (!b->proofs || !b->proof_off)) return fail(H2V_E_ARG, "null proofs / offsets");
    if (b->n && p->d.n_pi && !b->instances) return fail(H2V_E_ARG, "plan has public inputs but instances == NULL");
    if (b->n && p->d.n_ci && !b->committed) return fail(H2V_E_ARG, "plan has a committed instance but committed == NULL");
    HIPCHK(hipSetDevice(p->device));
    int rc = host_stream(ws);
    if (rc) return rc;
    if (ws->n_lanes) {
        if (b->n && (rc = ws_fits(ws, p, b->n, false))) return rc;
        const bool rlc_l = (flags & H2V_SUBMIT_RLC) && rlc_supported(p);
        uint32_t seed_l[8] = {};
        if (rlc_l && (rc = rlc_seed(opts, seed_l))) return rc;
        return submit_laned(p, b, ws, rlc_l, seed_l);
    }
    ws->pending_n = b->n;
    ws->pending_rlc = false;
    if (b->n) {
        if ((rc = ws_fits(ws, p, b->n, false))) return rc;
        if ((rc = stage_inputs(p, b, ws))) return rc;
        const bool rlc = (flags & H2V_SUBMIT_RLC) && rlc_supported(p);
        uint32_t seed[8] = {};
        if (rlc && (rc = rlc_seed(opts, seed))) { (void)hipStreamSynchronize(ws->hs); return rc; }
        if (ws->n_lanes) {
            rc = run_laned(p, (uint32_t)b->n, ws->in_proofs, ws->in_off, ws->in_inst, ws->in_ci, ws->accept, nullptr, ws, ws->hs, rlc, seed, true);
        } else if (rlc) {
            rc = run_rlc_or_routed(p, (uint32_t)b->n, ws->in_proofs, ws->in_off, ws->in_inst, ws->in_ci, ws->accept, nullptr, ws, ws->hs, seed,
                                   opts && (opts->flags & H2V_RLC_ONE_STREAM));
        } else {
            rc = run_pipeline(p->d, (uint32_t)b->n, ws->in_proofs, ws->in_off, ws->in_inst, ws->in_ci, ws->accept, nullptr, ws, ws->hs, nullptr, false);
        }
        ws->pending_rlc = rlc;
        if (rc == H2V_OK && hipMemcpyAsync(ws->h_accept, ws->accept, b->n, hipMemcpyDeviceToHost, ws->hs) != hipSuccess) rc = fail(H2V_E_DEVICE, "download of accept[] failed");
        if (rc == H2V_OK && rlc && ws->rlc && hipMemcpyAsync(ws->h_accept + ((b->n + 7) & ~(uint64_t)7), rlc_flags_of(ws), 4, hipMemcpyDeviceToHost, ws->hs) != hipSuccess)
            rc = fail(H2V_E_DEVICE, "download of the batch verdict failed");
        if (rc) {
            // the upload from the pinned block and some kernels may already be enqueued: nothing of this workspace may be
            // reused before they have drained
            const std::string e = g_err;
            (void)hipStreamSynchronize(ws->hs);
            for (uint32_t l = 0; l < ws->n_lanes; l++) if (ws->lane_st[l]) (void)hipStreamSynchronize(ws->lane_st[l]);
            return fail(rc, e);
        }
    }
    HIPCHK(hipEventRecord(ws->ev_host, ws->hs));
    ws->pending = true;
    return H2V_OK;
}
// Waits for the batch submitted on `ws`, copies its accept bytes out.  fell_back (optional): 1 when the batch was submitted
// in RLC mode and the batch check failed, so that the per-proof kernels produced accept[].
extern "C" int h2v_verify_batch_wait(h2v_workspace *ws, uint8_t *accept, int *fell_back) {
    if (!ws || !accept) return fail(H2V_E_ARG, "null argument");
    ALIVE(ws);
    if (ws->n_lanes) { HIPCHK(hipSetDevice(ws->device)); return wait_laned(ws, accept, fell_back); }
    if (!ws->pending) return fail(H2V_E_ARG, "no batch in flight on this workspace");
    HIPCHK(hipSetDevice(ws->device));
    ws->pending = false;
    HIPCHK(hipEventSynchronize(ws->ev_host));
    if (ws->pending_n) memcpy(accept, ws->h_accept, ws->pending_n);
    if (fell_back) {
        *fell_back = 0;
        if (ws->pending_rlc && ws->pending_n) {     // the batch verdict came back with the accept bytes (1 = passed)
            uint32_t passed = 1;
            memcpy(&passed, ws->h_accept + ((ws->pending_n + 7) & ~(uint64_t)7), 4);
            *fell_back = passed ? 0 : 1;
        }
    }
    return H2V_OK;
}
static int verify_host(const h2v_plan *p, const h2v_batch *b, uint8_t *accept, h2v_workspace *ws, uint32_t flags, const h2v_rlc_opts *opts, int *fell_back) {
    if (!p || !b || !accept) return fail(H2V_E_ARG, "null argument");
    if (fell_back) *fell_back = 0;
    if (b->n == 0) return H2V_OK;
    ALIVE(p); ALIVE(ws);
    HIPCHK(hipSetDevice(p->device));
    h2v_workspace *tmp = nullptr;
    int rc;
    if (!ws) {
        if ((rc = ws_create_for(p->d, p->device, b->n, false, &tmp))) return rc;
        ws = tmp;
    }
    rc = h2v_verify_batch_submit(p, b, ws, flags, opts);
    if (rc == H2V_OK) rc = h2v_verify_batch_wait(ws, accept, fell_back);
    if (tmp) h2v_workspace_free(tmp);
    return rc;
}
extern "C" int h2v_verify_batch(const h2v_plan *p, const h2v_batch *b, uint8_t *accept, h2v_workspace *ws) {
    return verify_host(p, b, accept, ws, 0, nullptr, nullptr);
}


// ---------------------------------------------------------------------------------------------- bucket MSM (Pippenger)
// Device buffers of one bucket MSM over at most cap_n terms (h2v_pippenger.hpp).
struct PipWs {
    uint32_t cap_n = 0, cap_halves = 0;
    uint32_t *pts28 = nullptr, *cnt = nullptr, *off = nullptr, *order = nullptr, *cls = nullptr, *list = nullptr, *partial = nullptr, *wsum = nullptr;
    int16_t *dig = nullptr;
};
static void pip_free(PipWs &w) {
    void *ptrs[] = {w.pts28, w.cnt, w.off, w.order, w.cls, w.list, w.partial, w.wsum, w.dig};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    w = PipWs();
}
static int pip_alloc(PipWs &w, uint32_t cap_n, uint32_t halves) {
    const size_t nbmax = (size_t)PIP_MAX_W * (1u << (PIP_MAX_C - 1));
    const size_t ent = (size_t)cap_n * halves * PIP_MAX_W;
    w.cap_n = cap_n; w.cap_halves = halves;
    bool ok = hipMalloc((void **)&w.pts28, (size_t)(cap_n ? cap_n : 1) * PIP_PT_DW * 4) == hipSuccess &&
              hipMalloc((void **)&w.dig, (ent ? ent : 1) * 2) == hipSuccess && hipMalloc((void **)&w.list, (ent ? ent : 1) * 4) == hipSuccess &&
              hipMalloc((void **)&w.cnt, nbmax * 4) == hipSuccess && hipMalloc((void **)&w.off, (nbmax + 1) * 4) == hipSuccess &&
              hipMalloc((void **)&w.order, nbmax * 4) == hipSuccess && hipMalloc((void **)&w.cls, PIP_CLS_DW * 4) == hipSuccess && hipMalloc((void **)&w.partial, nbmax * PIP_PART_DW * 4) == hipSuccess &&
              hipMalloc((void **)&w.wsum, (size_t)PIP_MAX_W * PIP_PART_DW * 4) == hipSuccess;
    if (!ok) { pip_free(w); return fail(H2V_E_DEVICE, "hipMalloc(bucket MSM workspace) failed"); }
    return H2V_OK;
}
// Window width for n terms.  Larger windows mean fewer windows (W = floor(128 / c) + 1 per GLV half) but 2^(c-1) buckets
// per window, capped at 512 so that one block's LDS holds a window in k_pip_reduce; at least ~24 entries per bucket on
// average.  chain = the most entries one lane of k_pip_accumulate sums (a bucket of more gets 2, 4, ... 256 lanes): short
// chains mean more lanes and deeper trees of complete additions.  H2V_OPT_RLC_WINDOW_BITS / _CHAIN force a shape (tests, tuning).
static void pip_shape(PipArgs &a) {
    const int env_c = g_opts.v[H2V_OPT_RLC_WINDOW_BITS], env_chain = g_opts.v[H2V_OPT_RLC_CHAIN];
    const double entries_per_window = (double)a.n * a.halves;
    uint32_t c = PIP_MAX_C;
    while (c > 4 && entries_per_window / (double)(1u << (c - 1)) < 24.0) c--;
    if (env_c >= 3 && env_c <= PIP_MAX_C) c = (uint32_t)env_c;
    a.c = c; a.NB = 1u << (c - 1); a.W = 128 / c + 1;
    while (a.W > PIP_MAX_W) { a.c++; a.NB <<= 1; a.W = 128 / a.c + 1; }
    a.chain = env_chain >= 2 && env_chain <= 1024 ? (uint32_t)env_chain : 20u;
}
static void pip_bind(const PipWs &w, PipArgs &a) {
    pip_shape(a);
    a.pts28 = w.pts28; a.dig = w.dig; a.cnt = w.cnt; a.off = w.off; a.order = w.order; a.cls = w.cls; a.list = w.list; a.partial = w.partial; a.wsum = w.wsum;
}
// Enqueues the six kernels of one or two bucket MSMs (np problems side by side in every launch) on `st`.
// a[]: n, halves, scal, pidx, pool0 / n_pool0 / pool1, out filled in.  ev (optional): 4 events recorded at the start,
// before and after k_pip_accumulate, and at the end.
static int pip_launch(const PipWs *const *ws, PipArgs *a, uint32_t np, hipStream_t st, hipEvent_t *ev) {
    PipArgs2 a2 = {};
    uint32_t max_n = 0, max_W = 0, max_NB = 0, max_acc_blocks = 0;
    for (uint32_t q = 0; q < np; q++) {
        if (a[q].n > ws[q]->cap_n || a[q].halves > ws[q]->cap_halves) return fail(H2V_E_ARG, "bucket MSM workspace too small");
        pip_bind(*ws[q], a[q]);
        a2.p[q] = a[q];
        const uint32_t nb = a[q].W * a[q].NB;
        // lanes of k_pip_accumulate: a bucket of count cnt > T gets fewer than 2 cnt / T lanes, any other non-empty one 1; each
        // class is padded to whole blocks
        const uint64_t lanes = 2ull * a[q].n * a[q].halves * a[q].W / a[q].chain + nb + 256ull * PIP_N_CLASSES;
        max_n = a[q].n > max_n ? a[q].n : max_n; max_W = a[q].W > max_W ? a[q].W : max_W; max_NB = a[q].NB > max_NB ? a[q].NB : max_NB;
        const uint32_t blocks = (uint32_t)((lanes + 255) / 256);
        max_acc_blocks = blocks > max_acc_blocks ? blocks : max_acc_blocks;
    }
    if (ev) HIPCHK(hipEventRecord(ev[0], st));
    for (uint32_t q = 0; q < np; q++) HIPCHK(hipMemsetAsync(a[q].cnt, 0, (size_t)a[q].W * a[q].NB * 4, st));
    if (max_n) hipLaunchKernelGGL(k_pip_digits, dim3((max_n + 255) / 256, np), dim3(256), 0, st, a2);
    hipLaunchKernelGGL(k_pip_scan, dim3(1, np), dim3(1024), 0, st, a2);
    if (max_n) hipLaunchKernelGGL(k_pip_scatter, dim3((max_n + 255) / 256, np), dim3(256), 0, st, a2);
    if (ev) HIPCHK(hipEventRecord(ev[1], st));
    hipLaunchKernelGGL(k_pip_accumulate, dim3(max_acc_blocks, np), dim3(256), 0, st, a2);
    if (ev) HIPCHK(hipEventRecord(ev[2], st));
    const size_t lds = (size_t)43 * max_NB * 4;
    HIPCHK(hipFuncSetAttribute((const void *)k_pip_reduce, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_pip_reduce, dim3(max_W, np), dim3(max_NB), lds, st, a2);
    hipLaunchKernelGGL(k_pip_combine, dim3(1, np), dim3(64), 0, st, a2);
    if (ev) HIPCHK(hipEventRecord(ev[3], st));
    HIPCHK(hipGetLastError());
    return H2V_OK;
}

// ---------------------------------------------------------------------------------------------- RLC batch mode
struct RlcWs {
    uint64_t cap = 0;
    uint32_t n_var = 0, n_fix = 0;
    uint32_t *r_scal = nullptr, *r_idx = nullptr, *l_scal = nullptr, *l_idx = nullptr, *vk_part = nullptr;
    uint8_t *good = nullptr;
    PipWs R, L;
    uint32_t *sums = nullptr;   // er (36 dwords) then el (36): the two MSM results
    uint32_t *misc = nullptr;   // [0..23] a dummy affine point, [24] status of the batch check, [26] its valid byte, [27] its accept byte
    uint32_t *flags = nullptr;  // [0] the batch check passed, [1 + g] group g (proofs 64 g ..) is final: cap / 64 + 2 dwords
    // fall-back stage 1 (group checks, created by the first call with at least GRP_MIN_N proofs): term lists, sums and
    // verdict buffers of the groups, the argument array of their 2 G bucket MSMs and ONE pool for those MSMs' buffers
    struct Grp {
        uint32_t n = 0, G = 0, stride = 0, max_n = 0, acc_blocks = 0;
        uint64_t key_gen = 0;          // the plan (load counter) the cached arguments were built for
        const uint32_t *key_pts = nullptr;
        uint32_t cap_stride = 0;
        std::vector<void *> staged;    // pinned host copies of argument arrays whose uploads may still be in flight
        uint32_t *g_scal = nullptr, *g_idx = nullptr, *er_g = nullptr, *el_g = nullptr, *pts_g = nullptr, *status_g = nullptr;
        uint8_t *valid_g = nullptr, *accept_g = nullptr, *pool = nullptr;
        PipArgs *args_d = nullptr;
        size_t cnt_bytes = 0, pool_bytes = 0;
        uint32_t cap_G = 0;
    } grp;
    static constexpr int NEV = 11;
    hipEvent_t ring[h2v_workspace::RING][NEV] = {};
    uint64_t calls = 0;
    uint64_t run_len = 0;       // calls since these buffers last became the workspace's current ones (rlc_ensure)
    uint32_t last_c = 0, last_W = 0, last_chain = 0, last_terms = 0;
    bool last_routed = false;   // the most recent call on this (ordinary) workspace went straight to the per-proof kernels
    uint8_t routed_ring[64] = {};
};
static void rlc_release(RlcWs *r) {
    void *ptrs[] = {r->r_scal, r->r_idx, r->l_scal, r->l_idx, r->vk_part, r->good, r->sums, r->misc, r->flags, r->grp.g_scal, r->grp.g_idx, r->grp.er_g,
                    r->grp.el_g, r->grp.pts_g, r->grp.status_g, r->grp.valid_g, r->grp.accept_g, r->grp.pool, r->grp.args_d};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    for (void *q : r->grp.staged) (void)hipHostFree(q);   // (the caller has drained the streams: ws_release)
    pip_free(r->R); pip_free(r->L);
    for (auto &set : r->ring) for (hipEvent_t e : set) if (e) (void)hipEventDestroy(e);
    delete r;
}
// One set of buffers per plan SHAPE (n_var, n_fix) the workspace has served: a workspace that alternates between plans (ws_fits
// allows any plan that fits; h2v_workspace_create_multi is made for it) swaps between parked sets instead of freeing and
// re-allocating at every switch - hipFree waits for the whole device.  At most eight shapes are kept.
static int rlc_ensure(h2v_workspace *w, const h2v_plan *p) {
    if (w->rlc && w->rlc->n_var == p->n_var && w->rlc->n_fix == p->n_fix) return H2V_OK;
    for (size_t k = 0; k < w->rlc_parked.size(); k++) {
        RlcWs *q = w->rlc_parked[k];
        if (q->n_var == p->n_var && q->n_fix == p->n_fix) {
            w->rlc_parked[k] = w->rlc;
            if (!w->rlc_parked[k]) w->rlc_parked.erase(w->rlc_parked.begin() + (long)k);
            w->rlc = q;
            q->run_len = 0;
            return H2V_OK;
        }
    }
    if (w->rlc) {
        if (w->rlc_parked.size() >= 8) { rlc_release(w->rlc_parked.front()); w->rlc_parked.erase(w->rlc_parked.begin()); }
        w->rlc_parked.push_back(w->rlc);
        w->rlc = nullptr;
    }
    RlcWs *r = new RlcWs();
    r->cap = w->cap; r->n_var = p->n_var; r->n_fix = p->n_fix;
    const size_t nr = (size_t)w->cap * p->n_var + p->n_fix, blocks = (w->cap + 63) / 64;
    bool ok = hipMalloc((void **)&r->r_scal, nr * 32) == hipSuccess && hipMalloc((void **)&r->r_idx, nr * 4) == hipSuccess &&
              hipMalloc((void **)&r->l_scal, (size_t)w->cap * 32) == hipSuccess && hipMalloc((void **)&r->l_idx, (size_t)w->cap * 4) == hipSuccess &&
              hipMalloc((void **)&r->vk_part, blocks * (p->n_fix ? p->n_fix : 1) * 32) == hipSuccess && hipMalloc((void **)&r->good, w->cap) == hipSuccess &&
              hipMalloc((void **)&r->sums, 72 * 4) == hipSuccess && hipMalloc((void **)&r->misc, 32 * 4) == hipSuccess &&
              hipMemset(r->misc, 0, 32 * 4) == hipSuccess && hipMalloc((void **)&r->flags, (blocks + 2) * 4) == hipSuccess &&
              hipMemset(r->flags, 0, (blocks + 2) * 4) == hipSuccess;
    for (auto &set : r->ring) for (hipEvent_t &e : set) if (ok) ok = hipEventCreate(&e) == hipSuccess;
    if (ok) ok = pip_alloc(r->R, (uint32_t)nr, 2) == H2V_OK && pip_alloc(r->L, (uint32_t)w->cap, 1) == H2V_OK;
    if (!ok) { rlc_release(r); return fail(H2V_E_DEVICE, "RLC workspace allocation failed"); }
    w->rlc = r;
    return H2V_OK;
}
static uint64_t rlc_calls_of(const h2v_workspace *w) { return w->rlc ? w->rlc->calls : 0; }
static const uint32_t *rlc_flags_of(const h2v_workspace *w) { return w->rlc->flags; }
static bool rlc_supported(const h2v_plan *p) { return !p->d.ivc && p->n_var > 0 && p->n_fix > 0 && p->n_var + p->n_fix == p->d.n_terms; }

// Fall-back stage 1: group checks (h2v_rlc.hpp: k_rlc_group_terms, k_pairing_rlc_groups).  Groups of 64 proofs; every group
// is two small bucket MSMs (right: 64 n_var + n_fix terms, 255-bit scalars; left: 64 terms, 128-bit) with 7-bit windows
// (64 buckets x 19 windows: ~20 entries per bucket on the right).  Batches below GRP_MIN_N proofs or above GRP_MAX_G groups
// go straight to the per-proof kernels.  Everything here is enqueued behind the batch check and returns at once when it
// passed.  H2V_OPT_RLC_GROUP_STAGE = -1 switches the stage off (measurements).  (Window width 5 / 6 / 7 / 8 bits, one rejecting proof per
// 4096-proof batch, sixteen batches in flight: 2.90 / 2.72 / 2.86 / 2.80 ms per batch - the three pairing stages in a row are
// what the fall-back waits for, not these sums.)
#define GRP_MIN_N 256u
#define GRP_MAX_G 512u
#define GRP_C 7u
static bool rlc_groups_on(uint32_t n) {
    return g_opts.v[H2V_OPT_RLC_GROUP_STAGE] >= 0 && n >= GRP_MIN_N && (n + 63) / 64 <= GRP_MAX_G;
}
// The argument array of the group stage is cached per (batch size, plan, point buffer).  When it has to change, the new one
// is uploaded ON THE CALL'S STREAM from a pinned host copy: the kernels of an earlier call on that stream read args_d when
// they run, and a synchronous hipMemcpy (through the NULL stream, not ordered with a non-blocking caller stream) could
// hand them the new batch size's pointers.  The key is the plan's load counter, not its address.
static int rlc_groups_ensure(RlcWs *r, const h2v_plan *p, h2v_workspace *w, uint32_t n, hipStream_t st) {
    RlcWs::Grp &g = r->grp;
    const uint32_t G = (n + 63) / 64, slots = H2V_SLOTS(p->d);
    const uint32_t W = 128 / GRP_C + 1, NB = 1u << (GRP_C - 1), nb = W * NB, stride = 64 * p->n_var + p->n_fix;
    if (g.n == n && g.key_gen == p->gen && g.key_pts == w->pts && g.stride == stride) return H2V_OK;
    if (G > g.cap_G || stride > g.cap_stride) {
        HIPCHK(hipStreamSynchronize(st));     // (the buffers about to be freed may be in use by the previous call)
        void *old[] = {g.g_scal, g.g_idx, g.er_g, g.el_g, g.pts_g, g.status_g, g.valid_g, g.accept_g, g.pool, g.args_d};
        for (void *q : old) if (q) (void)hipFree(q);
        for (void *q : g.staged) (void)hipHostFree(q);
        g = RlcWs::Grp();
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t per_r = up((size_t)stride * PIP_PT_DW * 4) + up((size_t)stride * 2 * W * 2) + up((size_t)stride * 2 * W * 4),
                     per_l = up((size_t)64 * PIP_PT_DW * 4) + up((size_t)64 * W * 2) + up((size_t)64 * W * 4),
                     per_any = up((size_t)(nb + 1) * 4) + up((size_t)nb * 4) + up(PIP_CLS_DW * 4) + up((size_t)nb * PIP_PART_DW * 4) + up((size_t)W * PIP_PART_DW * 4);
        g.cnt_bytes = (size_t)2 * G * up((size_t)nb * 4);
        g.pool_bytes = g.cnt_bytes + (size_t)G * (per_r + per_l + 2 * per_any);
        bool ok = hipMalloc((void **)&g.g_scal, (size_t)G * stride * 32) == hipSuccess && hipMalloc((void **)&g.g_idx, (size_t)G * stride * 4) == hipSuccess &&
                  hipMalloc((void **)&g.er_g, (size_t)G * 144) == hipSuccess && hipMalloc((void **)&g.el_g, (size_t)G * 144) == hipSuccess &&
                  hipMalloc((void **)&g.pts_g, (size_t)G * 96) == hipSuccess && hipMemset(g.pts_g, 0, (size_t)G * 96) == hipSuccess &&
                  hipMalloc((void **)&g.status_g, (size_t)G * 4) == hipSuccess && hipMalloc((void **)&g.valid_g, G) == hipSuccess &&
                  hipMemset(g.valid_g, 1, G) == hipSuccess && hipMalloc((void **)&g.accept_g, G) == hipSuccess &&
                  hipMalloc((void **)&g.pool, g.pool_bytes) == hipSuccess && hipMalloc((void **)&g.args_d, (size_t)2 * G * sizeof(PipArgs)) == hipSuccess;
        if (!ok) return fail(H2V_E_DEVICE, "hipMalloc(RLC group stage) failed");
        g.cap_G = G; g.cap_stride = stride;
    }
    // the argument array: problem 2 q = the right-hand sum of group q, 2 q + 1 its left-hand sum
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    std::vector<PipArgs> args(2 * (size_t)G);
    uint8_t *cur = g.pool + g.cnt_bytes;
    auto take = [&](size_t bytes) { uint8_t *q = cur; cur += up(bytes); return q; };
    uint32_t max_n = 0, acc_blocks = 1;
    for (uint32_t q = 0; q < G; q++) {
        const uint32_t ng = n - q * 64 < 64 ? n - q * 64 : 64;
        for (int side = 0; side < 2; side++) {
            PipArgs a = {};
            a.c = GRP_C; a.W = W; a.NB = NB; a.chain = 20;
            a.pool0 = w->pts; a.n_pool0 = n * slots;
            if (side == 0) {
                a.n = ng * p->n_var + p->n_fix; a.halves = 2;
                a.scal = g.g_scal + (size_t)q * stride * 8; a.pidx = g.g_idx + (size_t)q * stride; a.pool1 = p->d.vk_bases; a.out = g.er_g + (size_t)q * 36;
            } else {
                a.n = ng; a.halves = 1;
                a.scal = r->l_scal + (size_t)q * 64 * 8; a.pidx = r->l_idx + (size_t)q * 64; a.pool1 = nullptr; a.out = g.el_g + (size_t)q * 36;
            }
            const uint32_t cap_n = side == 0 ? stride : 64;
            a.cnt = (uint32_t *)(g.pool + (size_t)(2 * q + side) * up((size_t)nb * 4));
            a.pts28 = (uint32_t *)take((size_t)cap_n * PIP_PT_DW * 4);
            a.dig = (int16_t *)take((size_t)cap_n * a.halves * W * 2);
            a.list = (uint32_t *)take((size_t)cap_n * a.halves * W * 4);
            a.off = (uint32_t *)take((size_t)(nb + 1) * 4);
            a.order = (uint32_t *)take((size_t)nb * 4);
            a.cls = (uint32_t *)take(PIP_CLS_DW * 4);
            a.partial = (uint32_t *)take((size_t)nb * PIP_PART_DW * 4);
            a.wsum = (uint32_t *)take((size_t)W * PIP_PART_DW * 4);
            args[2 * q + side] = a;
            max_n = a.n > max_n ? a.n : max_n;
            const uint64_t lanes = 2ull * a.n * a.halves * W / a.chain + nb + 256ull * PIP_N_CLASSES;
            const uint32_t blocks = (uint32_t)((lanes + 255) / 256);
            acc_blocks = blocks > acc_blocks ? blocks : acc_blocks;
        }
    }
    if ((size_t)(cur - g.pool) > g.pool_bytes) return fail(H2V_E_DEVICE, "RLC group pool overrun");
    if (g.staged.size() >= 8) {               // (a workspace whose batch size keeps alternating: drain, then reuse)
        HIPCHK(hipStreamSynchronize(st));
        for (void *q : g.staged) (void)hipHostFree(q);
        g.staged.clear();
    }
    void *pinned = nullptr;
    if (hipHostMalloc(&pinned, args.size() * sizeof(PipArgs), hipHostMallocDefault) != hipSuccess) return fail(H2V_E_DEVICE, "staging allocation failed");
    g.staged.push_back(pinned);
    memcpy(pinned, args.data(), args.size() * sizeof(PipArgs));
    HIPCHK(hipMemcpyAsync(g.args_d, pinned, args.size() * sizeof(PipArgs), hipMemcpyHostToDevice, st));   // behind the earlier call's kernels
    g.n = n; g.G = G; g.stride = stride; g.max_n = max_n; g.acc_blocks = acc_blocks; g.key_gen = p->gen; g.key_pts = w->pts;
    return H2V_OK;
}
static int rlc_groups_launch(RlcWs *r, const h2v_plan *p, h2v_workspace *w, uint32_t n, uint8_t *accept, const H2vDevPlan &d1, hipStream_t st) {
    int rc = rlc_groups_ensure(r, p, w, n, st);
    if (rc) return rc;
    RlcWs::Grp &g = r->grp;
    const uint32_t G = g.G;
    RlcGroupArgs ga = {n, p->n_var, p->n_fix, (uint32_t)H2V_SLOTS(p->d), g.stride, p->d.terms, r->r_scal, r->r_idx, r->vk_part, g.g_scal, g.g_idx};
    hipLaunchKernelGGL(k_rlc_group_terms, dim3(G), dim3(64), 0, st, ga, r->flags);
    HIPCHK(hipMemsetAsync(g.pool, 0, g.cnt_bytes, st));
    HIPCHK(hipMemsetAsync(g.status_g, 0, (size_t)G * 4, st));
    hipLaunchKernelGGL(k_pip_digits_many, dim3((g.max_n + 255) / 256, 2 * G), dim3(256), 0, st, g.args_d, r->flags);
    hipLaunchKernelGGL(k_pip_scan_many, dim3(1, 2 * G), dim3(1024), 0, st, g.args_d, r->flags);
    hipLaunchKernelGGL(k_pip_scatter_many, dim3((g.max_n + 255) / 256, 2 * G), dim3(256), 0, st, g.args_d, r->flags);
    hipLaunchKernelGGL(k_pip_accumulate_many, dim3(g.acc_blocks, 2 * G), dim3(256), 0, st, g.args_d, r->flags);
    const uint32_t Wg = 128 / GRP_C + 1;
    hipLaunchKernelGGL(k_pip_wsum_many, dim3((2 * G * Wg * 4 + 63) / 64), dim3(64), 0, st, g.args_d, 2 * G, Wg, r->flags);
    hipLaunchKernelGGL(k_pip_horner_many, dim3(2 * G), dim3(64), 0, st, g.args_d, r->flags);
    hipLaunchKernelGGL(k_pairing_rlc_groups, dim3(G), dim3(64), COOP_LDS_BYTES(1), st, d1, g.pts_g, g.valid_g, g.er_g, g.el_g, g.status_g, g.accept_g, G, n,
                       r->good, accept, r->flags, w->rlc_stats_ptr ? w->rlc_stats_ptr : w->rlc_stats);
    HIPCHK(hipGetLastError());
    return H2V_OK;
}

// One batch in RLC mode.  Phase 1 as in run_pipeline (the decompression launch builds no window tables), then the batch
// check; the per-proof MSM + pairing kernels are queued behind it and return at once unless the batch check failed.
static int run_rlc(const h2v_plan *p, uint32_t n, const uint8_t *proofs, const uint64_t *off, const uint8_t *inst, const uint8_t *ci,
                   uint8_t *accept, uint32_t *status_out, h2v_workspace *w, hipStream_t st, const uint32_t seed[8], bool one_stream_opt) {
    const H2vDevPlan &d = p->d;
    int rc = rlc_ensure(w, p);
    if (rc) return rc;
    opts_from(w);
    struct Reset { ~Reset() { g_opts = LaunchOptions(); } } reset_opts;
    RlcWs *r = w->rlc;
    const uint32_t slots = H2V_SLOTS(d);
    hipEvent_t *ev = r->ring[r->calls % h2v_workspace::RING];
    r->calls++;
    r->run_len++;
    // two streams per batch: the caller's (transcript + combiner, then everything else) and one for the decompression
    // one_stream (h2v_rlc_opts.flags & H2V_RLC_ONE_STREAM, or H2V_OPT_STREAMS = 1): everything on
    // the caller's stream, decompression before the combiner - one stream per batch in flight instead of two
    const bool one_stream = w->one_stream_mode >= 0 ? w->one_stream_mode != 0 : (one_stream_opt || w->in_flight_hint >= 3);   // (H2V_OPT_STREAMS, the flag, or the caller's in-flight hint)
    if (!one_stream && (rc = ws_streams(w, 0, false, true, false))) return rc;
    hipStream_t pm = st, ps = one_stream ? st : w->pside[0];
    if (!one_stream) {
        HIPCHK(hipEventRecord(w->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(ps, w->ev_fork, 0));
    }
    // phase 1
    HIPCHK(hipEventRecord(ev[0], ps));
    {
        const uint32_t dec_grid = (n * slots + 63) / 64, units = 2 * dec_grid, max_blocks = (uint32_t)(msm_n_simd() / 4.0);
        uint32_t blocks = (units + 3) / 4;
        if (blocks > max_blocks) blocks = max_blocks;
        HIPCHK(hipMemsetAsync(w->dec_ctr, 0, 4, ps));
        hipLaunchKernelGGL(k_g1_decompress_queue, dim3(blocks), dim3(256), 0, ps, d, n, proofs, off, ci, inst, w->pts, w->valid, (uint32_t *)nullptr, w->valid_sub, w->dec_ctr, dec_grid);
    }
    HIPCHK(hipEventRecord(ev[1], ps));
    HIPCHK(hipEventRecord(w->ev_join[0], ps));
    HIPCHK(hipEventRecord(ev[2], pm));
    rc = launch_vm(d, n, w->stride, proofs, off, inst, ci, w->regs, w->scalars, w->status, nullptr, pm);
    if (rc) return rc;
    HIPCHK(hipEventRecord(ev[3], pm));
    if (!one_stream) HIPCHK(hipStreamWaitEvent(pm, w->ev_join[0], 0));
    HIPCHK(hipEventRecord(ev[10], pm));
    // the batch check
    RlcArgs ra = {n, p->n_var, p->n_fix, slots, d.pi_point, d.n_terms, d.terms, w->scalars, w->status, w->valid, w->valid_sub, {},
                  r->r_scal, r->r_idx, r->l_scal, r->l_idx, r->vk_part, r->good};
    for (int k = 0; k < 8; k++) ra.seed[k] = seed[k];
    const uint32_t blocks = (n + 63) / 64;
    hipLaunchKernelGGL(k_rlc_prepare, dim3(blocks), dim3(64), 0, pm, ra);
    hipLaunchKernelGGL(k_rlc_vk_sum, dim3((p->n_fix + 63) / 64), dim3(64), 0, pm, ra, blocks);
    HIPCHK(hipEventRecord(ev[4], pm));
    // both sums in the same launches: R = sum r_i er_i (255-bit scalars, GLV) and L = sum r_i pi_i (128-bit scalars)
    PipArgs pa[2] = {};
    pa[0].n = n * p->n_var + p->n_fix; pa[0].halves = 2; pa[0].scal = r->r_scal; pa[0].pidx = r->r_idx; pa[0].pool0 = w->pts; pa[0].n_pool0 = n * slots;
    pa[0].pool1 = d.vk_bases; pa[0].out = r->sums;
    pa[1].n = n; pa[1].halves = 1; pa[1].scal = r->l_scal; pa[1].pidx = r->l_idx; pa[1].pool0 = w->pts; pa[1].n_pool0 = n * slots; pa[1].pool1 = nullptr; pa[1].out = r->sums + 36;
    const PipWs *pws[2] = {&r->R, &r->L};
    if ((rc = pip_launch(pws, pa, 2, pm, ev + 5))) return rc;   // ev[5..8]
    r->last_c = pa[0].c; r->last_W = pa[0].W; r->last_chain = pa[0].chain; r->last_terms = pa[0].n;
    // one pairing check over a one-proof view of the plan: el = L, er = R (both Jacobian), no per-proof points
    H2vDevPlan d1 = d;
    d1.n_points = 1; d1.n_ci = 0; d1.ivc = 0; d1.pi_point = 0;
    uint32_t *st1 = r->misc + 24, *skip = r->flags;
    uint8_t *valid1 = (uint8_t *)(r->misc + 26), *acc1 = (uint8_t *)(r->misc + 27);
    HIPCHK(hipMemsetAsync(r->misc + 24, 0, 4, pm));            // status of the batch check
    HIPCHK(hipMemsetAsync(valid1, 1, 1, pm));
    const uint32_t n_groups = (n + 63) / 64;
    hipLaunchKernelGGL(k_pairing_rlc, dim3(1), dim3(64), COOP_LDS_BYTES(1), pm, d1, r->misc, valid1, r->sums, r->sums + 36, st1, acc1, n, r->good, accept, skip,
                       n_groups, w->rlc_fail_ptr, w->rlc_stats_ptr ? w->rlc_stats_ptr : w->rlc_stats, rlc_groups_on(n) ? 1u : 0u);
    HIPCHK(hipEventRecord(ev[9], pm));
    // fall-back, skipped on the device when the batch check passed.  Stage 1 finds the groups of 64 proofs that hold a failing
    // proof (everything else is final); stage 2 - window tables, per-proof MSM, per-proof pairing - decides inside those
    // groups.  Walking grids of one wave per SIMD: finding out that a launch is not needed costs a few microseconds, and a
    // launch that IS needed has the whole chip.
    if (rlc_groups_on(n) && (rc = rlc_groups_launch(r, p, w, n, accept, d1, pm))) return rc;
    const uint32_t cond_grid = (uint32_t)(msm_n_simd() / 4.0);
    {
        const uint32_t nb = (n * slots + 63) / 64;
        hipLaunchKernelGGL(k_build_tables, dim3(nb < 4 * cond_grid ? nb : 4 * cond_grid), dim3(64), 0, pm, n * slots, w->pts, w->valid, w->pt_tab, skip, slots);
    }
    {
        H2vMsmArgs ma = {d.terms, 0, d.n_main_terms, d.n_terms, 0, slots, {d.n_main_terms, d.n_main_terms, d.n_main_terms}, {w->er, nullptr, nullptr},
                         w->pt_tab, d.vk_tab, nullptr, 0, 0, skip};
        launch_msm_range(d, ma, n, w->scalars, w->pts, nullptr, pm);
    }
    {
        const uint32_t nb = (n + 1) / 2, grid = nb < 4 * cond_grid ? nb : 4 * cond_grid;
        hipLaunchKernelGGL(k_pairing_coop, dim3(grid), dim3(64), COOP_LDS_BYTES(COOP_GROUPS_PER_WAVE), pm, d, n, w->pts, w->valid, w->valid_sub, w->er, (const uint32_t *)nullptr, w->status, accept, (uint32_t *)nullptr, skip);
    }
    HIPCHK(hipGetLastError());
    if (status_out) HIPCHK(hipMemcpyAsync(status_out, w->status, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    return H2V_OK;
}
// The batch coefficients must be unpredictable to whoever made the proofs: 32 bytes from the OS per call (getrandom; fails
// closed).  A caller-given seed (H2V_RLC_SEED_GIVEN) is for TESTS and reproducible measurements only; even then every call
// gets different coefficients - the process-wide call counter is mixed in - so that a service that copied a fixed seed
// from a test or from bench.py does not hand provers a combination they can cancel against... if they can predict the
// counter: a given seed is not a production setting.
#include <sys/random.h>
#include <atomic>
static int rlc_seed(const h2v_rlc_opts *o, uint32_t seed[8]) {
    if (o && (o->flags & H2V_RLC_SEED_GIVEN)) {
        static std::atomic<uint64_t> calls{0};
        const uint64_t k = calls.fetch_add(1);
        memcpy(seed, o->seed, 32);
        seed[5] ^= (uint32_t)k; seed[6] ^= (uint32_t)(k >> 32);
        return H2V_OK;
    }
    size_t got = 0;
    while (got < 32) {
        const ssize_t r = getrandom((uint8_t *)seed + got, 32 - got, 0);
        if (r <= 0) return fail(H2V_E_DEVICE, "no OS randomness for the batch coefficients (getrandom)");
        got += (size_t)r;
    }
    return H2V_OK;
}
static int rlc_check_batch(const h2v_plan *p, const h2v_batch *b, const uint8_t *accept) {
    if (!p || !b || !accept) return fail(H2V_E_ARG, "null argument");
    if (b->n && (!b->proofs || !b->proof_off)) return fail(H2V_E_ARG, "null proofs / offsets");
    if (b->n && p->d.n_pi && !b->instances) return fail(H2V_E_ARG, "plan has public inputs but instances == NULL");
    if (b->n && p->d.n_ci && !b->committed) return fail(H2V_E_ARG, "plan has a committed instance but committed == NULL");
    if (b->n > (1ull << 22)) return fail(H2V_E_LIMIT, "RLC batches are limited to 2^22 proofs");
    return H2V_OK;
}
// an RLC call on an ORDINARY workspace: the batch check, or - routed - the per-proof pipeline (flags[0] = 0: "fell back")
static int run_rlc_or_routed(const h2v_plan *p, uint32_t n, const uint8_t *proofs, const uint64_t *off, const uint8_t *inst, const uint8_t *ci,
                             uint8_t *accept, uint32_t *status_out, h2v_workspace *w, hipStream_t st, const uint32_t seed[8], bool one_stream_opt) {
    int rc = rlc_stats_ensure(w);
    if (rc) return rc;
    if (rlc_route(w)) {
        if ((rc = rlc_ensure(w, p))) return rc;
        HIPCHK(hipMemsetAsync(w->rlc->flags, 0, 4, st));
        w->rlc->routed_ring[w->rlc->calls % h2v_workspace::RING] = 1;   // (its event set stays unrecorded: h2v_workspace_rlc_result reports zeros)
        w->rlc->calls++;
        w->rlc->run_len++;
        w->rlc->last_routed = true;
        rc = run_routed(p, n, proofs, off, inst, ci, accept, status_out, w, st, w->rlc_stats);
    } else {
        rc = run_rlc(p, n, proofs, off, inst, ci, accept, status_out, w, st, seed, one_stream_opt);
        if (rc == H2V_OK) { w->rlc->last_routed = false; w->rlc->routed_ring[(w->rlc->calls - 1) % h2v_workspace::RING] = 0; }
    }
    if (rc == H2V_OK) HIPCHK(hipMemcpyAsync(w->h_rlc_stats, w->rlc_stats, 8, hipMemcpyDeviceToHost, st));
    return rc;
}
extern "C" int h2v_verify_batch_rlc_device(const h2v_plan *p, const h2v_batch *b, uint8_t *accept, uint32_t *status, h2v_workspace *ws,
                                           void *stream, const h2v_rlc_opts *opts) {
    int rc = rlc_check_batch(p, b, accept);
    if (rc) return rc;
    if (b->n == 0) return H2V_OK;
    if (!ws) return fail(H2V_E_ARG, "the RLC entry points need a workspace (results of the batch check live in it)");
    ALIVE(p); ALIVE(ws);
    HIPCHK(hipSetDevice(p->device));
    if ((rc = ws_fits(ws, p, b->n, false))) return rc;
    if (ws->pending) return fail(H2V_E_ARG, "the workspace has a host batch in flight: call h2v_verify_batch_wait first");
    if ((rc = null_stream_check(ws, stream))) return rc;
    // recursive plans fold an accumulator per proof (the challenge hashes that proof's own MSM result): no batch form
    if (!rlc_supported(p)) {
        if (ws->n_lanes) return run_laned(p, (uint32_t)b->n, b->proofs, b->proof_off, b->instances, b->committed, accept, status, ws, (hipStream_t)stream, false, nullptr, false);
        return run_pipeline(p->d, (uint32_t)b->n, b->proofs, b->proof_off, b->instances, b->committed, accept, status, ws, (hipStream_t)stream, nullptr, false);
    }
    uint32_t seed[8];
    if ((rc = rlc_seed(opts, seed))) return rc;
    if (ws->n_lanes && co_wanted(ws, p, b->n)) return coalesce_call(p, b, accept, status, ws, (hipStream_t)stream, true, seed);
    if (ws->n_lanes) return run_laned(p, (uint32_t)b->n, b->proofs, b->proof_off, b->instances, b->committed, accept, status, ws, (hipStream_t)stream, true, seed, false);
    return run_rlc_or_routed(p, (uint32_t)b->n, b->proofs, b->proof_off, b->instances, b->committed, accept, status, ws, (hipStream_t)stream, seed,
                             opts && (opts->flags & H2V_RLC_ONE_STREAM));
}
extern "C" int h2v_verify_batch_rlc(const h2v_plan *p, const h2v_batch *b, uint8_t *accept, h2v_workspace *ws, const h2v_rlc_opts *opts,
                                    int *fell_back) {
    int rc = rlc_check_batch(p, b, accept);
    if (rc) return rc;
    return verify_host(p, b, accept, ws, H2V_SUBMIT_RLC, opts, fell_back);
}
// After the stream of an RLC call has been synchronised: did the batch check pass (1) or did the per-proof kernels run (0)?
// kernel times of a past call (calls_back = 0: the most recent).
static int rlc_timings_of(const RlcWs *r, uint32_t calls_back, h2v_rlc_timings *tm);
extern "C" int h2v_workspace_rlc_result(h2v_workspace *w, uint32_t calls_back, uint32_t *batch_accepted, h2v_rlc_timings *tm) {
    ALIVE(w);
    if (w && w->n_lanes) {
        // laned: the AND of the chunks' batch verdicts; times summed over the chunks, total = first start .. last verdict
        if (w->calls == 0 || calls_back >= h2v_workspace::RING || calls_back >= w->calls) return fail(H2V_E_ARG, "no such call in the event ring");
        const int slot = (int)((w->calls - 1 - calls_back) % h2v_workspace::RING);
        if (!w->lring_rlc[slot]) return fail(H2V_E_ARG, "that call did not run in RLC mode");
        if (w->lring_co[slot] && w->lring_share[slot] == 0.0f) return fail(H2V_E_ARG, "that call is in a group of coalesced calls that has not run yet: h2v_workspace_join first");
        if (w->lring_routed[slot]) {       // routed to the per-proof kernels: no batch check ran (h2v_workspace_timings has no record either)
            if (batch_accepted) *batch_accepted = 0;
            if (tm) memset(tm, 0, sizeof *tm);
            return H2V_OK;
        }
        if (batch_accepted) {
            uint32_t failed = 0;
            HIPCHK(hipMemcpy(&failed, w->rlc_fail + slot, 4, hipMemcpyDeviceToHost));
            *batch_accepted = failed == 0 ? 1u : 0u;
        }
        if (!tm) return H2V_OK;
        memset(tm, 0, sizeof *tm);
        hipEvent_t first = nullptr;
        for (uint32_t c = 0; c < w->lring_chunks[slot]; c++) {
            uint32_t l; uint64_t idx;
            laned_chunk_pos(w, slot, c, true, &l, &idx);
            h2v_workspace *lw = w->lane[l];
            // the buffers (and event ring) the chunk ran on: the lane's current ones, or parked since by another plan's call
            const RlcWs *r1 = w->lring_rlc_obj[slot][l];
            bool alive = r1 && lw->rlc == r1;
            for (const RlcWs *q : lw->rlc_parked) alive = alive || (r1 && q == r1);
            if (!alive || r1->calls - 1 - idx >= (uint64_t)h2v_workspace::RING) return fail(H2V_E_ARG, "the lanes' event rings have wrapped since that call");
            h2v_rlc_timings t1;
            int rc = rlc_timings_of(r1, (uint32_t)(r1->calls - 1 - idx), &t1);
            if (rc) return rc;
            tm->transcript_combiner_ms += t1.transcript_combiner_ms; tm->g1_decompress_ms += t1.g1_decompress_ms; tm->prepare_ms += t1.prepare_ms;
            tm->bucket_sort_ms += t1.bucket_sort_ms; tm->bucket_accumulate_ms += t1.bucket_accumulate_ms; tm->bucket_reduce_ms += t1.bucket_reduce_ms;
            tm->pairing_ms += t1.pairing_ms;
            tm->msm_terms = t1.msm_terms; tm->window_bits = t1.window_bits; tm->windows = t1.windows; tm->max_chain = t1.max_chain;
            const hipEvent_t *ev = r1->ring[idx % h2v_workspace::RING];
            if (!first) first = ev[0];
            float span = 0;
            HIPCHK(hipEventElapsedTime(&span, first, ev[9]));
            if (span > tm->total_ms) tm->total_ms = span;
        }
        if (w->lring_co[slot]) {     // a coalesced call: its share of the one batch check that served its group (total_ms: the group's)
            const float f = w->lring_share[slot];
            tm->transcript_combiner_ms *= f; tm->g1_decompress_ms *= f; tm->prepare_ms *= f; tm->bucket_sort_ms *= f;
            tm->bucket_accumulate_ms *= f; tm->bucket_reduce_ms *= f; tm->pairing_ms *= f;
        }
        return H2V_OK;
    }
    if (!w || !w->rlc || w->rlc->calls == 0) return fail(H2V_E_ARG, "no RLC call was made with this workspace");
    RlcWs *r = w->rlc;
    HIPCHK(hipSetDevice(w->device));
    if (calls_back >= r->run_len) return fail(H2V_E_ARG, "an RLC call of another plan has used this workspace since that call");
    if (batch_accepted) HIPCHK(hipMemcpy(batch_accepted, r->flags, 4, hipMemcpyDeviceToHost));
    return tm ? rlc_timings_of(r, calls_back, tm) : H2V_OK;
}
static int rlc_timings_of(const RlcWs *r, uint32_t calls_back, h2v_rlc_timings *tm) {
    {
        if (calls_back >= h2v_workspace::RING || calls_back >= r->calls) return fail(H2V_E_ARG, "no such call in the event ring");
        const hipEvent_t *ev = r->ring[(r->calls - 1 - calls_back) % h2v_workspace::RING];
        memset(tm, 0, sizeof *tm);
        if (r->routed_ring[(r->calls - 1 - calls_back) % h2v_workspace::RING]) return H2V_OK;   // routed: no batch check ran
        HIPCHK(hipEventSynchronize(ev[9]));
        HIPCHK(hipEventElapsedTime(&tm->g1_decompress_ms, ev[0], ev[1]));
        HIPCHK(hipEventElapsedTime(&tm->transcript_combiner_ms, ev[2], ev[3]));
        HIPCHK(hipEventElapsedTime(&tm->prepare_ms, ev[10], ev[4]));
        HIPCHK(hipEventElapsedTime(&tm->bucket_sort_ms, ev[5], ev[6]));
        HIPCHK(hipEventElapsedTime(&tm->bucket_accumulate_ms, ev[6], ev[7]));
        HIPCHK(hipEventElapsedTime(&tm->bucket_reduce_ms, ev[7], ev[8]));
        HIPCHK(hipEventElapsedTime(&tm->pairing_ms, ev[8], ev[9]));
        float a0, a1;
        HIPCHK(hipEventElapsedTime(&a0, ev[0], ev[9]));
        HIPCHK(hipEventElapsedTime(&a1, ev[2], ev[9]));
        tm->total_ms = a0 > a1 ? a0 : a1;
        tm->msm_terms = r->last_terms; tm->window_bits = r->last_c; tm->windows = r->last_W; tm->max_chain = r->last_chain;
    }
    return H2V_OK;
}

// ---------------------------------------------------------------------------------------------- trace
extern "C" int h2v_trace(const h2v_plan *p, const uint8_t *proof, size_t proof_len, const uint8_t *instances,
                         const uint8_t *committed, uint8_t *scalars_out, uint8_t *msm_scalars_out, uint8_t el_out[96],
                         uint8_t er_out[96], uint32_t *status_out, uint8_t *accept_out) {
    if (!p || !proof) return fail(H2V_E_ARG, "null argument");
    ALIVE(p);
    HIPCHK(hipSetDevice(p->device));
    h2v_workspace *ws = nullptr;
    int rc = ws_create_for(p->d, p->device, 1, true, &ws);
    if (rc) return rc;
    uint64_t off[2] = {0, proof_len};
    h2v_batch b = {1, proof, off, instances, committed};
    uint8_t *d_pts96 = nullptr;
    rc = stage_inputs(p, &b, ws);
    if (rc == H2V_OK) rc = run_pipeline(p->d, 1, ws->in_proofs, ws->in_off, ws->in_inst, ws->in_ci, ws->accept, nullptr, ws, ws->hs, nullptr, true);
    if (rc == H2V_OK && hipStreamSynchronize(ws->hs) != hipSuccess) rc = fail(H2V_E_DEVICE, "trace pipeline failed");
    do {
        if (rc) break;
        if (hipMalloc((void **)&d_pts96, 192) != hipSuccess) { rc = fail(H2V_E_DEVICE, "hipMalloc failed"); break; }
        // el / er as they enter the pairing: after the accumulator fold when the plan is recursive
        if (p->d.ivc) hipLaunchKernelGGL(k_export_points, dim3(1), dim3(64), 0, nullptr, 1u, 1, ws->el2, d_pts96);
        else hipLaunchKernelGGL(k_export_points, dim3(1), dim3(64), 0, nullptr, 1u, 0, ws->pts + (size_t)p->d.pi_point * 24, d_pts96);
        hipLaunchKernelGGL(k_export_points, dim3(1), dim3(64), 0, nullptr, 1u, 1, p->d.ivc ? ws->er2 : ws->er, d_pts96 + 96);
        if (hipDeviceSynchronize() != hipSuccess) { rc = fail(H2V_E_DEVICE, "trace kernels failed"); break; }
        uint8_t both[192];
        if (hipMemcpy(both, d_pts96, 192, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(H2V_E_DEVICE, "download failed"); break; }
        if (el_out) memcpy(el_out, both, 96);
        if (er_out) memcpy(er_out, both + 96, 96);
        if (scalars_out && p->d.n_trace && hipMemcpy(scalars_out, ws->trace, (size_t)p->d.n_trace * 32, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(H2V_E_DEVICE, "download failed"); break; }
        if (msm_scalars_out && hipMemcpy(msm_scalars_out, ws->scalars, (size_t)p->d.n_terms * 32, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(H2V_E_DEVICE, "download failed"); break; }
        if (status_out && hipMemcpy(status_out, ws->status, 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(H2V_E_DEVICE, "download failed"); break; }
        if (accept_out && hipMemcpy(accept_out, ws->accept, 1, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(H2V_E_DEVICE, "download failed"); break; }
    } while (0);
    if (d_pts96) (void)hipFree(d_pts96);
    h2v_workspace_free(ws);
    return rc;
}

// ---------------------------------------------------------------------------------------------- probes
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { return hipMalloc(&p, n ? n : 8) == hipSuccess ? 0 : -1; }
    template <class T> T *as() { return (T *)p; }
};
static int pick_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(H2V_E_DEVICE, "no HIP device: the HIP backend is required (no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(H2V_E_ARG, "device index out of range");
    ALIVE_DEV(device);
    HIPCHK(hipSetDevice(device));
    return H2V_OK;
}
extern "C" int h2v_probe_field(int device, int op, uint32_t n, const uint32_t *a, const uint32_t *b, uint32_t *out) {
    int rc = pick_device(device);
    if (rc) return rc;
    if (op < 0 || op > 5 || !a || !b || !out || n == 0) return fail(H2V_E_ARG, "bad argument");
    const size_t bytes = (size_t)n * (op < 4 ? 48 : 32);
    DevBuf da, db, dout;
    if (da.alloc(bytes) || db.alloc(bytes) || dout.alloc(bytes)) return fail(H2V_E_DEVICE, "hipMalloc failed");
    HIPCHK(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_field, dim3((n + 63) / 64), dim3(64), 0, nullptr, op, n, da.as<uint32_t>(), db.as<uint32_t>(), dout.as<uint32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    return H2V_OK;
}
extern "C" int h2v_probe_quad_madd(int device, const uint32_t *pq /* 48 dwords */, int neg, uint32_t *out /* 210 dwords */) {
    int rc = pick_device(device);
    if (rc) return rc;
    DevBuf din, dout;
    if (din.alloc(48 * 4) || dout.alloc(210 * 4)) return fail(H2V_E_DEVICE, "hipMalloc failed");
    HIPCHK(hipMemcpy(din.p, pq, 48 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_quad_madd, dim3(1), dim3(64), 0, nullptr, din.as<uint32_t>(), neg, dout.as<uint32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dout.p, 210 * 4, hipMemcpyDeviceToHost));
    return H2V_OK;
}
extern "C" int h2v_probe_blake2b(int device, uint32_t n, uint32_t len, const uint8_t *msgs, uint8_t *digests) {
    int rc = pick_device(device);
    if (rc) return rc;
    if (!msgs || !digests || n == 0) return fail(H2V_E_ARG, "bad argument");
    DevBuf dm, dout;
    if (dm.alloc((size_t)n * len) || dout.alloc((size_t)n * 32)) return fail(H2V_E_DEVICE, "hipMalloc failed");
    if (len) HIPCHK(hipMemcpy(dm.p, msgs, (size_t)n * len, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_blake2b, dim3((n + 63) / 64), dim3(64), 0, nullptr, n, len, dm.as<uint8_t>(), dout.as<uint32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(digests, dout.p, (size_t)n * 32, hipMemcpyDeviceToHost));
    return H2V_OK;
}
// A throw-away plan whose "proof" is `slots` consecutive compressed points and whose MSM takes them in order.
struct MiniPlan {
    H2vDevPlan d{};
    DevBuf points, terms;
    int build(uint32_t slots, uint32_t n_terms, const uint32_t *lines_sg2, const uint32_t *lines_g2) {
        std::vector<uint32_t> pts(slots), terms_h(2 * (n_terms ? n_terms : 1));
        for (uint32_t j = 0; j < slots; j++) pts[j] = 48 * j;
        for (uint32_t t = 0; t < n_terms; t++) { terms_h[2 * t] = H2V_TERM_PROOF_POINT; terms_h[2 * t + 1] = t; }
        if (points.alloc(slots * 4) || terms.alloc(terms_h.size() * 4)) return -1;
        if (hipMemcpy(points.p, pts.data(), slots * 4, hipMemcpyHostToDevice) != hipSuccess) return -1;
        if (hipMemcpy(terms.p, terms_h.data(), terms_h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return -1;
        d.proof_len = 48 * slots; d.n_points = slots; d.n_terms = n_terms; d.n_main_terms = n_terms; d.pi_point = 0;
        d.points = points.as<uint32_t>(); d.terms = terms.as<uint32_t>();
        d.lines_sg2 = lines_sg2; d.lines_g2 = lines_g2;
        return 0;
    }
};
static int upload_offsets(DevBuf &doff, uint32_t n, uint32_t rec) {
    std::vector<uint64_t> off(n + 1);
    for (uint32_t i = 0; i <= n; i++) off[i] = (uint64_t)i * rec;
    if (doff.alloc((n + 1) * 8)) return -1;
    return hipMemcpy(doff.p, off.data(), (n + 1) * 8, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}
extern "C" int h2v_probe_g1_decompress(int device, uint32_t n, const uint8_t *compressed, uint8_t *xy_be, uint8_t *valid) {
    int rc = pick_device(device);
    if (rc) return rc;
    if (!compressed || !xy_be || !valid || n == 0) return fail(H2V_E_ARG, "bad argument");
    MiniPlan mp;
    DevBuf din, doff, dpts, dvalid, dout;
    if (mp.build(1, 0, nullptr, nullptr) || upload_offsets(doff, n, 48) || din.alloc((size_t)n * 48) || dpts.alloc((size_t)n * 96) ||
        dvalid.alloc(n) || dout.alloc((size_t)n * 96)) return fail(H2V_E_DEVICE, "probe setup failed");
    HIPCHK(hipMemcpy(din.p, compressed, (size_t)n * 48, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_g1_decompress, dim3((n + 63) / 64), dim3(128), 0, nullptr, mp.d, n, din.as<uint8_t>(), doff.as<uint64_t>(), (const uint8_t *)nullptr, (const uint8_t *)nullptr, dpts.as<uint32_t>(), dvalid.as<uint8_t>(), (uint32_t *)nullptr, 0u, (uint8_t *)nullptr);
    hipLaunchKernelGGL(k_export_points, dim3((n + 63) / 64), dim3(64), 0, nullptr, n, 0, dpts.as<uint32_t>(), dout.as<uint8_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(xy_be, dout.p, (size_t)n * 96, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(valid, dvalid.p, n, hipMemcpyDeviceToHost));
    return H2V_OK;
}
extern "C" int h2v_probe_g1_msm(int device, uint32_t n, uint32_t T, const uint8_t *scalars, const uint8_t *bases_compressed, uint8_t *out_xy_be) {
    int rc = pick_device(device);
    if (rc) return rc;
    if (!scalars || !bases_compressed || !out_xy_be || n == 0 || T == 0 || T > 64) return fail(H2V_E_ARG, "bad argument");
    ProbeOpts probe_opts;
    MiniPlan mp;
    DevBuf din, doff, dsc, dpts, dvalid, der, dout, dtab;
    if (mp.build(T, T, nullptr, nullptr) || upload_offsets(doff, n, 48 * T) || din.alloc((size_t)n * T * 48) || dsc.alloc((size_t)n * T * 32) ||
        dpts.alloc((size_t)n * T * 96) || dvalid.alloc((size_t)n * T) || der.alloc((size_t)n * 144) || dout.alloc((size_t)n * 96))
        return fail(H2V_E_DEVICE, "probe setup failed");
    HIPCHK(hipMemcpy(din.p, bases_compressed, (size_t)n * T * 48, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dsc.p, scalars, (size_t)n * T * 32, hipMemcpyHostToDevice));
    if (dtab.alloc((size_t)n * T * 448 * 4)) return fail(H2V_E_DEVICE, "hipMalloc failed");
    hipLaunchKernelGGL(k_g1_decompress, dim3((n * T + 63) / 64), dim3(128), 0, nullptr, mp.d, n, din.as<uint8_t>(), doff.as<uint64_t>(), (const uint8_t *)nullptr, (const uint8_t *)nullptr, dpts.as<uint32_t>(), dvalid.as<uint8_t>(), dtab.as<uint32_t>(), 0u, (uint8_t *)nullptr);
    launch_msm(mp.d, n, dsc.as<uint32_t>(), dpts.as<uint32_t>(), dtab.as<uint32_t>(), der.as<uint32_t>(), nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(k_export_points, dim3((n + 63) / 64), dim3(64), 0, nullptr, n, 1, der.as<uint32_t>(), dout.as<uint8_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_xy_be, dout.p, (size_t)n * 96, hipMemcpyDeviceToHost));
    return H2V_OK;
}

// the fixed-base launch of a split MSM over the plan's own VK-base terms, scalars given by the caller
extern "C" int h2v_probe_g1_msm_fixed(const h2v_plan *p, uint32_t n, uint32_t k, const uint8_t *scalars, uint8_t *out_xy_be, uint32_t *n_fix_out) {
    if (!p) return fail(H2V_E_ARG, "bad argument");
    ALIVE(p);
    if (n_fix_out) *n_fix_out = p->n_fix;
    if (!p->fix_tab || !p->n_fix) return fail(H2V_E_ARG, "this plan has no fixed-base tables");
    if (!scalars || !out_xy_be || n == 0 || k == 0 || k > 4) return fail(H2V_E_ARG, "bad argument");
    HIPCHK(hipSetDevice(p->device));
    const H2vDevPlan &d = p->d;
    DevBuf dsc, der, dout;
    if (dsc.alloc((size_t)n * d.n_fix * 32) || der.alloc((size_t)n * 144) || dout.alloc((size_t)n * 96)) return fail(H2V_E_DEVICE, "hipMalloc failed");
    HIPCHK(hipMemcpy(dsc.p, scalars, (size_t)n * d.n_fix * 32, hipMemcpyHostToDevice));
    H2vMsmArgs mf = {d.terms, d.n_var, d.n_fix, d.n_fix, 0, (uint32_t)H2V_SLOTS(d), {d.n_fix, d.n_fix, d.n_fix}, {der.as<uint32_t>(), nullptr, nullptr}, nullptr, d.vk_tab,
                     d.fix_tab, k, (d.n_fix + k - 1) / k};
    const uint32_t bs = mf.n_fixl <= 64 ? 64u : mf.n_fixl <= 256 ? 256u : 512u;
    if (mf.n_fixl > bs) return fail(H2V_E_LIMIT, "too many VK bases for one block");
    const uint32_t pbf = bs / mf.n_fixl;
    hipLaunchKernelGGL(k_g1_msm_fixed, dim3((n + pbf - 1) / pbf), dim3(bs), (size_t)bs * 172, nullptr, d, mf, n, pbf, dsc.as<uint32_t>(), (const uint32_t *)nullptr, (uint32_t *)nullptr);
    hipLaunchKernelGGL(k_export_points, dim3((n + 63) / 64), dim3(64), 0, nullptr, n, 1, der.as<uint32_t>(), dout.as<uint8_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_xy_be, dout.p, (size_t)n * 96, hipMemcpyDeviceToHost));
    return H2V_OK;
}

// sum_n s_n * P_n through the bucket MSM: n scalars (32 B LE, < r) and n compressed bases; out 96 B affine big-endian
extern "C" int h2v_probe_g1_msm_pippenger(int device, uint32_t n, const uint8_t *scalars, const uint8_t *bases_compressed, uint8_t *out_xy_be) {
    int rc = pick_device(device);
    if (rc) return rc;
    if (!scalars || !bases_compressed || !out_xy_be || n == 0 || n > (1u << 24)) return fail(H2V_E_ARG, "bad argument");
    ProbeOpts probe_opts;
    MiniPlan mp;
    DevBuf din, doff, dsc, dpts, dvalid, dres, dout;
    if (mp.build(1, 0, nullptr, nullptr) || upload_offsets(doff, n, 48) || din.alloc((size_t)n * 48) || dsc.alloc((size_t)n * 32) ||
        dpts.alloc((size_t)n * 96) || dvalid.alloc(n) || dres.alloc(144) || dout.alloc(96)) return fail(H2V_E_DEVICE, "probe setup failed");
    HIPCHK(hipMemcpy(din.p, bases_compressed, (size_t)n * 48, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dsc.p, scalars, (size_t)n * 32, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_g1_decompress, dim3((n + 63) / 64), dim3(128), 0, nullptr, mp.d, n, din.as<uint8_t>(), doff.as<uint64_t>(), (const uint8_t *)nullptr, (const uint8_t *)nullptr, dpts.as<uint32_t>(), dvalid.as<uint8_t>(), (uint32_t *)nullptr, 0u, (uint8_t *)nullptr);
    PipWs pw;
    if ((rc = pip_alloc(pw, n, 2))) return rc;
    PipArgs a = {};
    a.n = n; a.halves = 2; a.scal = dsc.as<uint32_t>(); a.pidx = nullptr; a.pool0 = dpts.as<uint32_t>(); a.n_pool0 = n; a.pool1 = nullptr; a.out = dres.as<uint32_t>();
    const PipWs *pws[1] = {&pw};
    rc = pip_launch(pws, &a, 1, nullptr, nullptr);
    if (rc == H2V_OK) {
        hipLaunchKernelGGL(k_export_points, dim3(1), dim3(64), 0, nullptr, 1u, 1, dres.as<uint32_t>(), dout.as<uint8_t>());
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = fail(H2V_E_DEVICE, "bucket MSM kernels failed");
        else if (hipMemcpy(out_xy_be, dout.p, 96, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(H2V_E_DEVICE, "download failed");
    }
    pip_free(pw);
    return rc;
}
extern "C" int h2v_probe_pairing_ex(const h2v_plan *p, uint32_t n, const uint8_t *p1c, const uint8_t *p2c, uint8_t *out, int impl, uint8_t *dbg);
extern "C" int h2v_probe_pairing(const h2v_plan *p, uint32_t n, const uint8_t *p1c, const uint8_t *p2c, uint8_t *out) {
    return h2v_probe_pairing_ex(p, n, p1c, p2c, out, -1, nullptr);
}
// impl: -1 default, 0 one-lane-per-proof kernel, 1 cooperative kernel (the launcher's choice of engine), 2 / 3 its narrow / wide
// engine whatever n; dbg (optional): n * 24 * 48 bytes
// (f after the Miller loop and - cooperative kernel only - after the final exponentiation; canonical LE limbs)
extern "C" int h2v_probe_pairing_ex(const h2v_plan *p, uint32_t n, const uint8_t *p1c, const uint8_t *p2c, uint8_t *out, int impl, uint8_t *dbg) {
    if (!p || !p1c || !p2c || !out || n == 0) return fail(H2V_E_ARG, "bad argument");
    ALIVE(p);
    ProbeOpts probe_opts;
    DevBuf ddbg;
    HIPCHK(hipSetDevice(p->device));
    MiniPlan mp;
    DevBuf din, doff, dsc, dpts, dvalid, der, dst, dacc, dtab;
    mp.d.lines28_sg2 = p->d.lines28_sg2;
    mp.d.lines28_g2 = p->d.lines28_g2;
    if (mp.build(2, 1, p->d.lines_sg2, p->d.lines_g2) || upload_offsets(doff, n, 96) || din.alloc((size_t)n * 96) || dsc.alloc((size_t)n * 32) ||
        dpts.alloc((size_t)n * 192) || dvalid.alloc((size_t)n * 2) || der.alloc((size_t)n * 144) || dst.alloc((size_t)n * 4) || dacc.alloc(n))
        return fail(H2V_E_DEVICE, "probe setup failed");
    // MSM "sum" is 1 * p2 (term 0 must read slot 1): patch the single term
    uint32_t term[2] = {H2V_TERM_PROOF_POINT, 1};
    HIPCHK(hipMemcpy(mp.terms.p, term, 8, hipMemcpyHostToDevice));
    std::vector<uint8_t> in((size_t)n * 96), one((size_t)n * 32, 0);
    for (uint32_t i = 0; i < n; i++) { memcpy(&in[(size_t)i * 96], p1c + (size_t)i * 48, 48); memcpy(&in[(size_t)i * 96 + 48], p2c + (size_t)i * 48, 48); one[(size_t)i * 32] = 1; }
    HIPCHK(hipMemcpy(din.p, in.data(), in.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dsc.p, one.data(), one.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dst.p, 0, (size_t)n * 4));
    if (dtab.alloc((size_t)n * 2 * 448 * 4)) return fail(H2V_E_DEVICE, "hipMalloc failed");
    hipLaunchKernelGGL(k_g1_decompress, dim3((n * 2 + 63) / 64), dim3(128), 0, nullptr, mp.d, n, din.as<uint8_t>(), doff.as<uint64_t>(), (const uint8_t *)nullptr, (const uint8_t *)nullptr, dpts.as<uint32_t>(), dvalid.as<uint8_t>(), dtab.as<uint32_t>(), 0u, (uint8_t *)nullptr);
    launch_msm(mp.d, n, dsc.as<uint32_t>(), dpts.as<uint32_t>(), dtab.as<uint32_t>(), der.as<uint32_t>(), nullptr, nullptr, nullptr);
    if (ddbg.alloc(dbg ? (size_t)n * 24 * 48 : 8)) return fail(H2V_E_DEVICE, "hipMalloc failed");
    launch_pairing_impl(impl, mp.d, n, dpts.as<uint32_t>(), dvalid.as<uint8_t>(), nullptr, der.as<uint32_t>(), nullptr, dst.as<uint32_t>(), dacc.as<uint8_t>(), dbg ? ddbg.as<uint32_t>() : nullptr, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dacc.p, n, hipMemcpyDeviceToHost));
    if (dbg) HIPCHK(hipMemcpy(dbg, ddbg.p, (size_t)n * 24 * 48, hipMemcpyDeviceToHost));
    return H2V_OK;
}
