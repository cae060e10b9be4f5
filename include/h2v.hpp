// C++ host-side mirror of the reference's verify interface for the hot path, header-only, over the C-ABI of h2v.h.
//
// The reference's toolchain (Rust) is absent from this image, so the layer a Rust maintainer would write above the
// FFI is given here in C++ with the reference's names and argument meaning:
//
//     let mut t = CircuitTranscript::<CardanoFriendlyBlake2b>::init_from_bytes(&proof);     // examples/simple_mul.rs:97
//     let guard = prepare(&vk, &[&[]], &[&[&instance]], &mut t)?;                            // examples/simple_mul.rs:98
//     guard.verify(&kzg_params.verifier_params())?;                                          // examples/simple_mul.rs:101
//
//     h2v::CircuitTranscript t = h2v::CircuitTranscript::init_from_bytes(proof);
//     h2v::Guard guard = h2v::prepare(vk, {}, {instance_scalars_le32}, t);                   // throws h2v::Error on misuse
//     guard.verify();                                                                        // throws h2v::VerifyError
//
// `VerifyingKey` wraps a plan blob produced by plutus_halo2_verifier_gen_amd.plan.compile_plan(vk).to_bytes()
// (the counterpart of extract_circuit, /root/reference/src/plutus_gen/extraction/mod.rs:31).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "h2v.h"

namespace h2v {

struct Error : std::runtime_error {  // API misuse / device error (negative H2V_E_* codes)
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
struct VerifyError : std::runtime_error {  // Err(plonk::Error): the proof does not verify
    uint32_t status;                       // H2V_ST_* bits
    explicit VerifyError(uint32_t st) : std::runtime_error("proof rejected (status " + std::to_string(st) + ")"), status(st) {}
};
inline void check(int rc) {
    if (rc != H2V_OK) throw Error(rc, h2v_last_error());
}

class VerifyingKey {  // VerifyingKey<F, KZGCommitmentScheme<Bls12>> + ParamsVerifierKZG for this path
  public:
    VerifyingKey(const uint8_t *plan_blob, size_t len, int device = 0) { check(h2v_plan_load(plan_blob, len, device, &p_)); }
    VerifyingKey(const VerifyingKey &) = delete;
    VerifyingKey &operator=(const VerifyingKey &) = delete;
    ~VerifyingKey() { h2v_plan_free(p_); }
    const h2v_plan *handle() const { return p_; }
    uint32_t proof_len() const { uint32_t v; check(h2v_plan_info(p_, &v, nullptr, nullptr, nullptr)); return v; }
    uint32_t n_public_inputs() const { uint32_t v; check(h2v_plan_info(p_, nullptr, &v, nullptr, nullptr)); return v; }
    uint32_t n_committed_instances() const { uint32_t v; check(h2v_plan_info(p_, nullptr, nullptr, &v, nullptr)); return v; }

  private:
    h2v_plan *p_ = nullptr;
};

class CircuitTranscript {  // verifier-side transcript: a cursor over the proof bytes (hashing happens on the GPU)
  public:
    static CircuitTranscript init_from_bytes(std::vector<uint8_t> proof) { return CircuitTranscript(std::move(proof)); }
    const std::vector<uint8_t> &bytes() const { return proof_; }
    void mark_consumed(size_t n) { consumed_ = n; }
    void assert_empty() const {  // examples/ivc.rs:92-94
        if (consumed_ != proof_.size()) throw VerifyError(0);
    }

  private:
    explicit CircuitTranscript(std::vector<uint8_t> p) : proof_(std::move(p)) {}
    std::vector<uint8_t> proof_;
    size_t consumed_ = 0;
};

class Guard {  // CS::VerificationGuard: consumed by verify() (Guard::verify) or check() (DualMSM::check)
  public:
    Guard(const VerifyingKey &vk, std::vector<uint8_t> proof, std::vector<uint8_t> instances, std::vector<uint8_t> committed)
        : vk_(vk), proof_(std::move(proof)), inst_(std::move(instances)), ci_(std::move(committed)) {}
    void verify() {
        const uint32_t st = run();
        if (st) throw VerifyError(st);
    }
    bool check() { return run() == 0; }

  private:
    uint32_t run() {
        if (used_) throw Error(H2V_E_ARG, "guard already consumed");
        used_ = true;
        uint32_t st = 0;
        uint8_t acc = 0;
        h2v::check(h2v_trace(vk_.handle(), proof_.data(), proof_.size(), inst_.empty() ? nullptr : inst_.data(),
                             ci_.empty() ? nullptr : ci_.data(), nullptr, nullptr, nullptr, nullptr, &st, &acc));
        return acc ? 0u : (st ? st : H2V_ST_PAIRING);
    }
    const VerifyingKey &vk_;
    std::vector<uint8_t> proof_, inst_, ci_;
    bool used_ = false;
};

// prepare(&vk, committed_instances, instances, &mut transcript): committed = 0 or 1 compressed G1 (48 B),
// instances = the public-input scalars of the single public column, 32 B little-endian each.
inline Guard prepare(const VerifyingKey &vk, const std::vector<std::vector<uint8_t>> &committed_instances,
                     const std::vector<std::vector<uint8_t>> &instances, CircuitTranscript &transcript) {
    if (instances.size() != vk.n_public_inputs()) throw Error(H2V_E_ARG, "wrong number of public inputs");
    if (committed_instances.size() != vk.n_committed_instances()) throw Error(H2V_E_ARG, "wrong number of committed instances");
    std::vector<uint8_t> inst, ci;
    for (const auto &s : instances) {
        if (s.size() != 32) throw Error(H2V_E_ARG, "instance scalars are 32 bytes little-endian");
        inst.insert(inst.end(), s.begin(), s.end());
    }
    for (const auto &c : committed_instances) {
        if (c.size() != 48) throw Error(H2V_E_ARG, "committed instances are 48-byte compressed G1");
        ci.insert(ci.end(), c.begin(), c.end());
    }
    transcript.mark_consumed(vk.proof_len());
    return Guard(vk, transcript.bytes(), std::move(inst), std::move(ci));
}

// The batched form the hardware wants: accept[i] for n independent proofs (host buffers).
inline std::vector<uint8_t> verify_batch(const VerifyingKey &vk, const h2v_batch &batch, h2v_workspace *ws = nullptr) {
    std::vector<uint8_t> accept(batch.n);
    check(h2v_verify_batch(vk.handle(), &batch, accept.data(), ws));
    return accept;
}
// The same vector through the batch-accept fast path (one random linear combination of the batch's pairing equations:
// one bucketed G1 MSM + one pairing; the per-proof kernels only if the batch check fails).  fell_back (optional) tells
// which of the two produced the vector.  seed: 32 bytes that provers cannot predict, or nullptr = drawn from the OS.
inline std::vector<uint8_t> verify_batch_rlc(const VerifyingKey &vk, const h2v_batch &batch, h2v_workspace *ws = nullptr,
                                             bool *fell_back = nullptr, const uint8_t *seed = nullptr) {
    std::vector<uint8_t> accept(batch.n);
    h2v_rlc_opts opts{};
    if (seed) { for (int k = 0; k < 32; k++) opts.seed[k] = seed[k]; opts.flags = H2V_RLC_SEED_GIVEN; }
    int fb = 0;
    check(h2v_verify_batch_rlc(vk.handle(), &batch, accept.data(), ws, seed ? &opts : nullptr, &fb));
    if (fell_back) *fell_back = fb != 0;
    return accept;
}

// A reusable workspace, and the streaming form: submit() returns once the batch is copied and everything is enqueued,
// wait() collects the accept vector.  Alternate two Workspaces to overlap the upload of one batch with the kernels of the
// previous one.
class Workspace {
  public:
    Workspace(const VerifyingKey &vk, uint64_t max_batch) : vk_(vk) { check(h2v_workspace_create(vk.handle(), max_batch, &w_)); }
    Workspace(const Workspace &) = delete;
    Workspace &operator=(const Workspace &) = delete;
    ~Workspace() { h2v_workspace_free(w_); }
    h2v_workspace *handle() const { return w_; }
    /// tuning hint (h2v_workspace_hint_in_flight): the caller keeps n batches in flight, one workspace each; never changes results
    void hint_in_flight(uint32_t n) { check(h2v_workspace_hint_in_flight(w_, n)); }
    void submit(const h2v_batch &batch, bool rlc = false) {
        check(h2v_verify_batch_submit(vk_.handle(), &batch, w_, rlc ? H2V_SUBMIT_RLC : 0u, nullptr));
        n_ = batch.n;
    }
    std::vector<uint8_t> wait(bool *fell_back = nullptr) {
        std::vector<uint8_t> accept(n_ ? n_ : 1);
        int fb = 0;
        check(h2v_verify_batch_wait(w_, accept.data(), &fb));
        accept.resize(n_);
        if (fell_back) *fell_back = fb != 0;
        return accept;
    }

  private:
    const VerifyingKey &vk_;
    h2v_workspace *w_ = nullptr;
    uint64_t n_ = 0;
};

// A verifier that keeps `depth` batches in flight (one workspace each, h2v_workspace_hint_in_flight(depth)): push() hands a
// batch over (its host buffers may be reused at once) and returns the accept vector of the batch pushed `depth` calls
// earlier, if there is one; drain() collects the rest in order.  depth 5 (per proof) / 11 (RLC) reach the device-resident
// throughput on one MI355X (tools/bench_host_path.py).  Every batch must fit `max_batch`.
class BatchStream {
  public:
    BatchStream(const VerifyingKey &vk, uint64_t max_batch, unsigned depth, bool rlc = false) : rlc_(rlc) {
        if (depth == 0) throw Error(H2V_E_ARG, "BatchStream: depth must be at least 1");
        for (unsigned k = 0; k < depth; k++) {
            ws_.emplace_back(new Workspace(vk, max_batch));
            ws_.back()->hint_in_flight(depth);
        }
    }
    ~BatchStream() { for (Workspace *w : ws_) delete w; }
    BatchStream(const BatchStream &) = delete;
    BatchStream &operator=(const BatchStream &) = delete;
    /// returns true and fills `done` when the batch pushed depth calls earlier has been collected
    bool push(const h2v_batch &batch, std::vector<uint8_t> *done = nullptr) {
        Workspace &w = *ws_[next_ % ws_.size()];
        bool have = false;
        if (next_ >= ws_.size()) {
            std::vector<uint8_t> acc = w.wait();
            if (done) *done = std::move(acc);
            have = true;
            collected_++;
        }
        w.submit(batch, rlc_);
        next_++;
        return have;
    }
    /// accept vectors of the batches still in flight, oldest first
    std::vector<std::vector<uint8_t>> drain() {
        std::vector<std::vector<uint8_t>> out;
        for (; collected_ < next_; collected_++) out.push_back(ws_[collected_ % ws_.size()]->wait());
        return out;
    }

  private:
    std::vector<Workspace *> ws_;
    uint64_t next_ = 0, collected_ = 0;
    bool rlc_;
};

}  // namespace h2v
