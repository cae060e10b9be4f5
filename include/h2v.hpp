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
// `VerifyingKey` wraps a plan: VerifyingKey::from_json(description) compiles one behind the C-ABI (h2v_plan_compile, the
// counterpart of extract_circuit, /root/reference/src/plutus_gen/extraction/mod.rs:31; the description is the JSON of
// docs/vk_schema.json), or it takes a blob made earlier (plutus_halo2_verifier_gen_amd.plan.compile_plan(vk).to_bytes()
// gives the same bytes).
#pragma once
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
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
/// h2v_shutdown: releases everything the library owns on `device` (-1: every device) - pool streams, workspaces, plans.
/// Call it once after the last verify call and before main() returns (or hold a ShutdownGuard in main): objects of this
/// header that are still alive afterwards are empty shells whose destructors only free host memory.
inline void shutdown(int device = -1) { check(h2v_shutdown(device)); }
struct ShutdownGuard {   // `h2v::ShutdownGuard guard;` as the first local of main(): runs after every other local is gone
    ShutdownGuard() = default;
    ShutdownGuard(const ShutdownGuard &) = delete;
    ShutdownGuard &operator=(const ShutdownGuard &) = delete;
    ~ShutdownGuard() { (void)h2v_shutdown(-1); }
};

class VerifyingKey {  // VerifyingKey<F, KZGCommitmentScheme<Bls12>> + ParamsVerifierKZG for this path
  public:
    VerifyingKey(const uint8_t *plan_blob, size_t len, int device = 0) { check(h2v_plan_load(plan_blob, len, device, &p_)); }
    /// from the verifying-key description (JSON, docs/vk_schema.json): h2v_plan_compile + h2v_plan_load, no Python in the loop
    static std::vector<uint8_t> compile(const std::string &vk_json) {
        uint8_t *blob = nullptr;
        size_t n = 0;
        check(h2v_plan_compile(vk_json.data(), vk_json.size(), &blob, &n));
        std::vector<uint8_t> out(blob, blob + n);
        h2v_blob_free(blob);
        return out;
    }
    static std::unique_ptr<VerifyingKey> from_json(const std::string &vk_json, int device = 0) {
        const std::vector<uint8_t> blob = compile(vk_json);
        return std::unique_ptr<VerifyingKey>(new VerifyingKey(blob.data(), blob.size(), device));
    }
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
// (A Workspace, BatchStream or NodeStream keeps a reference to its VerifyingKey: the key must outlive it.)
class Workspace {
  public:
    Workspace(const VerifyingKey &vk, uint64_t max_batch) : vk_(vk) { check(h2v_workspace_create(vk.handle(), max_batch, &w_)); }
    /// a LANED workspace (h2v_workspace_create_lanes): calls are cut into chunks that the library pipelines through its own
    /// lanes and streams; n_lanes / chunk = 0: the library's choice
    Workspace(const VerifyingKey &vk, uint64_t max_batch, uint32_t n_lanes, uint32_t chunk) : vk_(vk) {
        check(h2v_workspace_create_lanes(vk.handle(), max_batch, n_lanes, chunk, &w_));
    }
    /// ONE laned workspace for several keys of one device (h2v_workspace_create_multi: lanes sized for the largest of every
    /// dimension); submit(vk, batch) names the key of each batch, submit(batch) means the first key
    Workspace(const std::vector<const VerifyingKey *> &vks, uint64_t max_batch, uint32_t n_lanes = 0, uint32_t chunk = 0) : vk_(*vks.at(0)) {
        std::vector<const h2v_plan *> ps;
        for (const VerifyingKey *k : vks) ps.push_back(k->handle());
        check(h2v_workspace_create_multi(ps.data(), (uint32_t)ps.size(), max_batch, n_lanes, chunk, &w_));
    }
    Workspace(const Workspace &) = delete;
    Workspace &operator=(const Workspace &) = delete;
    ~Workspace() { h2v_workspace_free(w_); }
    h2v_workspace *handle() const { return w_; }
    const VerifyingKey &vk() const { return vk_; }
    /// tuning hint (h2v_workspace_hint_in_flight): the caller keeps n batches in flight, one workspace each; never changes results
    void hint_in_flight(uint32_t n) { check(h2v_workspace_hint_in_flight(w_, n)); }
    /// calls of n proofs the workspace keeps in flight before a call waits for a lane (h2v_workspace_depth)
    uint32_t depth(uint64_t n, bool rlc = false) const {
        uint32_t d = 1;
        check(h2v_workspace_depth(w_, n, rlc ? 1 : 0, &d));
        return d;
    }
    void submit(const h2v_batch &batch, bool rlc = false) { submit(vk_, batch, rlc); }
    void submit(const VerifyingKey &vk, const h2v_batch &batch, bool rlc = false) {
        check(h2v_verify_batch_submit(vk.handle(), &batch, w_, rlc ? H2V_SUBMIT_RLC : 0u, nullptr));
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

// A verifier that keeps up to `depth` host-buffer batches in flight on ONE laned workspace (`depth` lanes, one chunk per
// batch): push() hands a batch over (its host buffers may be reused at once) and, once `depth` batches are in flight,
// returns the accept vector of the OLDEST; drain() collects the rest in order.  depth 6 (per proof) / 11-16 (RLC) reach the
// device-resident throughput on one MI355X (tools/bench_host_path.py).  Every batch must fit `max_batch`.  (Round 2 needed
// `depth` workspaces for this.)
class BatchStream {
  public:
    BatchStream(const VerifyingKey &vk, uint64_t max_batch, unsigned depth, bool rlc = false)
        : depth_(depth), rlc_(rlc), ws_(vk, max_batch, depth == 0 || depth > 16 ? 1u : depth, (uint32_t)max_batch) {
        if (depth == 0 || depth > 16) throw Error(H2V_E_ARG, "BatchStream: depth must be 1 .. 16 (a laned workspace has at most 16 lanes)");
    }
    BatchStream(const BatchStream &) = delete;
    BatchStream &operator=(const BatchStream &) = delete;
    /// returns true and fills `done` when the oldest batch in flight has been collected (`depth` were in flight)
    bool push(const h2v_batch &batch, std::vector<uint8_t> *done = nullptr) {
        bool have = false;
        if (sizes_.size() >= depth_) {
            std::vector<uint8_t> acc = collect();
            if (done) *done = std::move(acc);
            have = true;
        }
        check(h2v_verify_batch_submit(ws_.vk().handle(), &batch, ws_.handle(), rlc_ ? H2V_SUBMIT_RLC : 0u, nullptr));
        sizes_.push_back(batch.n);
        return have;
    }
    /// accept vectors of the batches still in flight, oldest first
    std::vector<std::vector<uint8_t>> drain() {
        std::vector<std::vector<uint8_t>> out;
        while (!sizes_.empty()) out.push_back(collect());
        return out;
    }

  private:
    std::vector<uint8_t> collect() {
        std::vector<uint8_t> accept(sizes_.front() ? sizes_.front() : 1);
        int fb = 0;
        check(h2v_verify_batch_wait(ws_.handle(), accept.data(), &fb));
        accept.resize(sizes_.front());
        sizes_.pop_front();
        return accept;
    }
    unsigned depth_;
    bool rlc_;
    Workspace ws_;
    std::deque<uint64_t> sizes_;
};

// Every GPU of one node behind one object, in ONE process (proofs are independent, so a batch is
// cut into contiguous index ranges [g n / G, (g + 1) n / G), one per device; the plan is replicated; nothing is exchanged
// but the accept bytes, which come back through each device's own pinned staging buffer).  Per device: its own
// VerifyingKey (plan upload), a BatchStream of `depth` workspaces and a host thread that stages and submits that device's
// shard, so the shards of one batch are packed and uploaded side by side.  push() / drain() as BatchStream; verify() is
// the blocking form.  The same device may be listed more than once (two pipelines on one GPU: how the GPU test runs on a
// one-GPU box).
class NodeStream {
  public:
    NodeStream(const uint8_t *plan_blob, size_t len, const std::vector<int> &devices, uint64_t max_batch_per_device, unsigned depth,
               bool rlc = false) {
        if (devices.empty() || depth == 0) throw Error(H2V_E_ARG, "NodeStream: at least one device and depth >= 1");
        for (int dev : devices) {
            std::unique_ptr<Dev> d(new Dev());
            d->vk.reset(new VerifyingKey(plan_blob, len, dev));
            d->bs.reset(new BatchStream(*d->vk, max_batch_per_device, depth, rlc));
            devs_.push_back(std::move(d));
        }
        for (auto &d : devs_) d->worker = std::thread([dp = d.get()] { dp->run(); });
    }
    ~NodeStream() {
        for (auto &d : devs_) { { std::lock_guard<std::mutex> l(d->mu); d->stop = true; } d->cv.notify_all(); }
        for (auto &d : devs_) if (d->worker.joinable()) d->worker.join();
    }
    NodeStream(const NodeStream &) = delete;
    NodeStream &operator=(const NodeStream &) = delete;
    size_t n_devices() const { return devs_.size(); }
    /// shard g of n proofs: the contiguous range [lo, hi)  (plutus_halo2_verifier_gen_amd/shard.py: shard_range)
    static std::pair<uint64_t, uint64_t> shard_range(uint64_t n, size_t g, size_t G) { return {n * g / G, n * (g + 1) / G}; }
    /// hands a batch over (the caller's buffers may be reused when this returns); true + `done` = the accept vector of the
    /// batch pushed `depth` calls earlier, reassembled in proof order
    bool push(const h2v_batch &b, std::vector<uint8_t> *done = nullptr) {
        const size_t G = devs_.size();
        const uint32_t n_pi = devs_[0]->vk->n_public_inputs(), n_ci = devs_[0]->vk->n_committed_instances();
        for (size_t g = 0; g < G; g++) {
            const auto r = shard_range(b.n, g, G);
            Dev &d = *devs_[g];
            Job j;
            j.n = r.second - r.first;
            // offsets rebased to the shard's first byte, so that only the shard's bytes travel to its device
            j.off.resize(j.n + 1);
            for (uint64_t i = 0; i <= j.n; i++) j.off[i] = b.proof_off[r.first + i] - b.proof_off[r.first];
            j.proofs = b.proofs + b.proof_off[r.first];
            j.inst = b.instances ? b.instances + r.first * n_pi * 32 : nullptr;
            j.ci = (b.committed && n_ci) ? b.committed + r.first * 48 : nullptr;
            { std::lock_guard<std::mutex> l(d.mu); d.jobs.push_back(std::move(j)); d.staged = false; }
            d.cv.notify_all();
        }
        bool have = true;
        std::string err;
        for (auto &d : devs_) {   // until every device has staged its shard (then the caller's buffers are free again)
            std::unique_lock<std::mutex> l(d->mu);
            d->cv.wait(l, [&] { return d->staged; });
            if (!d->error.empty()) err = d->error;
            have = have && !d->results.empty();
        }
        if (!err.empty()) throw Error(H2V_E_DEVICE, err);
        sizes_.push_back(b.n);
        if (have && done) *done = collect();
        else if (have) (void)collect();
        return have;
    }
    std::vector<std::vector<uint8_t>> drain() {
        for (auto &d : devs_) {
            { std::lock_guard<std::mutex> l(d->mu); d->drain = true; d->staged = false; }
            d->cv.notify_all();
        }
        for (auto &d : devs_) {
            std::unique_lock<std::mutex> l(d->mu);
            d->cv.wait(l, [&] { return d->staged; });
            if (!d->error.empty()) throw Error(H2V_E_DEVICE, d->error);
        }
        std::vector<std::vector<uint8_t>> out;
        while (!sizes_.empty()) out.push_back(collect());
        return out;
    }
    std::vector<uint8_t> verify(const h2v_batch &b) {
        if (!sizes_.empty()) throw Error(H2V_E_ARG, "NodeStream::verify with batches in flight: drain() first");
        push(b);
        return drain().at(0);
    }

  private:
    struct Job { uint64_t n = 0; std::vector<uint64_t> off; const uint8_t *proofs = nullptr, *inst = nullptr, *ci = nullptr; };
    struct Dev {
        std::unique_ptr<VerifyingKey> vk;
        std::unique_ptr<BatchStream> bs;
        std::thread worker;
        std::mutex mu;
        std::condition_variable cv;
        std::deque<Job> jobs;
        std::deque<std::vector<uint8_t>> results;
        bool stop = false, drain = false, staged = true;
        std::string error;
        void run() {
            for (;;) {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&] { return stop || drain || !jobs.empty(); });
                if (stop) return;
                try {
                    if (!jobs.empty()) {
                        Job j = std::move(jobs.front());
                        jobs.pop_front();
                        l.unlock();
                        std::vector<uint8_t> done;
                        bool have = false;
                        if (j.n) {
                            const h2v_batch sb{j.n, j.proofs, j.off.data(), j.inst, j.ci};
                            have = bs->push(sb, &done);
                        } else {
                            have = pending_empty_push(&done);
                        }
                        l.lock();
                        if (have) results.push_back(std::move(done));
                    } else {   // drain
                        l.unlock();
                        std::vector<std::vector<uint8_t>> rest = bs->drain();
                        l.lock();
                        for (auto &v : rest) results.push_back(std::move(v));
                        drain = false;
                    }
                } catch (const std::exception &e) {
                    if (!l.owns_lock()) l.lock();
                    error = e.what();
                    drain = false;
                }
                staged = true;
                l.unlock();
                cv.notify_all();
            }
        }
        // (a device whose shard is empty - fewer proofs than devices - still takes part in the lockstep of the streams)
        bool pending_empty_push(std::vector<uint8_t> *done) {
            static const uint64_t zero_off[1] = {0};
            const h2v_batch sb{0, nullptr, zero_off, nullptr, nullptr};
            return bs->push(sb, done);
        }
    };
    std::vector<uint8_t> collect() {   // the oldest batch: one result per device, concatenated in device (= proof) order
        std::vector<uint8_t> out;
        out.reserve(sizes_.front());
        for (auto &d : devs_) {
            std::lock_guard<std::mutex> l(d->mu);
            if (d->results.empty()) throw Error(H2V_E_ARG, "NodeStream: a device has no finished batch");
            out.insert(out.end(), d->results.front().begin(), d->results.front().end());
            d->results.pop_front();
        }
        sizes_.pop_front();
        return out;
    }
    std::vector<std::unique_ptr<Dev>> devs_;
    std::deque<uint64_t> sizes_;
};

}  // namespace h2v
