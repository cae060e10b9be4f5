// Test driver for include/h2v.hpp, written the way the reference's examples use the verifier
// (examples/simple_mul.rs:97-104): init_from_bytes -> prepare -> verify, one proof at a time, then the same proofs
// through verify_batch.
//
// usage: h2v_cpp_driver <plan.bin | vk.json> <batch.bin>      (vk.json: the key description is compiled behind the C-ABI)
// batch.bin: u32 n, u32 n_pi, u32 has_ci, then per proof: u32 len, bytes, n_pi * 32 B instances, [48 B committed]
// prints "single <bits>" and "batch <bits>", or "error <code> <text>" (exit 2) when the library cannot run.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>

#include "h2v.hpp"

static std::vector<uint8_t> slurp(const char *path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static uint32_t rd32(const std::vector<uint8_t> &b, size_t &o) {
    uint32_t v;
    std::memcpy(&v, b.data() + o, 4);
    o += 4;
    return v;
}

int main(int argc, char **argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: %s plan.bin batch.bin\n", argv[0]); return 64; }
    h2v::ShutdownGuard shutdown_last;   // h2v_shutdown(-1) after every other local is gone: the library's pool streams do not outlive main()
    try {
        std::vector<uint8_t> blob = slurp(argv[1]);
        const std::vector<uint8_t> bb = slurp(argv[2]);
        const std::string arg1 = argv[1];
        if (arg1.size() > 5 && arg1.substr(arg1.size() - 5) == ".json") {      // JSON -> plan, no Python in the loop
            blob = h2v::VerifyingKey::compile(std::string(blob.begin(), blob.end()));
            std::printf("compiled_plan_bytes %zu\n", blob.size());
        }
        h2v::VerifyingKey vk(blob.data(), blob.size(), 0);
        size_t o = 0;
        const uint32_t n = rd32(bb, o), n_pi = rd32(bb, o), has_ci = rd32(bb, o);
        if (n_pi != vk.n_public_inputs() || has_ci != vk.n_committed_instances()) throw h2v::Error(H2V_E_ARG, "batch / plan mismatch");
        std::vector<uint8_t> proofs, inst, ci;
        std::vector<uint64_t> off{0};
        std::string single;
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t len = rd32(bb, o);
            std::vector<uint8_t> proof(bb.begin() + o, bb.begin() + o + len);
            o += len;
            std::vector<std::vector<uint8_t>> instances, committed;
            for (uint32_t k = 0; k < n_pi; k++) { instances.emplace_back(bb.begin() + o, bb.begin() + o + 32); o += 32; }
            if (has_ci) { committed.emplace_back(bb.begin() + o, bb.begin() + o + 48); o += 48; }
            proofs.insert(proofs.end(), proof.begin(), proof.end());
            off.push_back(proofs.size());
            for (auto &s : instances) inst.insert(inst.end(), s.begin(), s.end());
            for (auto &c : committed) ci.insert(ci.end(), c.begin(), c.end());
            // the reference's call sequence
            h2v::CircuitTranscript t = h2v::CircuitTranscript::init_from_bytes(proof);
            h2v::Guard guard = h2v::prepare(vk, committed, instances, t);
            bool ok = true;
            try { guard.verify(); } catch (const h2v::VerifyError &) { ok = false; }
            single.push_back(ok ? '1' : '0');
        }
        h2v_batch batch{n, proofs.data(), off.data(), inst.data(), has_ci ? ci.data() : nullptr};
        const std::vector<uint8_t> acc = h2v::verify_batch(vk, batch);
        std::string bits;
        for (uint8_t a : acc) bits.push_back(a ? '1' : '0');
        std::printf("single %s\nbatch %s\n", single.c_str(), bits.c_str());
        // the same batch through the batch-accept fast path, and streamed over two workspaces (per proof, then RLC)
        bool fell_back = false;
        const std::vector<uint8_t> racc = h2v::verify_batch_rlc(vk, batch, nullptr, &fell_back);
        std::string rbits;
        for (uint8_t a : racc) rbits.push_back(a ? '1' : '0');
        std::printf("rlc %s\nrlc_fell_back %d\n", rbits.c_str(), fell_back ? 1 : 0);
        h2v::Workspace w0(vk, n), w1(vk, n);
        w0.hint_in_flight(5);   // (a tuning hint: other launch shapes, the same verdicts)
        w0.submit(batch, false);
        w1.submit(batch, true);
        std::string s0, s1;
        for (uint8_t a : w0.wait()) s0.push_back(a ? '1' : '0');
        for (uint8_t a : w1.wait()) s1.push_back(a ? '1' : '0');
        std::printf("stream0 %s\nstream1 %s\n", s0.c_str(), s1.c_str());
        // seven batches through a stream of depth 3 (per proof, then RLC): every collected vector equals the blocking call's
        for (int rlc = 0; rlc < 2; rlc++) {
            h2v::BatchStream bs(vk, n, 3, rlc != 0);
            std::vector<std::vector<uint8_t>> got;
            std::vector<uint8_t> done;
            for (int k = 0; k < 7; k++)
                if (bs.push(batch, &done)) got.push_back(done);
            for (auto &v : bs.drain()) got.push_back(v);
            bool same = got.size() == 7;
            for (auto &v : got) same = same && v == acc;
            std::printf("batch_stream_%s %d\n", rlc ? "rlc" : "per_proof", same ? 1 : 0);
        }
        // the node-level host API: one process, a device list (the same GPU twice on a one-GPU box), contiguous shards,
        // verdicts reassembled in order - blocking, then five batches streamed at depth 2; and a laned workspace
        {
            h2v::NodeStream node(blob.data(), blob.size(), {0, 0}, n, 2, false);
            bool same = node.verify(batch) == acc;
            std::vector<std::vector<uint8_t>> got;
            std::vector<uint8_t> done;
            for (int k = 0; k < 5; k++)
                if (node.push(batch, &done)) got.push_back(done);
            for (auto &v : node.drain()) got.push_back(v);
            same = same && got.size() == 5;
            for (auto &v : got) same = same && v == acc;
            h2v::NodeStream node_rlc(blob.data(), blob.size(), {0, 0, 0}, n, 1, true);
            same = same && node_rlc.verify(batch) == acc;
            std::printf("node_stream %d\n", same ? 1 : 0);
            h2v::Workspace laned(vk, n, 3, 2);     // three lanes, chunks of two proofs
            std::printf("laned %d\n", h2v::verify_batch(vk, batch, laned.handle()) == acc ? 1 : 0);
            h2v::VerifyingKey vk2(blob.data(), blob.size(), 0);   // a second key (the same circuit loaded again) on one workspace for both
            h2v::Workspace multi({&vk, &vk2}, n, 2, 3);
            multi.submit(vk2, batch);
            bool both = multi.wait() == acc;
            multi.submit(batch, true);
            both = both && multi.wait() == acc;
            std::printf("multi %d\n", both ? 1 : 0);
        }
        // a consumed guard must refuse a second use; a wrong instance count must be refused by prepare
        bool refused = false;
        try {
            h2v::CircuitTranscript t = h2v::CircuitTranscript::init_from_bytes(std::vector<uint8_t>(vk.proof_len()));
            (void)h2v::prepare(vk, {}, std::vector<std::vector<uint8_t>>(n_pi + 1, std::vector<uint8_t>(32)), t);
        } catch (const h2v::Error &e) { refused = e.code == H2V_E_ARG; }
        std::printf("misuse_refused %d\n", refused ? 1 : 0);
        return 0;
    } catch (const h2v::Error &e) {
        std::printf("error %d %s\n", e.code, e.what());
        return 2;
    }
}
