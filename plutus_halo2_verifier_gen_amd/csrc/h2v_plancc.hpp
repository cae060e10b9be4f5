// The plan compiler behind the C-ABI (h2v_plan_compile): verifying-key description (JSON, docs/vk_schema.json "h2v-vk/1")
// -> plan blob, on the host, in C++ - the counterpart of the reference's plan-time layer `extract_circuit`
// (/root/reference/src/plutus_gen/extraction/mod.rs:31-232 with extraction/pcs/mod.rs:36-109 and
// extraction/data/extraction_steps/proof.rs:13-143), which a Rust host reaches without leaving its process.  It emits,
// byte for byte, what plutus_halo2_verifier_gen_amd/plan.py: compile_plan(vk).to_bytes() emits (tests/test_plan_compile.py
// compares the two on every built-in circuit and on a fuzzed family of shapes); plan.py stays the readable statement of
// the construction, with the reference line numbers beside every block - the comments here only name the block.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <errno.h>
#include <stdlib.h>

#include <algorithm>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "h2v_hostmath.hpp"
#include "h2v_plan.h"

namespace h2vplan {
using namespace h2vhost;

struct CompileError : std::runtime_error { using std::runtime_error::runtime_error; };

// ------------------------------------------------------------------------------------------------ a small JSON reader
struct JVal {
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    bool b = false;
    std::string s;                       // STR: the text; NUM: the literal (integers of any size)
    std::vector<JVal> a;
    std::vector<std::pair<std::string, JVal>> o;
    const JVal *get(const char *key) const {
        for (const auto &kv : o) if (kv.first == key) return &kv.second;
        return nullptr;
    }
};
struct JParser {
    const char *p, *end;
    int depth = 0;
    JParser(const char *s, size_t n) : p(s), end(s + n) {}
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    [[noreturn]] void bad(const char *what) { throw CompileError(std::string("JSON: ") + what); }
    JVal value() {
        if (++depth > 4096) bad("nesting too deep");
        ws();
        if (p >= end) bad("unexpected end");
        JVal v;
        if (*p == '{') {
            v.kind = JVal::OBJ; p++; ws();
            if (p < end && *p == '}') { p++; depth--; return v; }
            for (;;) {
                ws();
                JVal k = value();
                if (k.kind != JVal::STR) bad("object key is not a string");
                ws();
                if (p >= end || *p != ':') bad("':' expected");
                p++;
                v.o.emplace_back(k.s, value());
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == '}') { p++; break; }
                bad("',' or '}' expected");
            }
        } else if (*p == '[') {
            v.kind = JVal::ARR; p++; ws();
            if (p < end && *p == ']') { p++; depth--; return v; }
            for (;;) {
                v.a.push_back(value());
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') { p++; break; }
                bad("',' or ']' expected");
            }
        } else if (*p == '"') {
            v.kind = JVal::STR; p++;
            while (p < end && *p != '"') {
                if (*p == '\\') {
                    p++;
                    if (p >= end) bad("bad escape");
                    const char c = *p++;
                    if (c == 'n') v.s.push_back('\n');
                    else if (c == 't') v.s.push_back('\t');
                    else if (c == 'u') {   // (names only: keep ASCII, replace the rest)
                        if (end - p < 4) bad("bad \\u escape");
                        unsigned cp = 0;
                        for (int i = 0; i < 4; i++) { const int h = hex_nibble(p[i]); if (h < 0) bad("bad \\u escape"); cp = cp * 16 + h; }
                        p += 4;
                        v.s.push_back(cp < 128 ? (char)cp : '?');
                    } else v.s.push_back(c);
                } else v.s.push_back(*p++);
            }
            if (p >= end) bad("unterminated string");
            p++;
        } else if (*p == 't' && end - p >= 4 && !strncmp(p, "true", 4)) { v.kind = JVal::BOOL; v.b = true; p += 4; }
        else if (*p == 'f' && end - p >= 5 && !strncmp(p, "false", 5)) { v.kind = JVal::BOOL; p += 5; }
        else if (*p == 'n' && end - p >= 4 && !strncmp(p, "null", 4)) { p += 4; }
        else if (*p == '-' || (*p >= '0' && *p <= '9')) {
            v.kind = JVal::NUM;
            while (p < end && (*p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E' || (*p >= '0' && *p <= '9'))) v.s.push_back(*p++);
        } else bad("unexpected character");
        depth--;
        return v;
    }
};
static inline int64_t jint(const JVal *v, const char *what, int64_t lo, int64_t hi) {
    if (!v || v->kind != JVal::NUM) throw CompileError(std::string(what) + ": an integer is required");
    errno = 0;
    char *e = nullptr;
    const long long x = strtoll(v->s.c_str(), &e, 10);
    if (errno || *e || x < lo || x > hi) throw CompileError(std::string(what) + ": out of range");
    return x;
}
static inline U256 jscalar(const JVal *v, const char *what) {   // a canonical Fr value
    U256 x;
    if (!v || v->kind != JVal::NUM || !u256_from_decimal(v->s, x) || !FR().reduced(x)) throw CompileError(std::string(what) + ": not a canonical scalar");
    return x;
}
static inline const std::vector<JVal> &jarr(const JVal *v, const char *what) {
    if (!v || v->kind != JVal::ARR) throw CompileError(std::string(what) + ": an array is required");
    return v->a;
}

// ------------------------------------------------------------------------------------------------ the key description
enum { ROT_LAST = 1 << 30 };
struct Expr {   // Expression<Scalar> as transpiled at languages/aiken.rs:122-182
    enum Tag { CONST, FIXED, ADVICE, NEG, SUM, PROD, SCALED } tag;
    U256 c;         // CONST value / SCALED factor (canonical)
    int idx = 0;    // query index
    int a = -1, b = -1;
};
struct VK {
    std::string name;
    int k = 0, bf = 0, degree = 0, n_adv = 0, n_fix = 0, n_pi = 0, n_ci = 0;
    U256 transcript_repr;
    std::vector<std::pair<int, int>> aq, fq, iq;
    std::vector<Expr> pool;
    std::vector<int> gates;
    std::vector<std::pair<std::vector<int>, std::vector<int>>> lookups;
    std::vector<std::pair<int, std::vector<int>>> trashcans;
    std::vector<std::pair<int, int>> perm_cols;      // (0 advice / 1 fixed / 2 instance, column)
    std::vector<std::string> fixed_comm, perm_comm;
    std::string s_g2;
    bool recursive = false;
    struct Inner { std::string name; U256 transcript_repr; std::vector<std::string> fixed_comm, perm_comm; };
    std::vector<Inner> inner;
    std::vector<int> adv_phase, chal_phase;
    int chunk_len() const { return degree - 2; }
    int n_chunks() const { return ((int)perm_cols.size() + chunk_len() - 1) / chunk_len(); }
};
static int parse_expr(VK &vk, const JVal &v, const std::string &where, int depth = 0) {
    if (depth > 2000) throw CompileError(where + ": expression too deep");
    if (v.kind != JVal::ARR || v.a.empty() || v.a[0].kind != JVal::STR) throw CompileError(where + ": not an expression node");
    const std::string &t = v.a[0].s;
    Expr e;
    if (t == "selector" || t == "instance" || t == "challenge")
        throw CompileError(where + ": " + t + " nodes are not supported (the reference panics here: languages/aiken.rs:134-156)");
    if (t == "const") { if (v.a.size() != 2) throw CompileError(where + ": const takes one value"); e.tag = Expr::CONST; e.c = jscalar(&v.a[1], where.c_str()); }
    else if (t == "fixed" || t == "advice") {
        if (v.a.size() != 2) throw CompileError(where + ": query node takes one index");
        e.tag = t == "fixed" ? Expr::FIXED : Expr::ADVICE;
        const size_t n = t == "fixed" ? vk.fq.size() : vk.aq.size();
        e.idx = (int)jint(&v.a[1], where.c_str(), 0, (int64_t)n - 1);
    } else if (t == "neg") { if (v.a.size() != 2) throw CompileError(where + ": neg takes one operand"); e.tag = Expr::NEG; e.a = parse_expr(vk, v.a[1], where, depth + 1); }
    else if (t == "scaled") {
        if (v.a.size() != 3) throw CompileError(where + ": scaled takes an expression and a scalar");
        e.tag = Expr::SCALED; e.a = parse_expr(vk, v.a[1], where, depth + 1); e.c = jscalar(&v.a[2], where.c_str());
    } else if (t == "sum" || t == "prod") {
        if (v.a.size() != 3) throw CompileError(where + ": " + t + " takes two operands");
        e.tag = t == "sum" ? Expr::SUM : Expr::PROD;
        e.a = parse_expr(vk, v.a[1], where, depth + 1); e.b = parse_expr(vk, v.a[2], where, depth + 1);
    } else throw CompileError(where + ": unknown expression node " + t);
    vk.pool.push_back(e);
    return (int)vk.pool.size() - 1;
}
static void parse_queries(const JVal *v, const char *what, int ncols, std::vector<std::pair<int, int>> &out) {
    for (const JVal &q : jarr(v, what)) {
        if (q.kind != JVal::ARR || q.a.size() != 2) throw CompileError(std::string(what) + ": (column, rotation) pairs");
        const std::pair<int, int> pr((int)jint(&q.a[0], what, 0, ncols - 1), (int)jint(&q.a[1], what, -(1 << 20), 1 << 20));
        if (std::find(out.begin(), out.end(), pr) != out.end()) throw CompileError(std::string("duplicate ") + what + " query");
        out.push_back(pr);
    }
}
static void parse_hex_list(const JVal *v, const char *what, size_t bytes, std::vector<std::string> &out) {
    for (const JVal &h : jarr(v, what)) {
        std::vector<uint8_t> raw;
        if (h.kind != JVal::STR || !hex_to_bytes(h.s, raw) || raw.size() != bytes) throw CompileError(std::string(what) + ": compressed points as hex");
        out.push_back(h.s);
    }
}
static VK parse_vk(const char *json, size_t len) {
    JParser jp(json, len);
    const JVal root = jp.value();
    jp.ws();
    if (jp.p != jp.end) throw CompileError("JSON: trailing characters");
    if (root.kind != JVal::OBJ) throw CompileError("the verifying-key description is a JSON object");
    static const char *known[] = {"schema_version", "name", "k", "blinding_factors", "cs_degree", "transcript_repr", "num_advice_columns",
                                  "num_fixed_columns", "advice_queries", "fixed_queries", "instance_queries", "gates", "lookups", "trashcans",
                                  "permutation_columns", "fixed_commitments", "permutation_commitments", "s_g2", "n_public_inputs",
                                  "n_committed_instances", "recursion_vks", "advice_column_phase", "challenge_phase"};
    for (const auto &kv : root.o) {
        bool ok = false;
        for (const char *k : known) ok = ok || kv.first == k;
        if (!ok) throw CompileError("unknown field in the verifying-key description: " + kv.first);
    }
    VK vk;
    if (root.get("schema_version") && jint(root.get("schema_version"), "schema_version", 0, 1 << 30) != 1) throw CompileError("unsupported schema_version (this build reads 1)");
    const JVal *nm = root.get("name");
    if (!nm || nm->kind != JVal::STR) throw CompileError("name: a string is required");
    vk.name = nm->s;
    vk.k = (int)jint(root.get("k"), "k", 1, 32);
    vk.bf = (int)jint(root.get("blinding_factors"), "blinding_factors", 0, 1 << 16);
    vk.degree = (int)jint(root.get("cs_degree"), "cs_degree", 3, 1 << 10);
    vk.transcript_repr = jscalar(root.get("transcript_repr"), "transcript_repr");
    vk.n_adv = (int)jint(root.get("num_advice_columns"), "num_advice_columns", 0, 4096);
    vk.n_fix = (int)jint(root.get("num_fixed_columns"), "num_fixed_columns", 0, 4096);
    vk.n_pi = (int)jint(root.get("n_public_inputs"), "n_public_inputs", 0, 1 << 16);
    vk.n_ci = (int)jint(root.get("n_committed_instances"), "n_committed_instances", 0, 1);
    parse_queries(root.get("advice_queries"), "advice", vk.n_adv, vk.aq);
    parse_queries(root.get("fixed_queries"), "fixed", vk.n_fix, vk.fq);
    parse_queries(root.get("instance_queries"), "instance", vk.n_ci + 1, vk.iq);
    int gi = 0;
    for (const JVal &g : jarr(root.get("gates"), "gates")) vk.gates.push_back(parse_expr(vk, g, "gate polynomial " + std::to_string(gi++)));
    int li = 0;
    for (const JVal &lk : jarr(root.get("lookups"), "lookups")) {
        const std::string where = "lookup " + std::to_string(li++);
        if (lk.kind != JVal::ARR || lk.a.size() != 2) throw CompileError(where + ": (input expressions, table expressions)");
        std::vector<int> ins, tabs;
        for (const JVal &e : jarr(&lk.a[0], where.c_str())) ins.push_back(parse_expr(vk, e, where));
        for (const JVal &e : jarr(&lk.a[1], where.c_str())) tabs.push_back(parse_expr(vk, e, where));
        if (ins.empty() || ins.size() != tabs.size()) throw CompileError(where + ": input and table expression lists must be non-empty and of equal length");
        vk.lookups.emplace_back(ins, tabs);
    }
    int ti = 0;
    for (const JVal &tc : jarr(root.get("trashcans"), "trashcans")) {
        const std::string where = "trashcan " + std::to_string(ti++);
        if (tc.kind != JVal::ARR || tc.a.size() != 2) throw CompileError(where + ": (selector, constraint expressions)");
        const int sel = parse_expr(vk, tc.a[0], where);
        std::vector<int> cons;
        for (const JVal &e : jarr(&tc.a[1], where.c_str())) cons.push_back(parse_expr(vk, e, where));
        vk.trashcans.emplace_back(sel, cons);
    }
    for (const JVal &pc : jarr(root.get("permutation_columns"), "permutation_columns")) {
        if (pc.kind != JVal::ARR || pc.a.size() != 2 || pc.a[0].kind != JVal::STR) throw CompileError("permutation_columns: (type, column) pairs");
        const int ty = pc.a[0].s == "advice" ? 0 : pc.a[0].s == "fixed" ? 1 : pc.a[0].s == "instance" ? 2 : -1;
        if (ty < 0) throw CompileError("permutation column of unknown type " + pc.a[0].s);
        const int col = (int)jint(&pc.a[1], "permutation column", 0, 1 << 20);
        const auto &qs = ty == 0 ? vk.aq : ty == 1 ? vk.fq : vk.iq;
        if (std::find(qs.begin(), qs.end(), std::make_pair(col, 0)) == qs.end())
            throw CompileError("permutation column " + pc.a[0].s + "[" + std::to_string(col) + "] has no query at the current rotation");
        vk.perm_cols.emplace_back(ty, col);
    }
    parse_hex_list(root.get("fixed_commitments"), "fixed_commitments", 48, vk.fixed_comm);
    parse_hex_list(root.get("permutation_commitments"), "permutation_commitments", 48, vk.perm_comm);
    if ((int)vk.fixed_comm.size() != vk.n_fix) throw CompileError("one fixed commitment per fixed column");
    if (vk.perm_comm.size() != vk.perm_cols.size() || vk.perm_cols.empty()) throw CompileError("one permutation commitment per permutation column (and at least one)");
    const JVal *sg = root.get("s_g2");
    std::vector<uint8_t> raw;
    if (!sg || sg->kind != JVal::STR || !hex_to_bytes(sg->s, raw) || raw.size() != 96) throw CompileError("s_g2 is a 96-byte compressed G2 point");
    vk.s_g2 = sg->s;
    const JVal *rec = root.get("recursion_vks");
    if (rec && rec->kind != JVal::NUL) {
        vk.recursive = true;
        for (const JVal &iv : jarr(rec, "recursion_vks")) {
            if (iv.kind != JVal::OBJ || iv.o.size() != 4) throw CompileError("inner verifying key: fields name / transcript_repr / fixed_commitments / permutation_commitments");
            VK::Inner in;
            const JVal *n2 = iv.get("name");
            if (!n2 || n2->kind != JVal::STR) throw CompileError("inner verifying key: name");
            in.name = n2->s;
            in.transcript_repr = jscalar(iv.get("transcript_repr"), "inner transcript_repr");
            parse_hex_list(iv.get("fixed_commitments"), "inner fixed_commitments", 48, in.fixed_comm);
            parse_hex_list(iv.get("permutation_commitments"), "inner permutation_commitments", 48, in.perm_comm);
            vk.inner.push_back(in);
        }
    }
    const JVal *ap = root.get("advice_column_phase");
    if (ap && ap->kind != JVal::NUL) {
        for (const JVal &x : jarr(ap, "advice_column_phase")) vk.adv_phase.push_back((int)jint(&x, "advice_column_phase", 0, 255));
        if ((int)vk.adv_phase.size() != vk.n_adv) throw CompileError("advice_column_phase: one phase (0..255) per advice column");
    } else vk.adv_phase.assign(vk.n_adv, 0);
    const JVal *cp = root.get("challenge_phase");
    if (cp && cp->kind != JVal::NUL) {
        int top = 0;
        for (int x : vk.adv_phase) top = std::max(top, x);
        for (const JVal &x : jarr(cp, "challenge_phase")) {
            vk.chal_phase.push_back((int)jint(&x, "challenge_phase", 0, 255));
            if (vk.chal_phase.back() > top) throw CompileError("a challenge of a phase beyond the last advice phase is never squeezed (proof.rs:24-29)");
        }
    }
    return vk;
}

// ------------------------------------------------------------------------------------------------ the program builder
typedef std::array<int, 4> Ins;   // op, dst, a, b on virtual registers
struct U256Less { bool operator()(const U256 &a, const U256 &b) const { return a < b; } };
struct Builder {
    std::vector<Ins> code;
    int n_virt = 0;
    std::vector<U256> consts;                        // canonical values
    std::map<U256, int, U256Less> const_idx, const_reg;
    std::map<std::tuple<int, int, int>, int> cse;
    std::map<int, int> not_before;
    int fresh() { return n_virt++; }
    void emit(int op, int dst = 0, int a = 0, int b = 0) { code.push_back(Ins{op, dst, a, b}); }
    int constant(const U256 &v) {      // v canonical
        auto it = const_reg.find(v);
        if (it != const_reg.end()) return it->second;
        if (!const_idx.count(v)) { const_idx[v] = (int)consts.size(); consts.push_back(v); }
        const int r = fresh();
        emit(H2V_OP_CONST, r, const_idx[v], 0);
        const_reg[v] = r;
        return r;
    }
    int constant_u64(uint64_t v) { return constant(U256(v)); }
    int pure(int op, int a, int b, bool commutative) {
        const auto key = commutative ? std::make_tuple(op, std::min(a, b), std::max(a, b)) : std::make_tuple(op, a, b);
        auto it = cse.find(key);
        if (it != cse.end()) return it->second;
        const int r = fresh();
        emit(op, r, a, b);
        cse[key] = r;
        return r;
    }
    int add(int a, int b) { return pure(H2V_OP_ADD, a, b, true); }
    int sub(int a, int b) { return pure(H2V_OP_SUB, a, b, false); }
    int mul(int a, int b) { return pure(H2V_OP_MUL, a, b, true); }
    int neg(int a) { return pure(H2V_OP_NEG, a, 0, false); }
    int inv(int a) { return pure(H2V_OP_INV, a, 0, false); }
    std::vector<int> batch_inverse(const std::vector<int> &regs) {
        if (regs.empty()) return {};
        std::vector<int> pre{regs[0]};
        for (size_t i = 1; i < regs.size(); i++) pre.push_back(mul(pre.back(), regs[i]));
        int iv = inv(pre.back());
        std::vector<int> out(regs.size());
        for (size_t i = regs.size() - 1; i > 0; i--) {
            out[i] = mul(iv, pre[i - 1]);
            iv = mul(iv, regs[i]);
        }
        out[0] = iv;
        return out;
    }
};

// Fr values of the key: canonical in, canonical out (Montgomery inside)
static inline U256 fr_mulc(const U256 &a, const U256 &b) { return FR().from_mont(FR().mul(FR().to_mont(a), FR().to_mont(b))); }
static inline U256 fr_subc(const U256 &a, const U256 &b) { return FR().sub(a, b); }
static inline U256 fr_powc(const U256 &a, uint64_t e) { return FR().from_mont(FR().pow(FR().to_mont(a), UInt<1>(e))); }
static inline U256 fr_invc(const U256 &a) { return FR().from_mont(FR().inv(FR().to_mont(a))); }
static U256 domain_omega(int k) {   // ROOT_OF_UNITY = 7^((r - 1) >> 32), squared 32 - k times (vk.py: domain_omega)
    U256 e = FR().p;
    U256 one(1);
    sub_from(e, one);
    for (int i = 0; i < 4; i++) e.w[i] = i + 1 < 4 ? (e.w[i] >> 32) | (e.w[i + 1] << 32) : e.w[i] >> 32;
    U256 w = FR().from_mont(FR().pow(FR().from_u64(7), e));
    for (int i = 0; i < 32 - k; i++) w = fr_mulc(w, w);
    return w;
}
static const U256 &delta_const() {
    static const U256 d = from_hex<4>("08634d0aa021aaf843cab354fabb0062f6502437c6a09c006c083479590189d7");
    return d;
}

// ------------------------------------------------------------------------------------------------ scheduling (plan.py: _schedule ...)
static inline bool is_transcript_op(int op) {
    return op == H2V_OP_ABSORB_REG || op == H2V_OP_ABSORB_CI || op == H2V_OP_READ_POINT || op == H2V_OP_READ_SCALAR || op == H2V_OP_SQUEEZE;
}
static inline int uses_of(const Ins &i, int (&u)[2]) {
    const int op = i[0];
    if (op == H2V_OP_ADD || op == H2V_OP_SUB || op == H2V_OP_MUL) { u[0] = i[2]; u[1] = i[3]; return 2; }
    if (op == H2V_OP_NEG || op == H2V_OP_INV || op == H2V_OP_ABSORB_REG || op == H2V_OP_OUT_SCALAR || op == H2V_OP_ASSERT_ZERO) { u[0] = i[2]; return 1; }
    return 0;
}
static inline bool defines(int op) {
    return op == H2V_OP_LOAD_INSTANCE || op == H2V_OP_READ_SCALAR || op == H2V_OP_SQUEEZE || op == H2V_OP_CONST || op == H2V_OP_ADD ||
           op == H2V_OP_SUB || op == H2V_OP_MUL || op == H2V_OP_NEG || op == H2V_OP_INV;
}
struct Rec { bool some = false; Ins ins{}; };
typedef std::vector<std::vector<Rec>> Bundles;
static Bundles schedule(const std::vector<Ins> &code, int lanes, bool pack_mul, const std::map<int, int> &not_before, int n_virt) {
    std::vector<int> slot_of(n_virt, -1);
    Bundles bundles;
    std::vector<char> exclusive, has_mul;
    std::vector<int> n_free;
    size_t first_open = 0;
    for (const Ins &ins : code) {
        const int op = ins[0], dst = ins[1];
        if (op == H2V_OP_END) continue;
        int e = 0, u[2];
        const int nu = uses_of(ins, u);
        for (int q = 0; q < nu; q++) e = std::max(e, slot_of[u[q]] + 1);
        if (op == H2V_OP_LOAD_INSTANCE) {
            auto it = not_before.find(dst);
            if (it != not_before.end()) e = std::max(e, slot_of[it->second]);
        }
        int s;
        if (is_transcript_op(op)) {
            s = (int)bundles.size();
            bundles.emplace_back(lanes);
            bundles.back()[0].some = true; bundles.back()[0].ins = ins;
            exclusive.push_back(1); has_mul.push_back(0); n_free.push_back(lanes - 1);
        } else {
            s = -1;
            const size_t start = std::max((size_t)e, first_open);
            if (op == H2V_OP_MUL && pack_mul)
                for (size_t k = start; k < bundles.size(); k++)
                    if (has_mul[k] && !exclusive[k] && n_free[k] > 0) { s = (int)k; break; }
            if (s < 0)
                for (size_t k = start; k < bundles.size(); k++)
                    if (!exclusive[k] && n_free[k] > 0) { s = (int)k; break; }
            if (s < 0) {
                s = (int)bundles.size();
                bundles.emplace_back(lanes);
                exclusive.push_back(0); has_mul.push_back(0); n_free.push_back(lanes);
            }
            for (Rec &r : bundles[s]) if (!r.some) { r.some = true; r.ins = ins; break; }
            n_free[s]--;
            has_mul[s] = has_mul[s] || op == H2V_OP_MUL;
            while (first_open < bundles.size() && (exclusive[first_open] || n_free[first_open] == 0)) first_open++;
        }
        if (defines(op)) slot_of[dst] = s;
    }
    bundles.emplace_back(lanes);
    bundles.back()[0].some = true; bundles.back()[0].ins = Ins{H2V_OP_END, 0, 0, 0};
    return bundles;
}
static double schedule_cost(const Bundles &bundles) {
    double total = 0.0;
    for (const auto &bun : bundles) {
        std::set<int> ops;
        for (const Rec &r : bun) if (r.some) ops.insert(r.ins[0]);
        double t = 0.05;
        for (int op : ops) {   // ascending: plan.py sums in sorted order
            if (op == H2V_OP_MUL) t += 1.0;
            else if (op == H2V_OP_INV) t += 14.0;
            else if (op == H2V_OP_ABSORB_REG) t += 1.6;
            else if (op == H2V_OP_ABSORB_CI) t += 0.8;
            else if (op == H2V_OP_READ_POINT) t += 0.8;
            else if (op == H2V_OP_READ_SCALAR) t += 1.7;
            else if (op == H2V_OP_SQUEEZE) t += 3.5;
            else t += 0.06;
        }
        total += t;
    }
    return total;
}
struct Allocated { std::vector<Ins> instrs; int n_regs = 0; std::vector<int> mapping; };
static Allocated allocate(const Bundles &bundles, const std::vector<int> &keep_alive, int n_virt) {
    std::vector<int> last_use(n_virt, -1);
    for (size_t s = 0; s < bundles.size(); s++)
        for (const Rec &r : bundles[s])
            if (r.some) { int u[2]; const int nu = uses_of(r.ins, u); for (int q = 0; q < nu; q++) last_use[u[q]] = (int)s; }
    const int end = (int)bundles.size();
    for (int r : keep_alive) last_use[r] = end;
    std::vector<int> free_list;
    int n_phys = 0;
    Allocated out;
    out.mapping.assign(n_virt, -1);
    std::map<int, std::vector<int>> expiring;
    for (size_t s = 0; s < bundles.size(); s++) {
        auto it = expiring.find((int)s);
        if (it != expiring.end()) { for (int pr : it->second) free_list.push_back(pr); expiring.erase(it); }
        for (const Rec &r : bundles[s]) {
            if (!r.some) { out.instrs.push_back(Ins{H2V_OP_NOP, 0, 0, 0}); continue; }
            const int op = r.ins[0], dst = r.ins[1], a = r.ins[2], c = r.ins[3];
            int pa = a, pc = c, pd = dst;
            if (op == H2V_OP_ADD || op == H2V_OP_SUB || op == H2V_OP_MUL) { pa = out.mapping[a]; pc = out.mapping[c]; }
            else if (op == H2V_OP_NEG || op == H2V_OP_INV || op == H2V_OP_ABSORB_REG || op == H2V_OP_OUT_SCALAR || op == H2V_OP_ASSERT_ZERO) pa = out.mapping[a];
            if (defines(op)) {
                if (!free_list.empty()) { pd = free_list.back(); free_list.pop_back(); }
                else pd = n_phys++;
                out.mapping[dst] = pd;
                const int lu = std::max(last_use[dst] < 0 ? (int)s : last_use[dst], (int)s);
                if (lu < end) expiring[lu + 1].push_back(pd);
            }
            out.instrs.push_back(Ins{op, pd, pa, pc});
        }
    }
    if (n_phys >= 65536) throw CompileError("program needs more than 65535 registers");
    out.n_regs = std::max(n_phys, 1);
    return out;
}
struct Sched { double cost = 0; Allocated al; int lanes = 0; bool fits = false; };
static const size_t VM_LDS_BYTES = (size_t)160 * 1024 - 8192 - 1024;
static void schedule_and_allocate(const Builder &b, const std::vector<int> &keep_alive, Sched &narrow, bool &has_wide, Sched &wide) {
    static const int choices[] = {1, 2, 4, 8, 16};
    std::vector<Sched> cands;
    for (int L : choices) {
        Sched best;
        bool have = false;
        for (int pack = 0; pack < 2; pack++) {
            const Bundles bun = schedule(b.code, L, pack != 0, b.not_before, b.n_virt);
            const double cost = schedule_cost(bun);
            if (!have || cost < best.cost) { best.cost = cost; best.al = allocate(bun, keep_alive, b.n_virt); best.lanes = L; have = true; }
        }
        best.fits = (size_t)best.al.n_regs * 32 * (64 / L) <= VM_LDS_BYTES;
        cands.push_back(std::move(best));
    }
    std::vector<const Sched *> fitting;
    for (const Sched &c : cands) if (c.fits) fitting.push_back(&c);
    has_wide = false;
    if (fitting.empty()) { narrow = cands[0]; return; }
    narrow = *fitting[0];
    double floor_cost = fitting[0]->cost;
    for (const Sched *c : fitting) floor_cost = std::min(floor_cost, c->cost);
    const Sched *w = nullptr;
    for (const Sched *c : fitting) if (c->cost <= 1.1 * floor_cost) { w = c; break; }
    if (w && w->lanes > narrow.lanes && w->cost < narrow.cost * 0.9) { has_wide = true; wide = *w; }
}

// ------------------------------------------------------------------------------------------------ compile_plan
struct CKey { int kind, idx; bool operator==(const CKey &o) const { return kind == o.kind && idx == o.idx; } bool operator<(const CKey &o) const { return std::tie(kind, idx) < std::tie(o.kind, o.idx); } };
enum { CK_ADVICE, CK_INSTANCE, CK_PERM, CK_LOOKUP, CK_PERM_INPUT, CK_PERM_TABLE, CK_TRASH, CK_FIXED, CK_COMMON, CK_VANISHING_G, CK_VANISHING_RAND };
struct EKey { int kind, i, j; };
enum { EK_ADVICE, EK_INSTANCE, EK_FIXED, EK_PERM, EK_LK, EK_TRASH, EK_COMMON, EK_VANISHING_S, EK_RANDOM };
struct BaseKey { int kind; std::string name; int idx; bool operator<(const BaseKey &o) const { return std::tie(kind, name, idx) < std::tie(o.kind, o.name, o.idx); } };
enum { BK_NEG_G1, BK_FIXED, BK_COMMON, BK_INNER_F, BK_INNER_P };

static inline std::tuple<int, int> rot_sort_key(int rot) {
    if (rot == ROT_LAST) return {0, 0};
    if (rot == -1) return {1, 0};
    if (rot == 0) return {2, 0};
    if (rot == 1) return {3, 0};
    return {4, rot};
}
static void pad16(std::vector<uint8_t> &b) { while (b.size() % 16) b.push_back(0); }
static void put32(std::vector<uint8_t> &b, uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); }
static void put_instr(std::vector<uint8_t> &b, const Ins &i) {
    b.push_back((uint8_t)i[0]); b.push_back(0);
    for (int q = 1; q < 4; q++) { b.push_back((uint8_t)(i[q] & 0xff)); b.push_back((uint8_t)((i[q] >> 8) & 0xff)); }
}
static void put_f2(std::vector<uint8_t> &b, const F2 &v) {
    uint8_t t[48];
    fp_mont392_bytes(v.a, t); b.insert(b.end(), t, t + 48);
    fp_mont392_bytes(v.b, t); b.insert(b.end(), t, t + 48);
}
static void put_slot28(std::vector<uint8_t> &b, const FpE &v) {   // bls12_381.py: fp_mont28_slot
    uint8_t t[48];
    fp_mont392_bytes(v, t);
    U384 m;
    for (int i = 0; i < 48; i++) m.w[i / 8] |= (uint64_t)t[i] << (8 * (i % 8));
    for (int i = 0; i < 14; i++) {
        const int bit = 28 * i;
        uint64_t limb = m.w[bit / 64] >> (bit % 64);
        if (bit % 64 > 36 && bit / 64 + 1 < 6) limb |= m.w[bit / 64 + 1] << (64 - bit % 64);
        put32(b, (uint32_t)(limb & 0xfffffffu));
    }
    for (int i = 0; i < 8; i++) b.push_back(0);
}
struct Line { F2 lam, c; };
static std::vector<Line> g2_line_table(const G2 &q) {   // bls12_381.py: g2_line_table
    if (q.inf) throw CompileError("G2 argument of the pairing must not be infinity");
    const uint64_t x_abs = 0xd201000000010000ull;
    std::vector<Line> table;
    G2 t = q;
    for (int bit = 62; bit >= 0; bit--) {
        F2 lam = f2_mul(f2_scale(f2_sqr(t.x), 3), f2_inv(f2_scale(t.y, 2)));
        table.push_back(Line{lam, f2_sub(f2_mul(lam, t.x), t.y)});
        t = pt_add(t, t);
        if ((x_abs >> bit) & 1) {
            lam = f2_mul(f2_sub(q.y, t.y), f2_inv(f2_sub(q.x, t.x)));
            table.push_back(Line{lam, f2_sub(f2_mul(lam, t.x), t.y)});
            t = pt_add(t, q);
        }
    }
    return table;
}

static std::vector<uint8_t> compile_plan(const VK &vk) {
    Builder b;
    const int L = (int)vk.lookups.size(), Cn = vk.n_chunks(), n_trash = (int)vk.trashcans.size(), n_splits = vk.degree - 1, n_ci = vk.n_ci;
    const U256 omega = domain_omega(vk.k), omega_inv = fr_invc(omega);
    U256 n_val(1);
    n_val.w[0] = 1ull << vk.k;
    const U256 bary = fr_invc(n_val);
    auto rot_value = [&](int rot) {
        const int64_t n = rot == ROT_LAST ? -(int64_t)(vk.bf + 1) : rot;
        return n >= 0 ? fr_powc(omega, (uint64_t)n) : fr_powc(omega_inv, (uint64_t)(-n));
    };
    uint32_t pos = 0, n_squeezes = 0, stream_len = 0;
    std::vector<uint32_t> points;
    auto read_point = [&]() {
        const int idx = (int)points.size();
        points.push_back(pos);
        b.emit(H2V_OP_READ_POINT, 0, pos & 0xffff, pos >> 16);
        pos += 48; stream_len += 49;
        return idx;
    };
    auto read_scalar = [&]() {
        const int r = b.fresh();
        b.emit(H2V_OP_READ_SCALAR, r, pos & 0xffff, pos >> 16);
        pos += 32; stream_len += 33;
        return r;
    };
    auto squeeze = [&]() {
        const int r = b.fresh();
        b.emit(H2V_OP_SQUEEZE, r);
        n_squeezes++; stream_len += 1;
        return r;
    };
    auto absorb = [&](int reg) { b.emit(H2V_OP_ABSORB_REG, 0, reg, 0); stream_len += 33; };
    const int one = b.constant_u64(1), zero = b.constant_u64(0);
    // ---- P1
    absorb(b.constant(vk.transcript_repr));
    if (n_ci) { b.emit(H2V_OP_ABSORB_CI); stream_len += 49; }
    absorb(b.constant_u64((uint64_t)vk.n_pi));
    std::vector<int> pis;
    for (int k = 0; k < vk.n_pi; k++) {
        const int r = b.fresh();
        b.emit(H2V_OP_LOAD_INSTANCE, r, k, 0);
        absorb(r);
        pis.push_back(r);
    }
    // ---- P2
    std::vector<int> adv_pts(vk.n_adv, -1);
    int max_phase = 0;
    for (int x : vk.adv_phase) max_phase = std::max(max_phase, x);
    for (int phase = 0; phase <= max_phase; phase++) {
        for (int i = 0; i < vk.n_adv; i++) if (vk.adv_phase[i] == phase) adv_pts[i] = read_point();
        for (int ph : vk.chal_phase) if (ph == phase) squeeze();
    }
    const int theta = squeeze();
    std::vector<int> lk_pin, lk_ptab;
    for (int i = 0; i < L; i++) { lk_pin.push_back(read_point()); lk_ptab.push_back(read_point()); }
    const int beta = squeeze(), gamma = squeeze();
    std::vector<int> perm_pts, lk_prod, trash_pts, split_pts;
    for (int i = 0; i < Cn; i++) perm_pts.push_back(read_point());
    for (int i = 0; i < L; i++) lk_prod.push_back(read_point());
    const int trash = squeeze();
    for (int i = 0; i < n_trash; i++) trash_pts.push_back(read_point());
    const int vanish_rand_pt = read_point();
    const int y = squeeze();
    for (int i = 0; i < n_splits; i++) split_pts.push_back(read_point());
    const int x = squeeze();
    int acc = one;
    for (int i = 0; i < vk.k; i++) acc = b.mul(b.mul(acc, acc), x);
    const int xn_minus_one = acc, xn = b.mul(xn_minus_one, x);
    std::vector<int> instance_eval;   // -1 = computed later
    for (const auto &q : vk.iq) instance_eval.push_back(q.first < n_ci ? read_scalar() : -1);
    std::vector<int> advice_eval, fixed_eval, perm_common, trash_eval;
    for (size_t i = 0; i < vk.aq.size(); i++) advice_eval.push_back(read_scalar());
    for (size_t i = 0; i < vk.fq.size(); i++) fixed_eval.push_back(read_scalar());
    const int random_eval = read_scalar();
    for (size_t i = 0; i < vk.perm_comm.size(); i++) perm_common.push_back(read_scalar());
    std::vector<std::array<int, 3>> perm_eval;
    for (int i = 0; i < Cn; i++) {
        std::array<int, 3> z{read_scalar(), read_scalar(), -1};
        if (i != Cn - 1) z[2] = read_scalar();
        perm_eval.push_back(z);
    }
    std::vector<std::array<int, 5>> lk_eval;
    for (int i = 0; i < L; i++) { std::array<int, 5> e; for (int q = 0; q < 5; q++) e[q] = read_scalar(); lk_eval.push_back(e); }
    for (int i = 0; i < n_trash; i++) trash_eval.push_back(read_scalar());
    // ---- queries -> commitment map -> point sets
    struct Query { CKey ck; EKey ek; int rot; };
    std::vector<Query> queries;
    for (size_t qi = 0; qi < vk.aq.size(); qi++) queries.push_back({{CK_ADVICE, vk.aq[qi].first}, {EK_ADVICE, (int)qi, 0}, vk.aq[qi].second});
    for (size_t qi = 0; qi < vk.iq.size(); qi++) if (vk.iq[qi].first < n_ci) queries.push_back({{CK_INSTANCE, vk.iq[qi].first}, {EK_INSTANCE, (int)qi, 0}, vk.iq[qi].second});
    for (int i = 0; i < Cn; i++) { queries.push_back({{CK_PERM, i}, {EK_PERM, i, 0}, 0}); queries.push_back({{CK_PERM, i}, {EK_PERM, i, 1}, 1}); }
    for (int i = Cn - 2; i >= 0; i--) queries.push_back({{CK_PERM, i}, {EK_PERM, i, 2}, ROT_LAST});
    for (int i = 0; i < L; i++) {
        queries.push_back({{CK_LOOKUP, i}, {EK_LK, i, 0}, 0});
        queries.push_back({{CK_PERM_INPUT, i}, {EK_LK, i, 2}, 0});
        queries.push_back({{CK_PERM_TABLE, i}, {EK_LK, i, 4}, 0});
        queries.push_back({{CK_PERM_INPUT, i}, {EK_LK, i, 3}, -1});
        queries.push_back({{CK_LOOKUP, i}, {EK_LK, i, 1}, 1});
    }
    for (int i = 0; i < n_trash; i++) queries.push_back({{CK_TRASH, i}, {EK_TRASH, i, 0}, 0});
    for (size_t qi = 0; qi < vk.fq.size(); qi++) queries.push_back({{CK_FIXED, vk.fq[qi].first}, {EK_FIXED, (int)qi, 0}, vk.fq[qi].second});
    for (size_t i = 0; i < vk.perm_comm.size(); i++) queries.push_back({{CK_COMMON, (int)i}, {EK_COMMON, (int)i, 0}, 0});
    queries.push_back({{CK_VANISHING_G, 0}, {EK_VANISHING_S, 0, 0}, 0});
    queries.push_back({{CK_VANISHING_RAND, 0}, {EK_RANDOM, 0, 0}, 0});
    std::vector<CKey> commitments;
    std::map<CKey, std::vector<std::pair<int, EKey>>> cmap;
    for (const Query &q : queries) {
        if (!cmap.count(q.ck)) { cmap[q.ck]; commitments.push_back(q.ck); }
        cmap[q.ck].emplace_back(q.rot, q.ek);
    }
    for (const CKey &ck : commitments)
        std::stable_sort(cmap[ck].begin(), cmap[ck].end(), [](const std::pair<int, EKey> &a, const std::pair<int, EKey> &c) { return rot_sort_key(a.first) < rot_sort_key(c.first); });
    std::vector<std::vector<int>> uniq_sets;
    std::map<CKey, int> set_of;
    for (const CKey &ck : commitments) {
        std::vector<int> pts;
        for (const auto &pe : cmap[ck]) pts.push_back(pe.first);
        auto it = std::find(uniq_sets.begin(), uniq_sets.end(), pts);
        if (it == uniq_sets.end()) { uniq_sets.push_back(pts); it = uniq_sets.end() - 1; }
        set_of[ck] = (int)(it - uniq_sets.begin());
    }
    std::vector<int> sort_order(uniq_sets.size());
    for (size_t i = 0; i < sort_order.size(); i++) sort_order[i] = (int)i;
    std::stable_sort(sort_order.begin(), sort_order.end(), [&](int a, int c) { return uniq_sets[a].size() < uniq_sets[c].size(); });
    const int S = (int)uniq_sets.size();
    // ---- PCS tail
    const int x1 = squeeze(), x2 = squeeze();
    const int f_pt = read_point();
    const int x3 = squeeze();
    std::vector<int> q_evals;
    for (int s = 0; s < S; s++) q_evals.push_back(read_scalar());
    const int x4 = squeeze();
    const int pi_pt = read_point();
    const uint32_t proof_len = pos;
    auto rotated = [&](int rot) {
        const U256 w = rot_value(rot);
        return w == U256(1) ? x : b.mul(b.constant(w), x);
    };
    const int x_last = rotated(ROT_LAST);
    // ---- the single batch inversion
    std::vector<int> inv_in;
    const int bf = vk.bf;
    std::vector<U256> van_rot_w;
    for (int i = -(bf + 1); i <= 0; i++) van_rot_w.push_back(rot_value(i));
    std::vector<int> van_idx, pi_idx;
    for (const U256 &w : van_rot_w) { van_idx.push_back((int)inv_in.size()); inv_in.push_back(b.sub(x, b.constant(w))); }
    bool need_pi_basis = false;
    for (int e : instance_eval) need_pi_basis = need_pi_basis || e < 0;
    need_pi_basis = need_pi_basis && vk.n_pi > 0;
    std::vector<U256> pi_rot_w;
    if (need_pi_basis) for (int i = 0; i <= vk.n_pi; i++) pi_rot_w.push_back(fr_powc(omega, (uint64_t)i));
    for (const U256 &w : pi_rot_w) { pi_idx.push_back((int)inv_in.size()); inv_in.push_back(b.sub(x, b.constant(w))); }
    const int xn_m1 = b.sub(xn, one);
    const int xn_idx = (int)inv_in.size();
    inv_in.push_back(xn_m1);
    std::map<int, int> xpow{{0, one}, {1, x}};
    std::function<int(int)> x_power = [&](int e) {
        auto it = xpow.find(e);
        if (it != xpow.end()) return it->second;
        const int r = b.mul(x_power(e - 1), x);
        xpow[e] = r;
        return r;
    };
    std::vector<std::vector<int>> interp_idx;
    for (int s = 0; s < S; s++) {
        const std::vector<int> &pts = uniq_sets[sort_order[s]];
        std::vector<U256> ws;
        for (int pt : pts) ws.push_back(rot_value(pt));
        std::vector<int> idxs;
        for (size_t i = 0; i < pts.size(); i++) {
            U256 c(1);
            for (size_t j = 0; j < pts.size(); j++) if (j != i) c = fr_mulc(c, fr_subc(ws[i], ws[j]));
            idxs.push_back((int)inv_in.size());
            if (pts.size() > 1) {
                const int cr = b.constant(c);                    // (left to right, as plan.py evaluates its arguments:
                const int xp = x_power((int)pts.size() - 1);     //  C++ leaves the order of sibling arguments open)
                inv_in.push_back(b.mul(cr, xp));
            } else inv_in.push_back(one);
        }
        interp_idx.push_back(idxs);
    }
    std::vector<std::vector<int>> set_points;
    std::vector<int> fden_idx;
    for (int s = 0; s < S; s++) {
        std::vector<int> pts;
        for (int pt : uniq_sets[sort_order[s]]) pts.push_back(rotated(pt));
        set_points.push_back(pts);
        int d = one;
        for (int pt : pts) d = b.mul(d, b.sub(x3, pt));
        fden_idx.push_back((int)inv_in.size());
        inv_in.push_back(d);
    }
    const std::vector<int> inv_out = b.batch_inverse(inv_in);
    // ---- P3
    const int common = b.mul(xn_m1, b.constant(bary));
    std::vector<int> basis;
    for (int i = 0; i < bf + 2; i++) {
        const int t = b.mul(inv_out[van_idx[i]], common);
        basis.push_back(b.mul(t, b.constant(van_rot_w[i])));
    }
    const int l_last = basis[0], l_0 = basis.back();
    int sum_blind = zero;
    for (int i = 1; i < 1 + bf; i++) sum_blind = b.add(basis[i], sum_blind);
    const int active_rows = b.sub(one, b.add(l_last, sum_blind));
    int pub_eval = zero;
    if (need_pi_basis) {
        std::vector<int> pbasis;
        for (int i = 0; i < vk.n_pi; i++) {
            const int t = b.mul(inv_out[pi_idx[i]], common);
            pbasis.push_back(b.mul(t, b.constant(pi_rot_w[i])));
        }
        int a2 = zero;
        for (int i = 0; i < vk.n_pi; i++) {
            const int r = b.fresh();
            b.emit(H2V_OP_LOAD_INSTANCE, r, i, 0);
            b.not_before[r] = pbasis[i];
            a2 = b.add(b.mul(pbasis[i], r), a2);
        }
        pub_eval = a2;
    }
    for (int &e : instance_eval) if (e < 0) e = pub_eval;
    // ---- P4
    std::function<int(int)> ev = [&](int ei) -> int {
        const Expr &e = vk.pool[ei];
        switch (e.tag) {
        case Expr::CONST: return b.constant(e.c);
        case Expr::FIXED: return fixed_eval[e.idx];
        case Expr::ADVICE: return advice_eval[e.idx];
        case Expr::NEG: return b.neg(ev(e.a));
        case Expr::SUM: { const int l = ev(e.a); const int r = ev(e.b); return b.add(l, r); }
        case Expr::PROD: { const int l = ev(e.a); const int r = ev(e.b); return b.mul(l, r); }
        case Expr::SCALED: { const int l = ev(e.a); return b.mul(l, b.constant(e.c)); }
        }
        return zero;
    };
    auto compress = [&](const std::vector<int> &exprs, int ch) {
        int a2 = zero;
        for (int e : exprs) { const int m = b.mul(a2, ch); a2 = b.add(m, ev(e)); }
        return a2;
    };
    std::vector<int> expressions;
    for (int g : vk.gates) expressions.push_back(ev(g));
    expressions.push_back(b.mul(l_0, b.sub(one, perm_eval[0][0])));
    const int zl = perm_eval[Cn - 1][0];
    expressions.push_back(b.mul(l_last, b.sub(b.mul(zl, zl), zl)));
    for (int i = 1; i < Cn; i++) expressions.push_back(b.mul(b.sub(perm_eval[i][0], perm_eval[i - 1][2]), l_0));
    const int bx = b.mul(beta, x);
    auto find_query = [](const std::vector<std::pair<int, int>> &qs, int col) {
        for (size_t i = 0; i < qs.size(); i++) if (qs[i].first == col && qs[i].second == 0) return (int)i;
        throw CompileError("permutation column without a query at the current rotation");
    };
    auto column_eval = [&](int ty, int col) {
        if (ty == 0) return advice_eval[find_query(vk.aq, col)];
        if (ty == 1) return fixed_eval[find_query(vk.fq, col)];
        return instance_eval[find_query(vk.iq, col)];
    };
    const int not_blind = b.sub(one, b.add(l_last, sum_blind));
    for (int i = 0; i < Cn; i++) {
        int left = perm_eval[i][1], right = perm_eval[i][0];
        for (int idx = 0; idx < vk.chunk_len(); idx++) {
            const int col = i * vk.chunk_len() + idx;
            if (col >= (int)vk.perm_cols.size()) break;
            const int e = column_eval(vk.perm_cols[col].first, vk.perm_cols[col].second);
            left = b.mul(left, b.add(b.add(e, b.mul(beta, perm_common[col])), gamma));
            right = b.mul(right, b.add(b.add(e, b.mul(bx, b.constant(fr_powc(delta_const(), (uint64_t)col)))), gamma));
        }
        expressions.push_back(b.mul(b.sub(left, right), not_blind));
    }
    for (int i = 0; i < L; i++) {
        const int tab = compress(vk.lookups[i].second, theta);
        const int inp = compress(vk.lookups[i].first, theta);
        const int prod = lk_eval[i][0], prod_next = lk_eval[i][1], pin = lk_eval[i][2], pinv = lk_eval[i][3], ptab = lk_eval[i][4];
        expressions.push_back(b.mul(l_0, b.sub(one, prod)));
        expressions.push_back(b.mul(l_last, b.sub(b.mul(prod, prod), prod)));
        const int l1 = b.mul(prod_next, b.add(pin, beta));
        const int left = b.mul(l1, b.add(ptab, gamma));
        const int r1 = b.mul(prod, b.add(inp, beta));
        const int right = b.mul(r1, b.add(tab, gamma));
        expressions.push_back(b.mul(b.sub(left, right), active_rows));
        const int d = b.sub(pin, ptab);
        expressions.push_back(b.mul(l_0, d));
        expressions.push_back(b.mul(b.mul(d, b.sub(pin, pinv)), active_rows));
    }
    for (int i = 0; i < n_trash; i++) {
        const int lhs = compress(vk.trashcans[i].second, trash);
        expressions.push_back(b.sub(lhs, b.mul(b.sub(one, ev(vk.trashcans[i].first)), trash_eval[i])));
    }
    int h_eval = zero;
    for (int e : expressions) h_eval = b.add(b.mul(h_eval, y), e);
    const int vanishing_s = b.mul(h_eval, inv_out[xn_idx]);
    // ---- P6
    auto eval_reg = [&](const EKey &ek) {
        switch (ek.kind) {
        case EK_ADVICE: return advice_eval[ek.i];
        case EK_INSTANCE: return instance_eval[ek.i];
        case EK_FIXED: return fixed_eval[ek.i];
        case EK_PERM: return perm_eval[ek.i][ek.j];
        case EK_LK: return lk_eval[ek.i][ek.j];
        case EK_TRASH: return trash_eval[ek.i];
        case EK_COMMON: return perm_common[ek.i];
        case EK_VANISHING_S: return vanishing_s;
        default: return random_eval;
        }
    };
    std::vector<G1> vk_bases;
    std::map<BaseKey, int> vk_base_idx;
    auto vk_base = [&](const BaseKey &key, const G1 &pt) {
        auto it = vk_base_idx.find(key);
        if (it != vk_base_idx.end()) return it->second;
        vk_base_idx[key] = (int)vk_bases.size();
        vk_bases.push_back(pt);
        return (int)vk_bases.size() - 1;
    };
    auto decompress_hex = [](const std::string &h, bool check) {
        std::vector<uint8_t> raw;
        hex_to_bytes(h, raw);
        G1 pt;
        const std::string err = g1_decompress(raw.data(), pt, check);
        if (!err.empty()) throw CompileError(err);
        return pt;
    };
    std::vector<G1> fixed_pts, perm_cpts;
    for (const std::string &h : vk.fixed_comm) fixed_pts.push_back(decompress_hex(h, true));
    for (const std::string &h : vk.perm_comm) perm_cpts.push_back(decompress_hex(h, true));
    std::vector<std::pair<int, int>> terms;
    std::vector<int> term_scalar;
    auto add_term = [&](int kind, int index, int reg) { terms.emplace_back(kind, index); term_scalar.push_back(reg); };
    std::vector<std::vector<int>> q_eval_sets;
    int x4p = one;
    for (int s = 0; s < S; s++) {
        const int old = sort_order[s];
        const int m = (int)uniq_sets[old].size();
        std::vector<int> acc_evals(m, -1);
        int x1p = one;
        for (const CKey &ck : commitments) {
            if (set_of[ck] != old) continue;
            const int coeff = b.mul(x4p, x1p);
            switch (ck.kind) {
            case CK_ADVICE: add_term(H2V_TERM_PROOF_POINT, adv_pts[ck.idx], coeff); break;
            case CK_INSTANCE: add_term(H2V_TERM_COMMITTED_INSTANCE, 0, coeff); break;
            case CK_PERM: add_term(H2V_TERM_PROOF_POINT, perm_pts[ck.idx], coeff); break;
            case CK_LOOKUP: add_term(H2V_TERM_PROOF_POINT, lk_prod[ck.idx], coeff); break;
            case CK_PERM_INPUT: add_term(H2V_TERM_PROOF_POINT, lk_pin[ck.idx], coeff); break;
            case CK_PERM_TABLE: add_term(H2V_TERM_PROOF_POINT, lk_ptab[ck.idx], coeff); break;
            case CK_TRASH: add_term(H2V_TERM_PROOF_POINT, trash_pts[ck.idx], coeff); break;
            case CK_FIXED: add_term(H2V_TERM_VK_BASE, vk_base(BaseKey{BK_FIXED, "", ck.idx}, fixed_pts[ck.idx]), coeff); break;
            case CK_COMMON: add_term(H2V_TERM_VK_BASE, vk_base(BaseKey{BK_COMMON, "", ck.idx}, perm_cpts[ck.idx]), coeff); break;
            case CK_VANISHING_RAND: add_term(H2V_TERM_PROOF_POINT, vanish_rand_pt, coeff); break;
            case CK_VANISHING_G: {
                int c = coeff;
                for (int i = 0; i < n_splits; i++) { add_term(H2V_TERM_PROOF_POINT, split_pts[i], c); c = b.mul(c, xn_minus_one); }
            } break;
            }
            const auto &pairs = cmap[ck];
            for (size_t j = 0; j < pairs.size(); j++) {
                const int t = b.mul(eval_reg(pairs[j].second), x1p);
                acc_evals[j] = acc_evals[j] < 0 ? t : b.add(acc_evals[j], t);
            }
            x1p = b.mul(x1p, x1);
        }
        q_eval_sets.push_back(acc_evals);
        x4p = b.mul(x4p, x4);
    }
    const int x4_S = x4p;
    add_term(H2V_TERM_PROOF_POINT, f_pt, x4_S);
    int f_eval = zero;
    std::vector<int> r_evals;
    for (int s = 0; s < S; s++) {
        const std::vector<int> &pts = set_points[s];
        const int m = (int)pts.size();
        int r_eval = zero;
        for (int i = 0; i < m; i++) {
            int num = one;
            for (int j = 0; j < m; j++) if (j != i) num = b.mul(num, b.sub(x3, pts[j]));
            r_eval = b.add(r_eval, b.mul(q_eval_sets[s][i], b.mul(num, inv_out[interp_idx[s][i]])));
        }
        r_evals.push_back(r_eval);
    }
    for (int s = S - 1; s >= 0; s--) {
        const int e = b.mul(b.sub(q_evals[s], r_evals[s]), inv_out[fden_idx[s]]);
        f_eval = b.add(b.mul(f_eval, x2), e);
    }
    int v = zero;
    x4p = one;
    for (int s = 0; s <= S; s++) {
        v = b.add(v, b.mul(x4p, s < S ? q_evals[s] : f_eval));
        x4p = b.mul(x4p, x4);
    }
    G1 neg_g1 = g1_generator();
    neg_g1.y = fp_neg(neg_g1.y);
    add_term(H2V_TERM_VK_BASE, vk_base(BaseKey{BK_NEG_G1, "", 0}, neg_g1), v);
    add_term(H2V_TERM_PROOF_POINT, pi_pt, x3);
    {   // per-proof terms first, VK-base terms last (stable)
        std::vector<int> order(terms.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
        std::stable_sort(order.begin(), order.end(), [&](int a2, int c2) { return (terms[a2].first == H2V_TERM_VK_BASE) < (terms[c2].first == H2V_TERM_VK_BASE); });
        std::vector<std::pair<int, int>> t2;
        std::vector<int> s2;
        for (int o : order) { t2.push_back(terms[o]); s2.push_back(term_scalar[o]); }
        terms.swap(t2); term_scalar.swap(s2);
    }
    // ---- recursion
    const int n_main_terms = (int)terms.size();
    std::vector<uint32_t> acc_coords;
    if (vk.recursive) {
        // ivc.py: fixed_bases / layout (emitters/aiken.rs:659-702)
        std::vector<std::string> bases_hex;
        {
            uint8_t c48[48];
            g1_compress(neg_g1, c48);
            static const char *hx = "0123456789abcdef";
            std::string h;
            for (int i = 0; i < 48; i++) { h.push_back(hx[c48[i] >> 4]); h.push_back(hx[c48[i] & 15]); }
            bases_hex.push_back(h);
        }
        for (const auto &h : vk.fixed_comm) bases_hex.push_back(h);
        for (const auto &h : vk.perm_comm) bases_hex.push_back(h);
        for (const auto &in : vk.inner) { for (const auto &h : in.fixed_comm) bases_hex.push_back(h); for (const auto &h : in.perm_comm) bases_hex.push_back(h); }
        const int n = vk.n_pi, f = (int)bases_hex.size(), nb_vks = 1 + (int)vk.inner.size();
        if (n < nb_vks + f + 10) throw CompileError("not enough public inputs to support recursion (aiken.rs:702)");
        auto I = [](int k) { return k - 1; };
        b.emit(H2V_OP_ASSERT_ZERO, 0, b.sub(pis[I(1)], b.constant(vk.transcript_repr)), 0);
        add_term(H2V_TERM_ACC_POINT, 0, pis[I(n - f - 5)]);
        add_term(H2V_TERM_ACC_POINT, 1, pis[I(n - f)]);
        std::vector<BaseKey> keys{BaseKey{BK_NEG_G1, "", 0}};
        for (size_t i = 0; i < fixed_pts.size(); i++) keys.push_back(BaseKey{BK_FIXED, "", (int)i});
        for (size_t i = 0; i < perm_cpts.size(); i++) keys.push_back(BaseKey{BK_COMMON, "", (int)i});
        for (const auto &in : vk.inner) {
            for (size_t i = 0; i < in.fixed_comm.size(); i++) keys.push_back(BaseKey{BK_INNER_F, in.name, (int)i});
            for (size_t i = 0; i < in.perm_comm.size(); i++) keys.push_back(BaseKey{BK_INNER_P, in.name, (int)i});
        }
        for (int k = 0; k < f; k++) add_term(H2V_TERM_VK_BASE, vk_base(keys[k], decompress_hex(bases_hex[k], true)), pis[I(n - f + 1 + k)]);
        acc_coords = {(uint32_t)I(n - f - 8), (uint32_t)I(n - f - 9), (uint32_t)I(n - f - 6), (uint32_t)I(n - f - 7),
                      (uint32_t)I(n - f - 3), (uint32_t)I(n - f - 4), (uint32_t)I(n - f - 1), (uint32_t)I(n - f - 2)};
    }
    for (size_t t = 0; t < term_scalar.size(); t++) b.emit(H2V_OP_OUT_SCALAR, (int)t, term_scalar[t], 0);
    b.emit(H2V_OP_END);
    // ---- trace registers (slot ids of plan.py: TRACE_NAMES)
    std::vector<std::pair<int, int>> trace_virt = {{0, theta}, {1, beta}, {2, gamma}, {3, trash}, {4, y}, {5, x}, {6, x1}, {7, x2}, {8, x3}, {9, x4},
                                                   {12, x_last}, {13, xn}, {14, l_last}, {15, l_0}, {16, active_rows}, {17, h_eval}, {18, vanishing_s},
                                                   {19, f_eval}, {20, v}};
    for (size_t i = 0; i < expressions.size() && i < 256; i++) trace_virt.emplace_back(32 + (int)i, expressions[i]);
    std::vector<int> keep_alive;
    for (const auto &tv : trace_virt) keep_alive.push_back(tv.second);
    Sched narrow, wide;
    bool has_wide = false;
    schedule_and_allocate(b, keep_alive, narrow, has_wide, wide);
    // ---- G2 side
    std::vector<uint8_t> sg2_raw;
    hex_to_bytes(vk.s_g2, sg2_raw);
    G2 s_g2;
    {
        const std::string err = g2_decompress(sg2_raw.data(), s_g2, true);
        if (!err.empty()) throw CompileError(err);
    }
    const std::vector<Line> lines_sg2 = g2_line_table(s_g2), lines_g2 = g2_line_table(g2_generator());
    if (lines_sg2.size() != H2V_MILLER_LINES) throw CompileError("line table length");
    // ---- to_bytes
    std::vector<uint8_t> body;
    std::map<std::string, uint32_t> offs;
    auto section = [&](const char *name) { pad16(body); offs[name] = (uint32_t)body.size(); };
    section("instr");
    for (const Ins &i : narrow.al.instrs) put_instr(body, i);
    if (has_wide) { section("instr_wide"); for (const Ins &i : wide.al.instrs) put_instr(body, i); }
    section("consts");
    for (const U256 &c : b.consts) { uint8_t t[32]; to_le_bytes(FR().to_mont(c), t); body.insert(body.end(), t, t + 32); }
    section("points");
    for (uint32_t o : points) put32(body, o);
    section("vk_bases");
    for (const G1 &pt : vk_bases) {
        uint8_t t[96];
        memset(t, 0, sizeof t);
        if (!pt.inf) { fp_mont392_bytes(pt.x, t); fp_mont392_bytes(pt.y, t + 48); }
        body.insert(body.end(), t, t + 96);
    }
    section("terms");
    for (const auto &t : terms) { put32(body, (uint32_t)t.first); put32(body, (uint32_t)t.second); }
    section("lines_sg2");
    for (const Line &l : lines_sg2) { put_f2(body, l.lam); put_f2(body, l.c); }
    section("lines_g2");
    for (const Line &l : lines_g2) { put_f2(body, l.lam); put_f2(body, l.c); }
    section("trace");
    for (const auto &tv : trace_virt) { put32(body, (uint32_t)tv.first); put32(body, (uint32_t)narrow.al.mapping[tv.second]); }
    auto put_line_slots = [&](const Line &l) {   // bls12_381.py: line_slots
        const F2 nl = f2_neg(l.lam), nxl = f2_mul(f2_xi(), nl), xc = f2_mul(f2_xi(), l.c);
        const FpE vs[8] = {nl.a, nl.b, nxl.a, nxl.b, l.c.a, l.c.b, xc.a, xc.b};
        for (const FpE &v8 : vs) put_slot28(body, v8);
    };
    section("lines28_sg2");
    for (const Line &l : lines_sg2) put_line_slots(l);
    section("lines28_g2");
    for (const Line &l : lines_g2) put_line_slots(l);
    pad16(body);
    const uint32_t hdr_len = 8 + 4 * H2V_PLAN_HDR_WORDS;
    std::vector<uint32_t> f = {H2V_PLAN_VERSION, proof_len, (uint32_t)vk.n_pi, (uint32_t)n_ci, (uint32_t)narrow.al.n_regs, (uint32_t)narrow.al.instrs.size(),
                               (uint32_t)b.consts.size(), (uint32_t)points.size(), (uint32_t)vk_bases.size(), (uint32_t)terms.size(),
                               (uint32_t)trace_virt.size(), (uint32_t)pi_pt, n_squeezes, stream_len};
    for (const char *k : {"instr", "consts", "points", "vk_bases", "terms", "lines_sg2", "lines_g2", "trace"}) f.push_back(hdr_len + offs[k]);
    f.push_back(hdr_len + (uint32_t)body.size());
    f.push_back(hdr_len + offs["lines28_sg2"]);
    f.push_back(hdr_len + offs["lines28_g2"]);
    f.push_back(vk.recursive ? 1u : 0u);
    f.push_back((uint32_t)n_main_terms);
    for (int k = 0; k < 8; k++) f.push_back(vk.recursive ? acc_coords[k] : 0u);
    f.push_back((uint32_t)narrow.lanes);
    if (has_wide) { f.push_back((uint32_t)wide.lanes); f.push_back((uint32_t)wide.al.n_regs); f.push_back((uint32_t)wide.al.instrs.size()); f.push_back(hdr_len + offs["instr_wide"]); }
    else { f.push_back(0); f.push_back(0); f.push_back(0); f.push_back(0); }
    while (f.size() < H2V_PLAN_HDR_WORDS) f.push_back(0);
    std::vector<uint8_t> out(H2V_PLAN_MAGIC, H2V_PLAN_MAGIC + 8);
    for (uint32_t w : f) put32(out, w);
    out.insert(out.end(), body.begin(), body.end());
    return out;
}

}  // namespace h2vplan
