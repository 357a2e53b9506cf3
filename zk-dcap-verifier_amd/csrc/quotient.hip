// h(X) numerator on the extended coset — replaces halo2_proofs (zkwebauthn @ c254c75,
// Cargo.lock:1314-1327) src/plonk/evaluation.rs Evaluator::evaluate_h.
//
// halo2 walks the extended rows three times on the CPU (custom gates; permutation; each lookup),
// re-reading `values` every time.  Here the proving key's GraphEvaluator programs, the permutation
// argument and the lookup arguments are compiled ONCE (at zk_quotient_program_load) into a single
// straight-line micro-program, so that one kernel launch evaluates a row completely: every coset
// column is read exactly once per rotation (HBM-bound in the ideal), the running value lives in
// registers, and the few live intermediates sit in LDS slots assigned by a linear-scan allocator
// (slot-major layout: lane-consecutive 16-byte accesses, conflict free).
//
// The micro-ISA (one uint4 per instruction) is internal; the input format is the "ZKQ1" blob
// documented in INTEGRATION.md.
#include "ctx.h"
#include "quotient.h"
#include <algorithm>
#include <functional>
#include <mutex>

namespace zk {

int ntt_pow_tables(zk_ctx* ctx, uint32_t log_n, const u256& omega, const void** lo, const void** hi, uint32_t* lo_bits);
int domain_coeff_to_extended_batch(zk_ctx* ctx, const void* const* coeffs, void* const* outs, size_t count, uint32_t k, uint32_t ek);
int domain_extended_to_coeff(zk_ctx* ctx, void* d_a, uint32_t k, uint32_t ek);
int domain_divide_by_vanishing(zk_ctx* ctx, void* d_a, uint32_t k, uint32_t ek);
u256 domain_omega(uint32_t k);

enum { VS_CONST = 0, VS_INTER, VS_FIXED, VS_ADVICE, VS_INSTANCE, VS_CHALLENGE, VS_BETA, VS_GAMMA, VS_THETA, VS_Y, VS_PREV };
enum { OP_ADD = 0, OP_SUB, OP_MUL, OP_SQUARE, OP_DOUBLE, OP_NEGATE, OP_HORNER, OP_STORE };
// ------------------------------------------------------------------------------------------------
// interpreter kernel
// ------------------------------------------------------------------------------------------------
#ifdef ZK_EMU
#define ZK_UNIFORM(x) (x)
#else
#define ZK_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#endif

constexpr uint32_t QUOT_NREG = 1;   // first slot of the allocator is a register, the rest LDS (3 -> 1: 128 -> 116 VGPRs, 9.74 -> 9.18 ms at k = 19: profiles/r02)


// One row per thread.  The loop is software-pipelined by one instruction: while instruction pc executes, the column /
// constant operands of instruction pc + 1 are already in flight (the kernel is otherwise bound by the latency of ~850
// dependent 32-byte loads per row, not by arithmetic: profiles/r01).  Slot, accumulator and X-power operands are read at
// execute time because the previous instruction may just have written them.
ZK_KERNEL void ZK_LAUNCH_BOUNDS(256) ZK_WAVES_PER_EU(4) quotient_kernel(QuotArgs q) {
    ZK_DYN_SHARED(uint4, smem);
    const uint32_t T = blockDim.x, tid = threadIdx.x;
    uint32_t idx0 = q.row_base + blockIdx.x * T + tid;                // this thread's row
    uint32_t oidx = idx0 - q.row_base;
    if (q.strided) {
        const uint32_t j = idx0 & ((1u << q.sub_log) - 1u), i = idx0 >> q.sub_log;
        idx0 = (i << q.stride_log) + j;
        oidx = (j << q.k_log) + i;
    }
    const uint32_t mask = (1u << q.size_log) - 1u;
    u256 acc = Fr::zero(), xpow = Fr::one(), rg0 = Fr::zero();        // slot 0 lives in VGPRs
    if (q.uses_xpow) {  // extended_omega^(position of this row in the extended domain)
        const uint32_t xi = idx0 * q.xpow_mul + q.xpow_add;
        xpow = load_u256(q.tw_lo, xi & ((1u << q.lo_bits) - 1u));
        const uint32_t h = xi >> q.lo_bits;
        if (h) xpow = Fr::mul(xpow, load_u256(q.tw_hi, h));
    }
    auto prefetch = [&](uint32_t src) -> u256 {        // memory operands only; everything else is resolved later
        const uint32_t kind = src >> 28, pay = src & 0x0fffffffu;
        if (kind == K_COL) return load_u256(q.cols[pay >> 8], (idx0 + q.rot_off[pay & 0xffu]) & mask);
        if (kind == K_CONST) return load_u256(q.consts, pay);
        return Fr::zero();
    };
    auto resolve = [&](uint32_t src, const u256& pre) -> u256 {
        const uint32_t kind = src >> 28, pay = src & 0x0fffffffu;
        switch (kind) {
            case K_SLOT: {
                switch (pay) {
                    case 0: return rg0;
                    default: break;
                }
                const uint32_t ls = pay - QUOT_NREG;
                uint4 l = smem[(2 * ls) * T + tid], h = smem[(2 * ls + 1) * T + tid];
                u256 o;
                o.v[0] = l.x; o.v[1] = l.y; o.v[2] = l.z; o.v[3] = l.w; o.v[4] = h.x; o.v[5] = h.y; o.v[6] = h.z; o.v[7] = h.w;
                return o;
            }
            case K_ACC: return acc;
            case K_XPOW: return xpow;
            default: return pre;       // K_COL / K_CONST (already loaded) or K_NONE
        }
    };
    uint4 ins = q.n_instr ? q.code[0] : make_uint4(M_MOV | (1u << 8), K_NONE << 28, K_NONE << 28, K_NONE << 28);
    u256 pa = prefetch(ZK_UNIFORM(ins.y)), pb = prefetch(ZK_UNIFORM(ins.z)), pc_ = prefetch(ZK_UNIFORM(ins.w));
    for (uint32_t pc = 0; pc < q.n_instr; pc++) {
        const uint32_t w0 = ZK_UNIFORM(ins.x), sa = ZK_UNIFORM(ins.y), sb = ZK_UNIFORM(ins.z), sc = ZK_UNIFORM(ins.w);
        // put the next instruction's loads in flight before this one's arithmetic
        uint4 nxt = ins;
        u256 na = pa, nb = pb, nc = pc_;
        if (pc + 1 < q.n_instr) {
            nxt = q.code[pc + 1];
            na = prefetch(ZK_UNIFORM(nxt.y)); nb = prefetch(ZK_UNIFORM(nxt.z)); nc = prefetch(ZK_UNIFORM(nxt.w));
        }
        const uint32_t op = w0 & 0xffu;
        u256 res;
        const u256 a = resolve(sa, pa);
        switch (op) {
            // every value of a row (slots, accumulator) lives in [0, 2p] (field.cuh, redundant ranges): products skip their final subtraction, sums and
            // differences are corrected by 2p (same cost as by p), memory operands arrive canonical, and the row's result is normalised once at the end
            case M_ADD: res = Fr::red2p(Fr::add_lazy(a, resolve(sb, pb))); break;
            case M_SUB: res = Fr::sub2(a, resolve(sb, pb)); break;
            case M_MUL: res = Fr::mul_lazy(a, resolve(sb, pb)); break;
            case M_SQR: res = Fr::sqr_lazy(a); break;
            case M_DBL: res = Fr::dbl2(a); break;
            case M_NEG: res = Fr::neg2(a); break;
            case M_MULADD: res = Fr::red2p(Fr::add_lazy(Fr::mul_lazy(a, resolve(sb, pb)), resolve(sc, pc_))); break;
            case M_FOLD2: res = Fr::mul2_add_2p(acc, resolve(sc, pc_), a, resolve(sb, pb)); break;
            default: res = a; break;
        }
        if ((w0 >> 8) & 0xffu) {
            acc = res;
        } else {
            const uint32_t slot = w0 >> 16;
            switch (slot) {
                case 0: rg0 = res; break;
                default: {
                    const uint32_t ls = slot - QUOT_NREG;
                    smem[(2 * ls) * T + tid] = make_uint4(res.v[0], res.v[1], res.v[2], res.v[3]);
                    smem[(2 * ls + 1) * T + tid] = make_uint4(res.v[4], res.v[5], res.v[6], res.v[7]);
                }
            }
        }
        ins = nxt; pa = na; pb = nb; pc_ = nc;
    }
    store_u256(q.out, oidx, Fr::normalize(acc));
}

// ------------------------------------------------------------------------------------------------
// compiler: ZKQ1 blob -> micro-program
// ------------------------------------------------------------------------------------------------
namespace {

struct VSrc { uint32_t kind, a, b; };
struct Calc { uint32_t op, target; VSrc s0, s1; std::vector<VSrc> parts; };
struct Graph {
    std::vector<u256> constants;
    std::vector<int32_t> rotations;
    uint32_t num_intermediates = 0;
    std::vector<Calc> calcs;
};

struct Reader {
    const uint32_t* w; size_t n, pos = 0; bool ok = true;
    uint32_t get() { if (pos >= n) { ok = false; return 0; } return w[pos++]; }
    VSrc vs() { VSrc v; v.kind = get(); v.a = get(); v.b = get(); return v; }
};

bool read_graph(Reader& r, Graph& g) {
    uint32_t nc = r.get();
    if (!r.ok || nc > (1u << 20)) return false;
    for (uint32_t i = 0; i < nc; i++) { u256 c; for (int j = 0; j < 8; j++) c.v[j] = r.get(); g.constants.push_back(c); }
    uint32_t nr = r.get();
    if (!r.ok || nr > 255) return false;
    for (uint32_t i = 0; i < nr; i++) g.rotations.push_back((int32_t)r.get());
    g.num_intermediates = r.get();
    uint32_t ncalc = r.get();
    // every intermediate is the target of some calculation, so more intermediates than calculations is malformed (and would size the compiler's tables from an
    // unchecked 32-bit word of the blob)
    if (!r.ok || ncalc > (1u << 22) || g.num_intermediates > ncalc) return false;
    for (uint32_t i = 0; i < ncalc; i++) {
        Calc c;
        c.op = r.get(); c.target = r.get(); c.s0 = r.vs();
        c.s1 = VSrc{0, 0, 0};
        if (c.op == OP_ADD || c.op == OP_SUB || c.op == OP_MUL) c.s1 = r.vs();
        else if (c.op == OP_HORNER) {
            c.s1 = r.vs();
            uint32_t np = r.get();
            if (!r.ok || np > (1u << 20)) return false;
            for (uint32_t p = 0; p < np; p++) c.parts.push_back(r.vs());
        } else if (c.op > OP_STORE) return false;
        if (!r.ok || c.target >= g.num_intermediates) return false;
        g.calcs.push_back(c);
    }
    return r.ok;
}

// Common factor of a theta-compression (graph -> graph, before any code is emitted).  halo2 compresses the m expressions of a lookup argument as Horner(0, [e_0 .. e_m-1], theta),
// each e_j evaluated on its own; lookups that are switched by a selector have e_j = q * a_j (the reference's create_bit_lookup: q * char, q * bit_j, sgx_dcap_verifier.rs:95-134), so the
// compression is q * Horner(0, [a_j], theta): m - 1 products fewer per row, the same field element (the compression is linear in its parts).  A part that IS the factor counts as
// factor * 1.  Applied when every part is the factor or a product with it whose only reader is this Horner; graphs whose intermediates are written more than once are left alone.
void factor_common_horner(Graph& g) {
    auto same = [](const VSrc& x, const VSrc& y) { return x.kind == y.kind && x.a == y.a && (x.b == y.b || x.kind == VS_INTER || x.kind == VS_CONST || x.kind >= VS_CHALLENGE); };
    std::vector<int> writer(g.num_intermediates, -1), readers(g.num_intermediates, 0);
    auto count = [&](const VSrc& v) { if (v.kind == VS_INTER && v.a < readers.size()) readers[v.a]++; };
    for (size_t i = 0; i < g.calcs.size(); i++) {
        const Calc& k = g.calcs[i];
        if (writer[k.target] >= 0) return;                             // an intermediate written twice: not the single-assignment form add_calculation produces
        writer[k.target] = (int)i;
        count(k.s0);
        if (k.op == OP_ADD || k.op == OP_SUB || k.op == OP_MUL || k.op == OP_HORNER) count(k.s1);
        for (auto& pp : k.parts) count(pp);
    }
    for (size_t i = 0; i < g.calcs.size(); i++) {
        if (g.calcs[i].op != OP_HORNER || g.calcs[i].parts.size() < 2) continue;
        const Calc k = g.calcs[i];
        if (!(k.s0.kind == VS_CONST && k.s0.a < g.constants.size() && Fr::is_zero(g.constants[k.s0.a]))) continue;
        auto mul_of = [&](const VSrc& part) -> const Calc* {            // the product behind a part, if this Horner is its only reader
            if (part.kind != VS_INTER || part.a >= writer.size() || writer[part.a] < 0 || writer[part.a] >= (int)i) return nullptr;
            const Calc& c = g.calcs[writer[part.a]];
            return c.op == OP_MUL && readers[part.a] == 1 ? &c : nullptr;
        };
        std::vector<VSrc> cands;
        if (const Calc* c0 = mul_of(k.parts[0])) { cands.push_back(c0->s0); cands.push_back(c0->s1); }
        cands.push_back(k.parts[0]);                                    // (the first part may be the bare factor: q * a, q * a * 2, q * a * 3, ...)
        for (const VSrc& f : cands) {
            std::vector<VSrc> rest;
            size_t products = 0;
            bool all = true;
            for (const VSrc& part : k.parts) {
                if (same(part, f)) { rest.push_back(VSrc{VS_CONST, 0xFFFFFFFFu, 0}); continue; }          // factor * 1 (the constant is appended below)
                const Calc* c = mul_of(part);
                if (c && same(c->s0, f)) { rest.push_back(c->s1); products++; }
                else if (c && same(c->s1, f)) { rest.push_back(c->s0); products++; }
                else { all = false; break; }
            }
            if (!all || products < 2) continue;                         // (one product saved at least: `products` go, one comes)
            uint32_t one_at = 0xFFFFFFFFu;
            for (auto& v : rest)
                if (v.kind == VS_CONST && v.a == 0xFFFFFFFFu) {
                    if (one_at == 0xFFFFFFFFu) {
                        for (uint32_t c = 0; c < g.constants.size(); c++) if (Fr::eq(g.constants[c], Fr::one())) { one_at = c; break; }
                        if (one_at == 0xFFFFFFFFu) { one_at = (uint32_t)g.constants.size(); g.constants.push_back(Fr::one()); }
                    }
                    v.a = one_at;
                }
            Calc h = k, m;
            h.target = g.num_intermediates++;
            h.parts = rest;
            m.op = OP_MUL; m.target = k.target; m.s0 = VSrc{VS_INTER, h.target, 0}; m.s1 = f;
            g.calcs[i] = h;
            g.calcs.insert(g.calcs.begin() + i + 1, m);
            writer.push_back((int)i);
            readers.push_back(1);
            for (auto& w : writer) if (w > (int)i) w++;                 // the calculations behind the insertion moved by one
            writer[k.target] = (int)i + 1;
            i++;
            break;
        }
    }
}

// virtual-register program
struct VIns { uint32_t op; int dst; /* -1 = ACC, else vreg */ uint32_t src[3]; int vsrc[3]; /* vreg id when kind==SLOT */ int nsrc; };

struct Builder {
    QuotProgram& P;
    std::vector<VIns> ins;
    int next_vreg = 0;
    int mode = 0;                            // 0: every identity; 1: only those of degree > low_deg; 2: only those of degree <= low_deg
    uint32_t low_deg = 3, pending = 0;       // pending: folds skipped since the last one taken (the next fold multiplies by y^(pending + 1))
    Builder(QuotProgram& p, int m, uint32_t ld) : P(p), mode(m), low_deg(ld) {}
    static uint32_t enc(uint32_t kind, uint32_t pay) { return (kind << 28) | pay; }
    struct Opnd { uint32_t word; int vreg; };
    Opnd slot(int v) { return Opnd{enc(K_SLOT, 0), v}; }
    Opnd cst(uint32_t i) { return Opnd{enc(K_CONST, i), -1}; }
    Opnd acc() { return Opnd{enc(K_ACC, 0), -1}; }
    Opnd xpow() { P.uses_xpow = true; return Opnd{enc(K_XPOW, 0), -1}; }
    uint32_t rot_id(int32_t rot) {
        for (size_t i = 0; i < P.rotations.size(); i++) if (P.rotations[i] == rot) return (uint32_t)i;
        P.rotations.push_back(rot);
        return (uint32_t)P.rotations.size() - 1;
    }
    Opnd col(uint32_t col_id, int32_t rot) { return Opnd{enc(K_COL, (col_id << 8) | rot_id(rot)), -1}; }
    int emit(uint32_t op, int dst, std::initializer_list<Opnd> ops) {
        VIns v; v.op = op; v.dst = dst; v.nsrc = 0;
        for (auto& o : ops) { v.src[v.nsrc] = o.word; v.vsrc[v.nsrc] = o.vreg; v.nsrc++; }
        for (int i = v.nsrc; i < 3; i++) { v.src[i] = enc(K_NONE, 0); v.vsrc[i] = -1; }   // unused operand: nothing is fetched
        ins.push_back(v);
        return dst;
    }
    int tmp(uint32_t op, std::initializer_list<Opnd> ops) { return emit(op, next_vreg++, ops); }
    // value numbering for the theta-compression chains: lookups that compress the SAME table tuple (range checks; 9 of the 11 lookups of the sgx-shaped circuit,
    // 5 of the 7 base64 lookups of the reference) share the chain instead of recomputing it per lookup.  Only used where no operand is the accumulator
    // (virtual registers are written once; column / constant operands do not change within a row).
    std::map<std::vector<uint64_t>, int> shared;
    int tmp_shared(uint32_t op, std::initializer_list<Opnd> ops) {
        std::vector<uint64_t> key{op};
        for (auto& o : ops) { if ((o.word >> 28) == K_ACC) return tmp(op, ops); key.push_back(((uint64_t)o.word << 32) | (uint32_t)(o.vreg + 1)); }
        auto it = shared.find(key);
        if (it != shared.end()) return it->second;
        const int v = tmp(op, ops);
        shared[key] = v;
        return v;
    }
    Opnd ypow(uint32_t e) {                  // the run's constant y^e
        if (e == 1) return cst(P.c_y);
        for (size_t i = 0; i < P.ypow_exps.size(); i++) if (P.ypow_exps[i] == e) return cst(P.c_ypow + (uint32_t)i);
        P.ypow_exps.push_back(e);
        return cst(P.c_ypow + (uint32_t)P.ypow_exps.size() - 1);
    }
    bool is_ypow(uint32_t word) const {
        if ((word >> 28) != K_CONST) return false;
        const uint32_t pay = word & 0x0fffffffu;
        return pay == P.c_y || (pay >= P.c_ypow && pay < P.c_ypow + P.ypow_exps.size());
    }
    // value = value*y + term for an identity of degree `deg` (in the columns); a part of the program that leaves this identity out owes the accumulator one power of y
    void fold(Opnd term, uint32_t deg) {
        const bool take = mode == 0 || (mode == 2) == (deg <= low_deg);
        if (!take) { pending++; P.folds_skipped++; return; }
        emit(M_MULADD, -1, {acc(), ypow(pending + 1), term});
        pending = 0;
        P.folds_taken++;
    }
    void skip_fold() { pending++; P.folds_skipped++; }
    void flush_pending() { if (pending && P.folds_taken) emit(M_MUL, -1, {acc(), ypow(pending)}); pending = 0; }
    // Peephole over the finished program: value = value*y + t, where t = a*b is the instruction just before and nothing else reads t, becomes
    // acc = acc*y + a*b with ONE Montgomery reduction (M_FOLD2, Field::mul2_add): 192 limb products instead of 256 on each of the ~90 folds of a row.
    void fuse_folds_pass() {
        std::vector<int> uses(next_vreg, 0);
        for (auto& v : ins) for (int i = 0; i < v.nsrc; i++) if (v.vsrc[i] >= 0) uses[v.vsrc[i]]++;
        std::vector<VIns> out;
        for (auto& v : ins) {
            const bool is_fold = v.op == M_MULADD && v.dst < 0 && v.nsrc == 3 && (v.src[0] >> 28) == K_ACC && is_ypow(v.src[1]) &&
                                 (v.src[2] >> 28) == K_SLOT && v.vsrc[2] >= 0;
            if (is_fold && !out.empty() && out.back().op == M_MUL && out.back().dst == v.vsrc[2] && uses[v.vsrc[2]] == 1 &&
                (out.back().src[0] >> 28) != K_ACC && (out.back().src[1] >> 28) != K_ACC) {
                const VIns m = out.back();
                out.pop_back();
                VIns f; f.op = M_FOLD2; f.dst = -1; f.nsrc = 3;
                f.src[0] = m.src[0]; f.vsrc[0] = m.vsrc[0]; f.src[1] = m.src[1]; f.vsrc[1] = m.vsrc[1];
                f.src[2] = v.src[1]; f.vsrc[2] = -1;
                out.push_back(f);
            } else out.push_back(v);
        }
        ins.swap(out);
    }

};

}  // namespace

constexpr uint32_t SPLIT_LOW_DEGREE = 3;     // identities up to this degree form the low part: their share of h(X) has degree below 2 n — two cosets

static int compile_program(zk_ctx* ctx, const uint32_t* words, size_t nwords, QuotProgram& P, int mode = 0) {
    Reader r{words, nwords};
    if (r.get() != 0x31514B5Au) return ctx->fail(ZK_ERR_PROGRAM, "quotient program: bad magic");
    P.k = r.get(); P.ek = r.get(); P.n_fixed = r.get(); P.n_advice = r.get(); P.n_instance = r.get(); P.n_challenges = r.get();
    P.blinding = r.get(); P.degree = r.get(); P.n_perm_cols = r.get();
    if (!r.ok || P.ek > 27 || P.k > P.ek || P.n_perm_cols > 4096 || P.n_fixed > 65536 || P.n_advice > 65536 || P.n_instance > 65536)
        return ctx->fail(ZK_ERR_PROGRAM, "quotient program: header out of range");
    for (uint32_t i = 0; i < 2 * P.n_perm_cols; i++) P.perm_cols.push_back(r.get());
    P.n_lookups = r.get();
    if (!r.ok || P.n_lookups > 4096) return ctx->fail(ZK_ERR_PROGRAM, "quotient program: truncated header");
    Graph custom;
    std::vector<Graph> lookups(P.n_lookups);
    if (!read_graph(r, custom)) return ctx->fail(ZK_ERR_PROGRAM, "quotient program: malformed custom-gate graph");
    for (auto& g : lookups) if (!read_graph(r, g)) return ctx->fail(ZK_ERR_PROGRAM, "quotient program: malformed lookup graph");
    if (ctx->tune.quot_factor_horner) {
        factor_common_horner(custom);
        for (auto& g : lookups) factor_common_horner(g);
    }
    if (P.n_perm_cols && P.degree < 3) return ctx->fail(ZK_ERR_PROGRAM, "quotient program: cs_degree < 3 with a permutation");
    const uint32_t chunk_len = P.n_perm_cols ? P.degree - 2 : 1;
    P.n_sets = P.n_perm_cols ? (P.n_perm_cols + chunk_len - 1) / chunk_len : 0;

    // column ids
    uint32_t c = 0;
    P.col_fixed = c; c += P.n_fixed;
    P.col_advice = c; c += P.n_advice;
    P.col_instance = c; c += P.n_instance;
    P.col_l0 = c++; P.col_llast = c++; P.col_lactive = c++;
    P.col_sigma = c; c += P.n_perm_cols;
    P.col_z = c; c += P.n_sets;
    P.col_lk_z = c; c += P.n_lookups;
    P.col_lk_a = c; c += P.n_lookups;
    P.col_lk_s = c; c += P.n_lookups;
    P.n_cols = c;
    if (c >= (1u << 20)) return ctx->fail(ZK_ERR_LIMIT, "quotient program: too many columns");
    // constant table: graph constants first
    std::vector<uint32_t> gbase;   // base index of each graph's constants
    auto add_consts = [&](const Graph& g) { gbase.push_back((uint32_t)P.graph_consts.size()); for (auto& k : g.constants) P.graph_consts.push_back(k); };
    add_consts(custom);
    for (auto& g : lookups) add_consts(g);
    uint32_t ci = (uint32_t)P.graph_consts.size();
    P.c_zero = ci++; P.c_one = ci++;
    P.c_chal = ci; ci += P.n_challenges;
    P.c_beta = ci++; P.c_gamma = ci++; P.c_theta = ci++; P.c_y = ci++;
    P.c_delta = ci; ci += P.n_perm_cols;
    P.c_ypow = ci;                               // y^e constants follow as the folds ask for them; n_consts is final after the last fold
    P.n_consts = ci;

    Builder B(P, mode, SPLIT_LOW_DEGREE);
    // ---- graphs -------------------------------------------------------------------------------
    // Demand-driven emission: a calculation is emitted right before its first use (depth-first from
    // the graph's result), so e.g. the gate polynomials of halo2's final Horner(previous, gates, y)
    // are produced one at a time instead of all being live at once.  Unreachable calculations vanish.
    // degree (in the columns) of a graph's result without emitting anything: what decides an identity's part before its instructions exist
    auto graph_degree = [&](const Graph& g, uint32_t* out) -> int {
        const size_t nc = g.calcs.size();
        *out = 0;
        std::vector<int> writer(g.num_intermediates, -1);
        std::vector<uint32_t> cdeg(nc, 0);
        for (size_t i = 0; i < nc; i++) {
            const Calc& k = g.calcs[i];
            auto od = [&](const VSrc& sv) -> uint32_t {
                if (sv.kind == VS_INTER) return (sv.a < writer.size() && writer[sv.a] >= 0) ? cdeg[writer[sv.a]] : 0u;
                return (sv.kind == VS_FIXED || sv.kind == VS_ADVICE || sv.kind == VS_INSTANCE) ? 1u : 0u;
            };
            uint32_t dg = od(k.s0);
            switch (k.op) {
                case OP_ADD: case OP_SUB: dg = std::max(od(k.s0), od(k.s1)); break;
                case OP_MUL: dg = od(k.s0) + od(k.s1); break;
                case OP_SQUARE: dg = 2 * od(k.s0); break;
                case OP_HORNER: {
                    const uint32_t np = (uint32_t)k.parts.size(), f = od(k.s1);
                    dg = od(k.s0) + np * f;
                    for (uint32_t pi = 0; pi < np; pi++) dg = std::max(dg, od(k.parts[pi]) + (np - 1 - pi) * f);
                    break;
                }
                default: break;
            }
            cdeg[i] = std::min(dg, 1u << 16);
            if (k.target >= writer.size()) return ctx->fail(ZK_ERR_PROGRAM, "quotient program: calculation target out of range");
            writer[k.target] = (int)i;
        }
        if (nc) *out = cdeg[nc - 1];
        return ZK_OK;
    };
    auto run_graph = [&](const Graph& g, uint32_t cbase, bool prev_is_acc, int* result_vreg, uint32_t* result_degree) -> int {
        const size_t nc = g.calcs.size();
        *result_vreg = -1;
        *result_degree = 0;
        if (nc == 0) return ZK_OK;
        // resolve Intermediate(t) operands to the calculation that last wrote t before the reader
        std::vector<int> writer(g.num_intermediates, -1);
        std::vector<std::vector<int>> dep(nc);       // per calc: producing calc index of each operand (or -1)
        auto operands_of = [&](const Calc& k) {
            std::vector<const VSrc*> o;
            o.push_back(&k.s0);
            if (k.op == OP_ADD || k.op == OP_SUB || k.op == OP_MUL || k.op == OP_HORNER) o.push_back(&k.s1);
            for (auto& pp : k.parts) o.push_back(&pp);
            return o;
        };
        for (size_t i = 0; i < nc; i++) {
            for (const VSrc* o : operands_of(g.calcs[i])) {
                int d = -1;
                {   // validate every operand, including those of calculations that end up unreachable
                    const bool rot_ok = o->b < g.rotations.size();
                    bool good = true;
                    switch (o->kind) {
                        case VS_CONST: good = o->a < g.constants.size(); break;
                        case VS_INTER: break;
                        case VS_FIXED: good = o->a < P.n_fixed && rot_ok; break;
                        case VS_ADVICE: good = o->a < P.n_advice && rot_ok; break;
                        case VS_INSTANCE: good = o->a < P.n_instance && rot_ok; break;
                        case VS_CHALLENGE: good = o->a < P.n_challenges; break;
                        default: good = o->kind <= VS_PREV; break;
                    }
                    if (!good) return ctx->fail(ZK_ERR_PROGRAM, "quotient program: operand out of range in calculation %zu", i);
                }
                if (o->kind == VS_INTER) {
                    if (o->a >= writer.size() || writer[o->a] < 0) return ctx->fail(ZK_ERR_PROGRAM, "quotient program: intermediate read before it is written");
                    d = writer[o->a];
                }
                dep[i].push_back(d);
            }
            writer[g.calcs[i].target] = (int)i;
        }
        // degree of every calculation as a polynomial in the columns (a column query 1; constants, challenges, beta / gamma / theta / y and the previous value 0)
        std::vector<uint32_t> cdeg(nc, 0);
        auto leaf_deg = [](const VSrc& sv) -> uint32_t { return (sv.kind == VS_FIXED || sv.kind == VS_ADVICE || sv.kind == VS_INSTANCE) ? 1u : 0u; };
        for (size_t i = 0; i < nc; i++) {
            const Calc& k = g.calcs[i];
            const std::vector<const VSrc*> ops = operands_of(k);
            auto od = [&](size_t oi) -> uint32_t { return dep[i][oi] >= 0 ? cdeg[dep[i][oi]] : leaf_deg(*ops[oi]); };
            uint32_t dg = od(0);
            switch (k.op) {
                case OP_ADD: case OP_SUB: dg = std::max(od(0), od(1)); break;
                case OP_MUL: dg = od(0) + od(1); break;
                case OP_SQUARE: dg = 2 * od(0); break;
                case OP_HORNER: {
                    const uint32_t np = (uint32_t)k.parts.size(), f = od(1);
                    dg = od(0) + np * f;
                    for (uint32_t pi = 0; pi < np; pi++) dg = std::max(dg, od(2 + pi) + (np - 1 - pi) * f);
                    break;
                }
                default: break;                                        // DOUBLE, NEGATE, STORE: the operand's
            }
            cdeg[i] = std::min(dg, 1u << 16);
        }
        *result_degree = cdeg[nc - 1];
        // Store(column | constant | challenge) — what halo2's add_expression emits for every query — is an ALIAS here: its readers
        // take the memory operand directly (one more 32-byte load per use) instead of parking the value in a slot from its first
        // to its last use; with CSE'd selectors and rotations shared between gates those slots are what limits occupancy.
        std::vector<char> is_alias(nc, 0);
        for (size_t i = 0; i + 1 < nc; i++)
            if (g.calcs[i].op == OP_STORE && dep[i][0] < 0 && g.calcs[i].s0.kind != VS_PREV) is_alias[i] = 1;
        // Rematerialisation: halo2's add_calculation shares every repeated sub-expression, however far apart its uses are (on the CPU an
        // intermediate is a memory cell).  Here a shared value occupies a slot from its first to its last use, and slots (LDS) set the
        // occupancy of the kernel: a value that costs at most QUOT_REMAT_OPS operations over memory operands is recomputed at each use.
        const uint32_t QUOT_REMAT_OPS = (uint32_t)ctx->tune.quot_remat_ops;
        std::vector<uint32_t> cost(nc, 0);
        std::vector<char> cheap(nc, 0);
        for (size_t i = 0; i < nc; i++) {
            if (is_alias[i]) continue;
            uint64_t c = g.calcs[i].op == OP_HORNER ? std::max<size_t>(g.calcs[i].parts.size(), 1) : 1;
            for (int dd : dep[i]) if (dd >= 0) c += cost[dd];
            cost[i] = (uint32_t)std::min<uint64_t>(c, 1u << 20);
            cheap[i] = g.calcs[i].op != OP_HORNER && cost[i] <= QUOT_REMAT_OPS && i + 1 < nc;
        }
        std::vector<int> vreg(nc, -1);
        // a cheap value is only recomputed when its previous copy is FAR behind (it would otherwise hold a slot across that distance); a copy
        // emitted a few instructions ago is simply reused (e.g. sel * advice shared by the 4-5 scaled inputs of one lookup)
        const size_t QUOT_REMAT_DISTANCE = (size_t)ctx->tune.quot_remat_distance;
        std::vector<size_t> emitted_at(nc, 0);
        auto stale = [&](int c) { return cheap[c] && vreg[c] >= 0 && B.ins.size() - emitted_at[c] > QUOT_REMAT_DISTANCE; };
        bool ok = true;
        auto leaf = [&](const VSrc& s) -> Builder::Opnd {
            switch (s.kind) {
                case VS_CONST: if (s.a >= g.constants.size()) ok = false; return B.cst(cbase + s.a);
                case VS_FIXED: if (s.a >= P.n_fixed || s.b >= g.rotations.size()) { ok = false; return B.cst(P.c_zero); } return B.col(P.col_fixed + s.a, g.rotations[s.b]);
                case VS_ADVICE: if (s.a >= P.n_advice || s.b >= g.rotations.size()) { ok = false; return B.cst(P.c_zero); } return B.col(P.col_advice + s.a, g.rotations[s.b]);
                case VS_INSTANCE: if (s.a >= P.n_instance || s.b >= g.rotations.size()) { ok = false; return B.cst(P.c_zero); } return B.col(P.col_instance + s.a, g.rotations[s.b]);
                case VS_CHALLENGE: if (s.a >= P.n_challenges) ok = false; return B.cst(P.c_chal + s.a);
                case VS_BETA: return B.cst(P.c_beta);
                case VS_GAMMA: return B.cst(P.c_gamma);
                case VS_THETA: return B.cst(P.c_theta);
                case VS_Y: return B.cst(P.c_y);
                case VS_PREV: return prev_is_acc ? B.acc() : B.cst(P.c_zero);
                default: ok = false; return B.cst(P.c_zero);
            }
        };
        // explicit stack (graphs can be deep): frame = (calc, next operand to make available)
        struct Frame { int calc; size_t next; };
        std::vector<Frame> st;
        st.push_back(Frame{(int)nc - 1, 0});
        // Horner needs interleaving (part_i must be emitted right before its step), so it keeps a
        // running value across operand visits.
        std::vector<int> horner_cur(nc, -1);
        std::vector<int> horner_first(nc, -1);       // a zero-start Horner whose first part is a memory operand (a column, a constant): no copy, the first step reads it in place
        size_t prev_reads = 0;                       // how many operands of the graph read PreviousValue (halo2: exactly one, the start of the final Horner)
        for (size_t i = 0; i < nc; i++) for (const VSrc* o : operands_of(g.calcs[i])) if (o->kind == VS_PREV) prev_reads++;
        while (!st.empty()) {
            Frame& f = st.back();
            const int ci = f.calc;
            const Calc& k = g.calcs[ci];
            if (vreg[ci] >= 0) { st.pop_back(); continue; }
            const std::vector<int>& d = dep[ci];
            auto opnd_at = [&](size_t oi) -> Builder::Opnd {
                const VSrc& sv = oi == 0 ? k.s0 : (oi == 1 && (k.op == OP_ADD || k.op == OP_SUB || k.op == OP_MUL || k.op == OP_HORNER)) ? k.s1
                                                : k.parts[oi - 2];
                if (d[oi] >= 0) return is_alias[d[oi]] ? leaf(g.calcs[d[oi]].s0) : B.slot(vreg[d[oi]]);
                return leaf(sv);
            };
            if (k.op != OP_HORNER) {
                if (f.next < d.size()) {               // make operand f.next available
                    const size_t oi = f.next++;
                    if (d[oi] >= 0 && !is_alias[d[oi]] && (vreg[d[oi]] < 0 || stale(d[oi]))) { vreg[d[oi]] = -1; st.push_back(Frame{d[oi], 0}); }
                    continue;
                }
                {   // an operand computed earlier may have been handed back (rematerialisation) while a later operand was being produced
                    bool missing = false;
                    for (size_t oi = 0; oi < d.size() && !missing; oi++)
                        if (d[oi] >= 0 && !is_alias[d[oi]] && vreg[d[oi]] < 0) { f.next = oi; missing = true; }
                    if (missing) continue;
                }
                int v = -1;
                switch (k.op) {
                    case OP_ADD: v = B.tmp(M_ADD, {opnd_at(0), opnd_at(1)}); break;
                    case OP_SUB: v = B.tmp(M_SUB, {opnd_at(0), opnd_at(1)}); break;
                    case OP_MUL: v = B.tmp(M_MUL, {opnd_at(0), opnd_at(1)}); break;
                    case OP_SQUARE: v = B.tmp(M_SQR, {opnd_at(0)}); break;
                    case OP_DOUBLE: v = B.tmp(M_DBL, {opnd_at(0)}); break;
                    case OP_NEGATE: v = B.tmp(M_NEG, {opnd_at(0)}); break;
                    default: v = B.tmp(M_MOV, {opnd_at(0)}); break;
                }
                vreg[ci] = v;
                emitted_at[ci] = B.ins.size();
                st.pop_back();
            } else {
                // operands: 0 = start, 1 = factor, 2.. = parts.  Steps happen as soon as part i is ready.
                if (f.next < d.size()) {
                    const size_t oi = f.next;
                    if (d[oi] >= 0 && !is_alias[d[oi]] && (vreg[d[oi]] < 0 || (oi >= 2 && stale(d[oi])))) { vreg[d[oi]] = -1; st.push_back(Frame{d[oi], 0}); continue; }
                    f.next++;
                    if (oi >= 2) {
                        // Horner(0, parts, f) — how halo2 compresses lookup expressions — starts with 0 * f + part_0: take part_0 as it is
                        const bool zero_start = horner_cur[ci] < 0 && horner_first[ci] < 0 && d[0] < 0 && k.s0.kind == VS_CONST && k.s0.a < g.constants.size() &&
                                                Fr::is_zero(g.constants[k.s0.a]);
                        // halo2's custom-gate evaluator ends in Horner(PreviousValue, gates, y): when that is the graph's result and the previous value IS the
                        // accumulator, every step is a fold of the accumulator itself (and fuses with the gate's last product, Builder::fold)
                        const bool in_acc = prev_is_acc && ci == (int)nc - 1 && d[0] < 0 && k.s0.kind == VS_PREV && d[1] < 0 && k.s1.kind == VS_Y && prev_reads == 1;
                        if (in_acc) { B.fold(opnd_at(oi), d[oi] >= 0 ? cdeg[d[oi]] : leaf_deg(k.parts[oi - 2])); horner_cur[ci] = -2; }
                        else if (zero_start) {
                            if (d[oi] >= 0 && !is_alias[d[oi]]) horner_cur[ci] = vreg[d[oi]];
                            else horner_first[ci] = (int)oi;
                        } else {
                            Builder::Opnd curv = horner_cur[ci] >= 0 ? B.slot(horner_cur[ci]) : horner_first[ci] >= 0 ? opnd_at((size_t)horner_first[ci]) : opnd_at(0);
                            horner_cur[ci] = B.tmp_shared(M_MULADD, {curv, opnd_at(1), opnd_at(oi)});
                        }
                    }
                    continue;
                }
                if (horner_cur[ci] == -2) { *result_vreg = -2; return ZK_OK; }          // the result already sits in the accumulator (only the last calculation gets here)
                vreg[ci] = horner_cur[ci] >= 0 ? horner_cur[ci] : B.tmp(M_MOV, {opnd_at(horner_first[ci] >= 0 ? (size_t)horner_first[ci] : 0)});
                st.pop_back();
            }
            if (!ok) return ctx->fail(ZK_ERR_PROGRAM, "quotient program: operand out of range in a calculation");
        }
        if (!ok) return ctx->fail(ZK_ERR_PROGRAM, "quotient program: operand out of range in a calculation");
        *result_vreg = vreg[nc - 1];
        return ZK_OK;
    };
    int res = -1;
    uint32_t res_deg = 0;
    int rc = run_graph(custom, gbase[0], true, &res, &res_deg);
    if (rc) return rc;
    // (a custom-gate graph that does not end in halo2's Horner(previous, gates, y) arrives as ONE value: it counts as one identity of the high part)
    if (res >= 0 && mode != 2) { B.emit(M_MOV, -1, {B.slot(res)}); P.folds_taken++; }
    else if (res != -2) B.emit(M_MOV, -1, {B.cst(P.c_zero)});   // no custom gates: value = 0 (GraphEvaluator returns zero); -2: the folds left it in the accumulator

    const Builder::Opnd one = B.cst(P.c_one), beta = B.cst(P.c_beta), gamma = B.cst(P.c_gamma);
    const Builder::Opnd l0 = B.col(P.col_l0, 0), llast = B.col(P.col_llast, 0), lact = B.col(P.col_lactive, 0);
    // ---- permutation and lookup arguments -----------------------------------------------------------
    // Every identity of the two arguments is l_0, l_last or l_active_row times a cofactor.  They are collected here in halo2's order (the order fixes each identity's
    // power of y) as closures that emit the COFACTOR, and then emitted either one by one (cofactor * l, fold with y: halo2's own sequence) or — tune quot_group_factors,
    // the default — group by group: h's numerator is a plain sum of y-weighted identities, so
    //     sum_i y^(e_i) l_G U_i  =  l_G * y^(e_last) * Horner_y(U_i with the gaps between the group's members as exponents),
    // one product per identity fewer (the cofactor joins the group's accumulator directly, fused with its own last product where it ends in one) at the price of two
    // products per group.  Same field element either way.
    enum { G_L0 = 0, G_LAST = 1, G_ACT = 2 };
    struct Ident { int group; uint32_t deg; std::function<int()> emit; };   // emit(): instructions that leave the cofactor in a virtual register; returns it
    std::vector<Ident> idents;
    int deferred_rc = ZK_OK;
    auto z = [&](uint32_t s_, int32_t rot) { return B.col(P.col_z + s_, rot); };
    if (P.n_sets) {
        const int32_t last_rot = -(int32_t)(P.blinding + 1);
        for (uint32_t j = 0; j < P.n_perm_cols; j++) {
            const uint32_t ty = P.perm_cols[2 * j], ix = P.perm_cols[2 * j + 1];
            if (ty > 2 || ix >= (ty == 0 ? P.n_advice : ty == 1 ? P.n_fixed : P.n_instance))
                return ctx->fail(ZK_ERR_PROGRAM, "quotient program: permutation column out of range");
        }
        idents.push_back({G_L0, 2, [&]() { return B.tmp(M_SUB, {one, z(0, 0)}); }});
        idents.push_back({G_LAST, 3, [&]() { const int t = B.tmp(M_SQR, {z(P.n_sets - 1, 0)}); return B.tmp(M_SUB, {B.slot(t), z(P.n_sets - 1, 0)}); }});
        for (uint32_t s_ = 1; s_ < P.n_sets; s_++) idents.push_back({G_L0, 2, [&, s_, last_rot]() { return B.tmp(M_SUB, {z(s_, 0), z(s_ - 1, last_rot)}); }});
        for (uint32_t s_ = 0; s_ < P.n_sets; s_++) {
            const uint32_t c0 = s_ * chunk_len, c1 = std::min(c0 + chunk_len, P.n_perm_cols);
            idents.push_back({G_ACT, 2 + (c1 - c0), [&, s_, c0, c1]() {   // l_active * z * one degree-1 factor per column of the set
                auto vcol = [&](uint32_t j) {
                    const uint32_t ty = P.perm_cols[2 * j], ix = P.perm_cols[2 * j + 1];
                    return B.col((ty == 0 ? P.col_advice : ty == 1 ? P.col_fixed : P.col_instance) + ix, 0);
                };
                int left = -1, right = -1;
                for (uint32_t j = c0; j < c1; j++) {
                    int u = B.tmp(M_MULADD, {beta, B.col(P.col_sigma + j, 0), vcol(j)});
                    u = B.tmp(M_ADD, {B.slot(u), gamma});
                    left = B.tmp(M_MUL, {left < 0 ? z(s_, 1) : B.slot(left), B.slot(u)});
                }
                for (uint32_t j = c0; j < c1; j++) {
                    int u = B.tmp(M_MULADD, {B.cst(P.c_delta + j), B.xpow(), vcol(j)});
                    u = B.tmp(M_ADD, {B.slot(u), gamma});
                    right = B.tmp(M_MUL, {right < 0 ? z(s_, 0) : B.slot(right), B.slot(u)});
                }
                return B.tmp(M_SUB, {B.slot(left), B.slot(right)});
            }});
        }
    }
    for (uint32_t n = 0; n < P.n_lookups; n++) {
        uint32_t tv_deg = 0;
        rc = graph_degree(lookups[n], &tv_deg);
        if (rc) return rc;
        auto zc = [&, n](int32_t rot) { return B.col(P.col_lk_z + n, rot); };
        auto ac = [&, n](int32_t rot) { return B.col(P.col_lk_a + n, rot); };
        auto sc = [&, n]() { return B.col(P.col_lk_s + n, 0); };
        idents.push_back({G_L0, 2, [&, zc]() { return B.tmp(M_SUB, {one, zc(0)}); }});
        idents.push_back({G_LAST, 3, [&, zc]() { const int t = B.tmp(M_SQR, {zc(0)}); return B.tmp(M_SUB, {B.slot(t), zc(0)}); }});
        // l_active * (z(wX) (A' + beta) (S' + gamma) - z * (compressed input + beta) (compressed table + gamma)): the graph's result is the second product
        idents.push_back({G_ACT, 1 + std::max(3u, 1 + tv_deg), [&, n, zc, ac, sc]() {
            int tv = -1;
            uint32_t dg = 0;
            const int r_ = run_graph(lookups[n], gbase[1 + n], false, &tv, &dg);
            if (r_ && !deferred_rc) deferred_rc = r_;
            const Builder::Opnd table_value = tv >= 0 ? B.slot(tv) : B.cst(P.c_zero);
            int t = B.tmp(M_ADD, {ac(0), beta});
            int u = B.tmp(M_ADD, {sc(), gamma});
            t = B.tmp(M_MUL, {B.slot(t), B.slot(u)});
            t = B.tmp(M_MUL, {B.slot(t), zc(1)});
            u = B.tmp(M_MUL, {zc(0), table_value});
            return B.tmp(M_SUB, {B.slot(t), B.slot(u)});
        }});
        idents.push_back({G_L0, 2, [&, ac, sc]() { return B.tmp(M_SUB, {ac(0), sc()}); }});
        idents.push_back({G_ACT, 3, [&, ac, sc]() {
            const int ams = B.tmp(M_SUB, {ac(0), sc()});
            const int t = B.tmp(M_SUB, {ac(0), ac(-1)});
            return B.tmp(M_MUL, {B.slot(t), B.slot(ams)});
        }});
    }
    auto lcol = [&](int g_) { return g_ == G_L0 ? l0 : g_ == G_LAST ? llast : lact; };
    auto takes = [&](uint32_t deg) { return mode == 0 || (mode == 2) == (deg <= SPLIT_LOW_DEGREE); };
    const uint32_t M = (uint32_t)idents.size();
    if (!ctx->tune.quot_group_factors) {
        for (auto& id : idents) {
            if (!takes(id.deg)) { B.skip_fold(); continue; }
            const int u = id.emit();
            const int t = B.tmp(M_MUL, {B.slot(u), lcol(id.group)});
            B.fold(B.slot(t), id.deg);
        }
        B.flush_pending();
    } else {
        // the value so far (the custom gates, folded with y) owes one power of y per identity that follows it
        int T = -1;
        if (P.folds_taken) {
            const uint32_t e = B.pending + M;
            T = e ? B.tmp(M_MUL, {B.acc(), B.ypow(e)}) : B.tmp(M_MOV, {B.acc()});
        }
        B.pending = 0;
        for (int g_ : {G_ACT, G_L0, G_LAST}) {
            int prev = -1;
            for (uint32_t i = 0; i < M; i++) {
                if (idents[i].group != g_) continue;
                if (!takes(idents[i].deg)) { P.folds_skipped++; continue; }
                const int u = idents[i].emit();
                if (prev < 0) B.emit(M_MOV, -1, {B.slot(u)});
                else B.emit(M_MULADD, -1, {B.acc(), B.ypow(i - (uint32_t)prev), B.slot(u)});
                prev = (int)i;
                P.folds_taken++;
            }
            if (prev < 0) continue;
            const uint32_t e = M - 1 - (uint32_t)prev;
            const int t2 = B.tmp(M_MUL, {B.acc(), lcol(g_)});
            if (T < 0) T = e ? B.tmp(M_MUL, {B.slot(t2), B.ypow(e)}) : t2;
            else T = e ? B.tmp(M_MULADD, {B.slot(t2), B.ypow(e), B.slot(T)}) : B.tmp(M_ADD, {B.slot(t2), B.slot(T)});
        }
        if (T >= 0) B.emit(M_MOV, -1, {B.slot(T)});
        else B.emit(M_MOV, -1, {B.cst(P.c_zero)});
    }
    if (deferred_rc) return deferred_rc;
    P.n_consts = P.c_ypow + (uint32_t)P.ypow_exps.size();
    if (P.rotations.size() > 255) return ctx->fail(ZK_ERR_LIMIT, "quotient program: more than 255 distinct rotations");

    B.fuse_folds_pass();
    // ---- dead-code elimination + linear-scan slot allocation -------------------------------------
    const int nv = B.next_vreg;
    std::vector<int> last_use(nv, -1);
    std::vector<char> keep(B.ins.size(), 1);
    for (int pass = 0; pass < 2; pass++) {   // one backward sweep removes chains of dead values
        std::fill(last_use.begin(), last_use.end(), -1);
        for (size_t i = 0; i < B.ins.size(); i++) if (keep[i]) for (int s = 0; s < B.ins[i].nsrc; s++) if (B.ins[i].vsrc[s] >= 0) last_use[B.ins[i].vsrc[s]] = (int)i;
        for (size_t i = B.ins.size(); i-- > 0;) {
            if (!keep[i] || B.ins[i].dst < 0) continue;
            bool used = false;
            for (size_t j = i + 1; j < B.ins.size() && !used; j++) if (keep[j]) for (int s = 0; s < B.ins[j].nsrc; s++) if (B.ins[j].vsrc[s] == B.ins[i].dst) used = true;
            if (!used) keep[i] = 0;
        }
    }
    std::fill(last_use.begin(), last_use.end(), -1);
    for (size_t i = 0; i < B.ins.size(); i++) if (keep[i]) for (int s = 0; s < B.ins[i].nsrc; s++) if (B.ins[i].vsrc[s] >= 0) last_use[B.ins[i].vsrc[s]] = (int)i;
    std::vector<int> slot_of(nv, -1);
    std::vector<int> free_slots;
    uint32_t n_slots = 0;
    for (size_t i = 0; i < B.ins.size(); i++) {
        if (!keep[i]) continue;
        VIns& v = B.ins[i];
        uint32_t w[3];
        for (int s = 0; s < 3; s++) {
            w[s] = v.src[s];
            if (s < v.nsrc && v.vsrc[s] >= 0) {
                if (slot_of[v.vsrc[s]] < 0) return ctx->fail(ZK_ERR_PROGRAM, "quotient program: intermediate read before it is written");
                w[s] = Builder::enc(K_SLOT, (uint32_t)slot_of[v.vsrc[s]]);
            }
        }
        for (int s = 0; s < v.nsrc; s++)   // operands die here -> their slots may be reused by dst
            if (v.vsrc[s] >= 0 && last_use[v.vsrc[s]] == (int)i && slot_of[v.vsrc[s]] >= 0) {
                bool dup = false;
                for (int q = 0; q < s; q++) if (v.vsrc[q] == v.vsrc[s]) dup = true;
                if (!dup) free_slots.push_back(slot_of[v.vsrc[s]]);
            }
        uint32_t w0 = v.op;
        if (v.dst < 0) w0 |= 1u << 8;
        else {
            int sl;
            if (!free_slots.empty()) { std::sort(free_slots.begin(), free_slots.end(), std::greater<int>()); sl = free_slots.back(); free_slots.pop_back(); }
            else sl = (int)n_slots++;
            slot_of[v.dst] = sl;
            if (sl >= 65536) return ctx->fail(ZK_ERR_LIMIT, "quotient program: too many live intermediates");
            w0 |= (uint32_t)sl << 16;
        }
        P.code.push_back(make_uint4(w0, w[0], w[1], w[2]));
    }
    P.n_slots = n_slots ? n_slots : 1;
    return ZK_OK;
}

int quotient_program_load(zk_ctx* ctx, const void* blob, size_t len, uint64_t* prog) {
    if (!blob || !prog || len < 48 || (len & 3)) return ctx->fail(ZK_ERR_ARG, "zk_quotient_program_load: bad blob pointer/length");
    std::vector<uint32_t> words(len / 4);
    memcpy(words.data(), blob, len);
    std::shared_ptr<QuotProgram> P(new QuotProgram());
    P->device = ctx->device;
    int rc = compile_program(ctx, words.data(), words.size(), *P);
    if (rc) return rc;
    if ((size_t)(P->n_slots > QUOT_NREG ? P->n_slots - QUOT_NREG : 0) * 64 * 32 > 160 * 1024)
        return ctx->fail(ZK_ERR_LIMIT, "quotient program needs %u live intermediates; this build keeps at most 80 in LDS", P->n_slots);
    auto upload = [&](QuotProgram& Q) -> bool {
        hipError_t e = hipMalloc(&Q.d_code, Q.code.size() * 16 + 16);
        if (e == hipSuccess) e = hipMemcpy(Q.d_code, Q.code.data(), Q.code.size() * 16, hipMemcpyHostToDevice);
        return e == hipSuccess;
    };
    if (!upload(*P)) return ctx->fail(ZK_ERR_HIP, "zk_quotient_program_load: device allocation failed");
    // the same program by degree (QuotProgram::part_hi / part_lo): worth it when the extended domain has at least four cosets' worth of rows per low-part coset pair,
    // i.e. cs_degree >= 4, and both parts hold identities
    if (ctx->tune.quot_degree_split && P->degree > SPLIT_LOW_DEGREE && P->ek > P->k) {
        std::shared_ptr<QuotProgram> hi(new QuotProgram()), lo(new QuotProgram());
        hi->device = lo->device = ctx->device;
        if (compile_program(ctx, words.data(), words.size(), *hi, 1) == ZK_OK && compile_program(ctx, words.data(), words.size(), *lo, 2) == ZK_OK &&
            hi->folds_taken && lo->folds_taken && hi->n_slots <= P->n_slots + 8 && lo->n_slots <= P->n_slots + 8) {
            if (!upload(*hi) || !upload(*lo)) return ctx->fail(ZK_ERR_HIP, "zk_quotient_program_load: device allocation failed");
            P->part_hi = hi; P->part_lo = lo;
        }
    }
    // tune quot_jit: the same micro-ops as generated straight-line kernels (quotient_jit.hip).  Asked for and not to be had is an error, not a silent change of executor.
    if (ctx->tune.quot_jit && P->ek > P->k) {
        rc = quot_jit_build(ctx, *P);
        if (rc) return rc;
    }
    *prog = ctx->next_handle++;
    ctx->programs[*prog] = P;
    return ZK_OK;
}
// a handle of `ctx` onto a program another context of the same device loaded: one compiled program (and one proving key) per process, however many
// contexts prove concurrently.  Called WITHOUT either context's lock held (capi.hip): takes the owner's, then ctx's.
int quotient_program_share(zk_ctx* ctx, zk_ctx* owner, uint64_t owner_prog, uint64_t* prog) {
    std::shared_ptr<QuotProgram> P;
    {
        std::lock_guard<std::mutex> lk(owner->mu);
        auto it = owner->programs.find(owner_prog);
        if (it != owner->programs.end()) P = it->second;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!P) return ctx->fail(ZK_ERR_ARG, "zk_quotient_program_share: unknown program %llu", (unsigned long long)owner_prog);
    if (owner->device != ctx->device) return ctx->fail(ZK_ERR_ARG, "zk_quotient_program_share: contexts on different devices");
    *prog = ctx->next_handle++;
    ctx->programs[*prog] = P;
    return ZK_OK;
}

int quotient_program_info(zk_ctx* ctx, uint64_t prog, uint32_t* n_instr, uint32_t* n_slots, uint32_t* n_columns) {
    auto it = ctx->programs.find(prog);
    if (it == ctx->programs.end()) return ctx->fail(ZK_ERR_ARG, "zk_quotient_program_info: unknown program");
    if (n_instr) *n_instr = (uint32_t)it->second->code.size();
    if (n_slots) *n_slots = it->second->n_slots;
    if (n_columns) *n_columns = it->second->n_cols;
    return ZK_OK;
}
// counts[op] = instructions with opcode op (add, sub, mul, sqr, dbl, neg, mov, muladd), counts[8] = memory (column / constant) operands
int quotient_program_kernels(zk_ctx* ctx, uint64_t prog, uint32_t* n_kernels) {
    auto it = ctx->programs.find(prog);
    if (it == ctx->programs.end() || !n_kernels) return ctx->fail(ZK_ERR_ARG, "zk_quotient_program_kernels: unknown program %llu", (unsigned long long)prog);
    const QuotProgram& P = *it->second;
    *n_kernels = quot_jit_kernel_count(P) + (P.part_hi ? quot_jit_kernel_count(*P.part_hi) : 0u) + (P.part_lo ? quot_jit_kernel_count(*P.part_lo) : 0u);
    return ZK_OK;
}
int quotient_program_opmix(zk_ctx* ctx, uint64_t prog, uint32_t part, uint32_t counts[9]) {
    auto it = ctx->programs.find(prog);
    if (it == ctx->programs.end() || !counts || part > 2) return ctx->fail(ZK_ERR_ARG, "zk_quotient_program_opmix: unknown program / null pointer / part");
    if (part && (!it->second->part_hi || !it->second->part_lo)) return ctx->fail(ZK_ERR_ARG, "zk_quotient_program_part_opmix: the program has no degree split");
    for (int i = 0; i < 9; i++) counts[i] = 0;
    const QuotProgram& Q = part == 1 ? *it->second->part_hi : part == 2 ? *it->second->part_lo : *it->second;
    for (const uint4& ins : Q.code) {
        const uint32_t op = ins.x & 0xffu;
        if (op < 8) counts[op]++;
        else if (op == M_FOLD2) counts[M_MULADD]++;        // a fused fold is a multiply-add (two products, one reduction)
        for (uint32_t src : {ins.y, ins.z, ins.w}) { const uint32_t kind = src >> 28; if (kind == K_COL || kind == K_CONST) counts[8]++; }
    }
    return ZK_OK;
}
// the degree split of a program (QuotProgram::part_hi / part_lo): how many cosets its low part is evaluated on (0: the program has none) and the sizes of the two parts
int quotient_program_split(zk_ctx* ctx, uint64_t prog, uint32_t* low_cosets, uint32_t* n_instr_high, uint32_t* n_instr_low) {
    auto it = ctx->programs.find(prog);
    if (it == ctx->programs.end()) return ctx->fail(ZK_ERR_ARG, "zk_quotient_program_split: unknown program");
    const QuotProgram& P = *it->second;
    const bool has = P.part_hi && P.part_lo;
    if (low_cosets) *low_cosets = has ? SPLIT_LOW_DEGREE - 1 : 0;
    if (n_instr_high) *n_instr_high = has ? (uint32_t)P.part_hi->code.size() : 0;
    if (n_instr_low) *n_instr_low = has ? (uint32_t)P.part_lo->code.size() : 0;
    return ZK_OK;
}
int quotient_program_release(zk_ctx* ctx, uint64_t prog) {
    auto it = ctx->programs.find(prog);
    if (it == ctx->programs.end()) return ctx->fail(ZK_ERR_ARG, "zk_quotient_program_release: unknown program %llu", (unsigned long long)prog);
    ctx->programs.erase(it);
    return ZK_OK;
}
void release_programs(zk_ctx* ctx) {
    ctx->programs.clear();
}

int quotient_set_lds_attr() {
#ifndef ZK_EMU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(quotient_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
    return 0;
}

// coset < 0: the whole extended domain (columns of 2^extended_k rows).  coset = j >= 0: only coset j of it — the rows j, j + 2^(ek-k), ... —
// with columns given as that coset's n = 2^k values (zk_coeff_to_coset_batch_dev); rotations then step by one row.  The 2^(ek-k) cosets are
// independent, which is what lets a proof's quotient be split over GPUs: out receives the n numerator values of the coset.
// row_count > 0: only rows [row_lo, row_lo + row_count) of that domain (the columns are complete, so rotations need no halo), out[i] = row row_lo + i —
// the unit that lets more ranks than cosets share a quotient.
// part: 0 = every identity; 1 / 2 = the high / low part of a program that has a degree split (QuotProgram::part_hi / part_lo).  low_cosets > 0 (part 2, coset < 0): the
// columns are the whole extended domain but only the rows of its cosets 0 .. low_cosets-1 are evaluated — thread i of coset j reads row i * 2^(ek-k) + j — and
// out receives low_cosets x n values, coset-major (what zk_cosets_to_pieces_dev takes).
int quotient_run(zk_ctx* ctx, uint64_t prog, const zk_quotient_args* a, int coset, uint64_t row_lo, uint64_t row_count, int part, uint32_t low_cosets) {
    auto it = ctx->programs.find(prog);
    if (it == ctx->programs.end()) return ctx->fail(ZK_ERR_ARG, "zk_quotient_run_dev: unknown program %llu", (unsigned long long)prog);
    if (part && (!it->second->part_hi || !it->second->part_lo)) return ctx->fail(ZK_ERR_ARG, "zk_quotient_run_part_dev: program %llu has no degree split (zk_quotient_program_split)", (unsigned long long)prog);
    QuotProgram& P = part == 1 ? *it->second->part_hi : part == 2 ? *it->second->part_lo : *it->second;
    if (low_cosets && (part != 2 || coset >= 0 || row_count || (low_cosets & (low_cosets - 1)) || low_cosets > (1u << (P.ek - P.k))))
        return ctx->fail(ZK_ERR_ARG, "zk_quotient_run_low_dev: %u cosets of the extended domain: a power of two, at most 2^(extended_k - k), low part only", low_cosets);
    if (!a || !a->out || !a->l0 || !a->l_last || !a->l_active_row || !a->beta || !a->gamma || !a->theta || !a->y)
        return ctx->fail(ZK_ERR_ARG, "zk_quotient_run_dev: null argument");
    if (a->n_sets != P.n_sets) return ctx->fail(ZK_ERR_ARG, "zk_quotient_run_dev: n_sets = %u but the program has %u permutation sets", a->n_sets, P.n_sets);
    if ((P.n_fixed && !a->fixed) || (P.n_advice && !a->advice) || (P.n_instance && !a->instance) || (P.n_perm_cols && !a->perm_cosets) ||
        (P.n_sets && !a->perm_products) || (P.n_lookups && (!a->lookup_product || !a->lookup_input || !a->lookup_table)) ||
        (P.n_challenges && !a->challenges))
        return ctx->fail(ZK_ERR_ARG, "zk_quotient_run_dev: missing column array");
    // column pointer table
    std::vector<const void*> cols(P.n_cols, nullptr);
    for (uint32_t i = 0; i < P.n_fixed; i++) cols[P.col_fixed + i] = a->fixed[i];
    for (uint32_t i = 0; i < P.n_advice; i++) cols[P.col_advice + i] = a->advice[i];
    for (uint32_t i = 0; i < P.n_instance; i++) cols[P.col_instance + i] = a->instance[i];
    cols[P.col_l0] = a->l0; cols[P.col_llast] = a->l_last; cols[P.col_lactive] = a->l_active_row;
    for (uint32_t i = 0; i < P.n_perm_cols; i++) cols[P.col_sigma + i] = a->perm_cosets[i];
    for (uint32_t i = 0; i < P.n_sets; i++) cols[P.col_z + i] = a->perm_products[i];
    for (uint32_t i = 0; i < P.n_lookups; i++) {
        cols[P.col_lk_z + i] = a->lookup_product[i]; cols[P.col_lk_a + i] = a->lookup_input[i]; cols[P.col_lk_s + i] = a->lookup_table[i];
    }
    for (auto p : cols) if (!p) return ctx->fail(ZK_ERR_ARG, "zk_quotient_run_dev: null column pointer");
    // constants of this run
    auto rd = [](const void* p) { u256 o; memcpy(&o, p, 32); return o; };
    std::vector<u256> consts(P.n_consts);
    for (size_t i = 0; i < P.graph_consts.size(); i++) consts[i] = P.graph_consts[i];
    consts[P.c_zero] = Fr::zero(); consts[P.c_one] = Fr::one();
    for (uint32_t i = 0; i < P.n_challenges; i++) consts[P.c_chal + i] = rd((const char*)a->challenges + 32 * i);
    const u256 beta = rd(a->beta);
    consts[P.c_beta] = beta; consts[P.c_gamma] = rd(a->gamma); consts[P.c_theta] = rd(a->theta); consts[P.c_y] = rd(a->y);
    for (size_t i = 0; i < P.ypow_exps.size(); i++) {                 // y^e for the folds that follow skipped identities (square and multiply on the host: a few dozen products)
        u256 acc = Fr::one(), base = consts[P.c_y];
        for (uint32_t e = P.ypow_exps[i]; e; e >>= 1) { if (e & 1) acc = Fr::mul(acc, base); base = Fr::sqr(base); }
        consts[P.c_ypow + i] = acc;
    }
    {   // delta_j = beta * ZETA * DELTA^j  (current_delta of evaluate_h without the omega^idx factor)
        const uint64_t zl[4] = BN254_FR_ZETA_M, dl[4] = BN254_FR_DELTA_M;
        u256 zeta, delta;
        for (int i = 0; i < 8; i++) { zeta.v[i] = (uint32_t)(zl[i >> 1] >> (32 * (i & 1))); delta.v[i] = (uint32_t)(dl[i >> 1] >> (32 * (i & 1))); }
        u256 cur = Fr::mul(beta, zeta);
        for (uint32_t j = 0; j < P.n_perm_cols; j++) { consts[P.c_delta + j] = cur; cur = Fr::mul(cur, delta); }
    }
    if (coset >= (int)(1u << (P.ek - P.k))) return ctx->fail(ZK_ERR_ARG, "zk_quotient_run_coset_dev: coset %d out of range", coset);
    const bool cm = coset >= 0;
    const uint32_t size_log = cm ? P.k : P.ek;
    const uint64_t size = 1ull << size_log;
    const int64_t rot_scale = cm ? 1 : 1ll << (P.ek - P.k);
    std::vector<uint32_t> rot_off(P.rotations.size() + 1, 0);
    for (size_t i = 0; i < P.rotations.size(); i++) {
        int64_t v = ((int64_t)P.rotations[i] * rot_scale) % (int64_t)size;
        if (v < 0) v += (int64_t)size;
        rot_off[i] = (uint32_t)v;
    }
    hipStream_t st = ctx->stream;
    // the run's constants | column pointers | rotation offsets, in this context's own buffer
    const size_t off_cols = ((size_t)P.n_consts * 32 + 32 + 255) & ~(size_t)255, off_rot = (off_cols + (size_t)P.n_cols * sizeof(void*) + 8 + 255) & ~(size_t)255;
    ZK_HIP(ctx->ws_quot.ensure(off_rot + (P.rotations.size() + 1) * 4));
    void* const d_consts = ctx->ws_quot.p;
    void* const d_cols = (char*)ctx->ws_quot.p + off_cols;
    void* const d_rot = (char*)ctx->ws_quot.p + off_rot;
    ZK_HIP(hipMemcpyAsync(d_consts, consts.data(), consts.size() * 32, hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(d_cols, cols.data(), cols.size() * sizeof(void*), hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(d_rot, rot_off.data(), rot_off.size() * 4, hipMemcpyHostToDevice, st));
    QuotArgs q;
    memset(&q, 0, sizeof q);
    q.code = (const uint4*)P.d_code; q.n_instr = (uint32_t)P.code.size(); q.consts = d_consts;
    q.cols = (const void* const*)d_cols; q.rot_off = (const uint32_t*)d_rot; q.size_log = size_log; q.out = a->out;
    q.uses_xpow = P.uses_xpow ? 1 : 0;
    q.xpow_mul = cm ? 1u << (P.ek - P.k) : 1u;
    q.xpow_add = cm ? (uint32_t)coset : 0u;
    if (P.uses_xpow) {
        int rc = ntt_pow_tables(ctx, P.ek, domain_omega(P.ek), &q.tw_lo, &q.tw_hi, &q.lo_bits);
        if (rc) return rc;
    }
    uint64_t rows = size;
    if (low_cosets) {
        rows = (uint64_t)low_cosets << P.k;
        q.sub_log = 0;
        while ((1u << q.sub_log) < low_cosets) q.sub_log++;
        q.stride_log = P.ek - P.k; q.k_log = P.k; q.strided = 1;
    }
    if (row_count) {
        if (row_lo + row_count > size || (row_count & (row_count - 1)) || row_lo % row_count)
            return ctx->fail(ZK_ERR_ARG, "zk_quotient_run_coset_rows_dev: rows [%llu, +%llu) must be an aligned power-of-two slice of the %llu rows",
                             (unsigned long long)row_lo, (unsigned long long)row_count, (unsigned long long)size);
        rows = row_count;
        q.row_base = (uint32_t)row_lo;
    }
    uint32_t T = (uint32_t)std::min(ctx->tune.quot_threads, 256);     // the kernel is compiled for <= 256 threads per workgroup
    if (T > rows) T = (uint32_t)rows;
    const uint32_t lds_slots = P.n_slots > QUOT_NREG ? P.n_slots - QUOT_NREG : 0;
    while (T > 64 && (size_t)lds_slots * T * 32 > 32 * 1024) T >>= 1;
    if (T < 1) T = 1;
    const size_t lds = (size_t)lds_slots * T * 32;
    if (lds > 160 * 1024) return ctx->fail(ZK_ERR_LIMIT, "quotient program needs %zu bytes of LDS", lds);
    EvTimer tq(ctx, "quotient");
    if (quot_jit_ready(P)) {                                          // (the PROGRAM records whether it was generated: contexts that borrow it need no tunable of their own)
        uint32_t Tj = (uint32_t)std::min(ctx->tune.quot_threads, 256);
        if (Tj > rows) Tj = (uint32_t)rows;
        int rcj = quot_jit_launch(ctx, P, q, rows, Tj);
        if (rcj) return rcj;
    } else {
        ZK_LAUNCH(quotient_kernel, (uint32_t)(rows / T), T, lds, st, q);
        ZK_CHECK_LAUNCH();
    }
    tq.stop();
    ZK_HIP(hipStreamSynchronize(st));
    tq.resolve();
    // every column once + the output, per row of the extended domain — counted once per evaluation of h's numerator (on its high part when the program is split; the
    // theta-compression programs of the lookups, which run on this interpreter too, are not part of that figure though their time is inside the "quotient" timer)
    if (ctx->timing && P.ek > P.k && part != 2) ctx->last_ms["quotient_alg_bytes"] += (double)rows * (P.n_cols + 1) * 32.0;
    return ZK_OK;
}


// ------------------------------------------------------------------------------------------------
// proving-key level entry points: the shape of halo2's own call (polynomials in, h(X) out)
// ------------------------------------------------------------------------------------------------
struct PkData {
    uint64_t prog = 0;
    std::vector<void*> fixed, sigma;           // extended cosets, device
    void* l[3] = {nullptr, nullptr, nullptr};   // l0, l_last, l_active_row
    // per-proof workspace (allocated once): staging for coefficient uploads + extended cosets of the proof's polys
    std::vector<void*> dyn_ext;
    void* stage = nullptr;
    void* h_ext = nullptr;
    std::vector<void*> lv;                      // (pk_load: the three l columns on their way into `l`)
    ~PkData() {
        for (void* p : fixed) if (p) (void)hipFree(p);
        for (void* p : sigma) if (p) (void)hipFree(p);
        for (void* p : dyn_ext) if (p) (void)hipFree(p);
        for (void* p : lv) if (p) (void)hipFree(p);
        for (void* p : l) if (p) (void)hipFree(p);
        if (stage) (void)hipFree(stage);
        if (h_ext) (void)hipFree(h_ext);
    }
};
static std::map<uint64_t, PkData*> g_pks;       // keyed by handle (handles are unique per process)
static std::mutex g_pk_mu;

static void pk_free(PkData* pk) { delete pk; }

// host columns -> extended cosets on the device.  form 0: n coefficients each; form 1: 2^ek coset values each
static int pk_upload_cols(zk_ctx* ctx, const QuotProgram& P, const void* const* cols, size_t count, int form, std::vector<void*>& out) {
    const size_t nb = (size_t)32 << P.k, eb = (size_t)32 << P.ek;
    out.assign(count, nullptr);
    if (count == 0) return ZK_OK;
    for (size_t i = 0; i < count; i++) {
        if (!cols[i]) return ctx->fail(ZK_ERR_ARG, "zk_pk_load: null column %zu", i);
        ZK_HIP(hipMalloc(&out[i], eb));
    }
    if (form == 1) {
        for (size_t i = 0; i < count; i++) ZK_HIP(hipMemcpyAsync(out[i], cols[i], eb, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
        return ZK_OK;
    }
    ZK_HIP(ctx->ws_scalars.ensure(count * nb));
    std::vector<const void*> src(count);
    for (size_t i = 0; i < count; i++) {
        src[i] = (char*)ctx->ws_scalars.p + i * nb;
        ZK_HIP(hipMemcpyAsync((void*)src[i], cols[i], nb, hipMemcpyHostToDevice, ctx->stream));
    }
    int rc = domain_coeff_to_extended_batch(ctx, src.data(), out.data(), count, P.k, P.ek);
    if (rc) return rc;
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

int pk_load(zk_ctx* ctx, uint64_t prog, const void* const* fixed, const void* const* sigma, const void* l0, const void* l_last, const void* l_active,
            int form, uint64_t* handle) {
    auto it = ctx->programs.find(prog);
    if (it == ctx->programs.end()) return ctx->fail(ZK_ERR_ARG, "zk_pk_load: unknown program %llu", (unsigned long long)prog);
    const QuotProgram& P = *it->second;
    if (!handle || !l0 || !l_last || !l_active || (P.n_fixed && !fixed) || (P.n_perm_cols && !sigma) || (form != 0 && form != 1))
        return ctx->fail(ZK_ERR_ARG, "zk_pk_load: null / bad argument");
    std::unique_ptr<PkData> pk(new PkData());                        // (every way out but the last frees what was uploaded)
    pk->prog = prog;
    int rc = pk_upload_cols(ctx, P, fixed, P.n_fixed, form, pk->fixed);
    if (!rc) rc = pk_upload_cols(ctx, P, sigma, P.n_perm_cols, form, pk->sigma);
    const void* ls[3] = {l0, l_last, l_active};
    if (!rc) rc = pk_upload_cols(ctx, P, ls, 3, form, pk->lv);
    if (rc) return rc;
    for (int i = 0; i < 3; i++) pk->l[i] = pk->lv[i];
    pk->lv.clear();
    const size_t ndyn = (size_t)P.n_advice + P.n_instance + P.n_sets + 3 * (size_t)P.n_lookups;
    pk->dyn_ext.assign(ndyn, nullptr);
    hipError_t e = hipSuccess;
    for (size_t i = 0; i < ndyn && e == hipSuccess; i++) e = hipMalloc(&pk->dyn_ext[i], (size_t)32 << P.ek);
    if (e == hipSuccess) e = hipMalloc(&pk->stage, std::max<size_t>(ndyn, 1) * ((size_t)32 << P.k));
    if (e == hipSuccess) e = hipMalloc(&pk->h_ext, (size_t)32 << P.ek);
    if (e != hipSuccess) return ctx->fail(ZK_ERR_HIP, "zk_pk_load: device allocation failed");
    std::lock_guard<std::mutex> lk(g_pk_mu);
    g_pks[((uint64_t)(uintptr_t)ctx << 20) ^ ctx->next_handle] = pk.get();
    (void)pk.release();
    *handle = ctx->next_handle++;
    return ZK_OK;
}
static PkData* pk_find(zk_ctx* ctx, uint64_t h) {
    std::lock_guard<std::mutex> lk(g_pk_mu);
    auto it = g_pks.find(((uint64_t)(uintptr_t)ctx << 20) ^ h);
    return it == g_pks.end() ? nullptr : it->second;
}
int pk_release(zk_ctx* ctx, uint64_t h) {
    std::lock_guard<std::mutex> lk(g_pk_mu);
    auto it = g_pks.find(((uint64_t)(uintptr_t)ctx << 20) ^ h);
    if (it == g_pks.end()) return ctx->fail(ZK_ERR_ARG, "zk_pk_release: unknown handle");
    pk_free(it->second);
    g_pks.erase(it);
    return ZK_OK;
}
void release_pks(zk_ctx* ctx) {   // called from zk_ctx_destroy: free what this context still owns
    std::lock_guard<std::mutex> lk(g_pk_mu);
    const uint64_t tag = (uint64_t)(uintptr_t)ctx << 20;
    for (auto it = g_pks.begin(); it != g_pks.end();) {
        if (((it->first ^ tag) >> 20) == 0) { pk_free(it->second); it = g_pks.erase(it); }
        else ++it;
    }
}

// Evaluator::evaluate_h (+ optionally the divide / extended_to_coeff of vanishing::Argument::construct):
// host coefficient-form polynomials in, host result out.
int evaluate_h_host(zk_ctx* ctx, uint64_t pkh, const void* const* advice, const void* const* instance, const void* const* perm_products,
                    const void* const* lk_product, const void* const* lk_input, const void* const* lk_table, const void* challenges,
                    const void* beta, const void* gamma, const void* theta, const void* y, int finish, void* out) {
    PkData* pk = pk_find(ctx, pkh);
    if (!pk) return ctx->fail(ZK_ERR_ARG, "zk_evaluate_h: unknown pk handle");
    auto it = ctx->programs.find(pk->prog);
    if (it == ctx->programs.end()) return ctx->fail(ZK_ERR_ARG, "zk_evaluate_h: the pk's program was released");
    const QuotProgram& P = *it->second;
    if (!out) return ctx->fail(ZK_ERR_ARG, "zk_evaluate_h: null output");
    const size_t nb = (size_t)32 << P.k;
    // gather the proof's polynomials in the order of pk->dyn_ext: advice | instance | perm products | lookup z | a' | s'
    struct Grp { const void* const* cols; size_t n; const char* what; };
    const Grp groups[] = {{advice, P.n_advice, "advice"}, {instance, P.n_instance, "instance"}, {perm_products, P.n_sets, "permutation product"},
                          {lk_product, P.n_lookups, "lookup product"}, {lk_input, P.n_lookups, "lookup permuted input"}, {lk_table, P.n_lookups, "lookup permuted table"}};
    std::vector<const void*> src;
    size_t idx = 0;
    for (const Grp& g : groups) {
        if (g.n && !g.cols) return ctx->fail(ZK_ERR_ARG, "zk_evaluate_h: missing %s polynomials", g.what);
        for (size_t i = 0; i < g.n; i++, idx++) {
            if (!g.cols[i]) return ctx->fail(ZK_ERR_ARG, "zk_evaluate_h: null %s polynomial %zu", g.what, i);
            void* d = (char*)pk->stage + idx * nb;
            ZK_HIP(hipMemcpyAsync(d, g.cols[i], nb, hipMemcpyHostToDevice, ctx->stream));
            src.push_back(d);
        }
    }
    if (!src.empty()) {
        int rc = domain_coeff_to_extended_batch(ctx, src.data(), pk->dyn_ext.data(), src.size(), P.k, P.ek);
        if (rc) return rc;
    }
    zk_quotient_args qa;
    ZK_STRUCT_INIT(qa);
    void* const* dyn = pk->dyn_ext.data();
    qa.fixed = pk->fixed.data(); qa.advice = dyn; qa.instance = dyn + P.n_advice;
    qa.l0 = pk->l[0]; qa.l_last = pk->l[1]; qa.l_active_row = pk->l[2];
    qa.perm_cosets = pk->sigma.data(); qa.perm_products = dyn + P.n_advice + P.n_instance; qa.n_sets = P.n_sets;
    qa.lookup_product = dyn + P.n_advice + P.n_instance + P.n_sets;
    qa.lookup_input = qa.lookup_product + P.n_lookups; qa.lookup_table = qa.lookup_input + P.n_lookups;
    qa.challenges = challenges; qa.beta = beta; qa.gamma = gamma; qa.theta = theta; qa.y = y; qa.out = pk->h_ext;
    int rc = quotient_run(ctx, pk->prog, &qa, -1, 0, 0, 0, 0);
    if (rc) return rc;
    size_t out_bytes = (size_t)32 << P.ek;
    if (finish) {
        rc = domain_divide_by_vanishing(ctx, pk->h_ext, P.k, P.ek);
        if (!rc) rc = domain_extended_to_coeff(ctx, pk->h_ext, P.k, P.ek);
        if (rc) return rc;
        out_bytes = nb * (P.degree - 1);
    }
    ZK_HIP(hipMemcpyAsync(out, pk->h_ext, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

}  // namespace zk
