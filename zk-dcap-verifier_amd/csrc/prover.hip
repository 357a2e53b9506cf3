// Native create_proof: halo2_proofs::plonk::create_proof + ProverSHPLONK (zkwebauthn/halo2 @ c254c75, Cargo.lock:1314-1327) as the reference calls
// them at circuits/src/sgx_dcap_verifier.rs:814-822 — the per-proof path of a phase-batched, HBM-resident `plonk/prover.rs`, written in C++ because the
// reference's host side is compiled code (Rust) and no Rust toolchain exists in the build image.  It is a CLIENT of the C ABI (include/zkmi355.h): every
// O(n) step is one of the zk_* entry points the Rust prover would call, in the order of INTEGRATION.md's phase table; what stays on the host is what
// stays on the host in the reference — Fiat-Shamir hashing (Blake2b, src/transcript.rs), point encoding, rotation-set bookkeeping and the O(#points^2)
// interpolations of SHPLONK.  zk-dcap-verifier_amd/plonk/prover.py + shplonk.py are the Python twin (phase by phase, draw by draw): both must emit the
// bytes of the independent CPU prover's goldens (tests/test_native_prover.py).  Single circuit instance, no user challenges, Blake2b transcript (stack A).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "field.cuh"
#include "../../include/zkmi355.h"
#include "abi_guard.h"

using namespace zk;

int zk_internal_fail(zk_ctx* ctx, int code, const char* msg);   // capi.hip: sets zk_last_error(ctx)
zk_ctx* zk_internal_helper_ctx(zk_ctx* ctx);                      // capi.hip: the helper context of ctx (ctx.h), or null
void zk_internal_trim_helper(zk_ctx* ctx);                        // capi.hip: give the helper context's grow-only device memory back (zk_plonk_trim)

namespace {
int pk_fail(zk_ctx* ctx, int code, const char* fmt, ...) {
    char buf[384];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return zk_internal_fail(ctx, code, buf);
}

// ---- Blake2b-512 with personalisation (RFC 7693), incremental, copyable ------------------------------------------------------------------------------
struct Blake2b {
    uint64_t h[8], t = 0;
    uint8_t buf[128];
    size_t len = 0;
    static constexpr uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                                       0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    explicit Blake2b(const char person[16]) {
        for (int i = 0; i < 8; i++) h[i] = IV[i];
        h[0] ^= 0x01010000ull ^ 64;                                  // digest length 64, no key, fanout = depth = 1
        uint64_t p0, p1;
        memcpy(&p0, person, 8); memcpy(&p1, person + 8, 8);
        h[6] ^= p0; h[7] ^= p1;                                      // parameter block bytes 48..63
    }
    static uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
    void compress(const uint8_t* block, bool last) {
        static const uint8_t S[12][16] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
                                          {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4},   {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
                                          {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13},   {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
                                          {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11},   {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
                                          {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5},   {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
                                          {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
        uint64_t m[16], v[16];
        memcpy(m, block, 128);
        for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; }
        v[12] ^= t;
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
            v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32); v[c] = v[c] + v[d]; v[b] = rotr(v[b] ^ v[c], 24);
            v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = rotr(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; r++) {
            const uint8_t* s = S[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]); G(1, 5, 9, 13, m[s[2]], m[s[3]]); G(2, 6, 10, 14, m[s[4]], m[s[5]]); G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]); G(1, 6, 11, 12, m[s[10]], m[s[11]]); G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
    }
    void update(const void* data, size_t n) {
        const uint8_t* p = (const uint8_t*)data;
        while (n) {
            if (len == 128) { t += 128; compress(buf, false); len = 0; }        // a full buffer is only compressed when more input follows
            const size_t take = std::min(n, 128 - len);
            memcpy(buf + len, p, take);
            len += take; p += take; n -= take;
        }
    }
    void digest(uint8_t out[64]) const {                              // of a copy: the state keeps absorbing afterwards (Blake2bWrite clones to squeeze)
        Blake2b c = *this;
        c.t += c.len;
        memset(c.buf + c.len, 0, 128 - c.len);
        c.compress(c.buf, true);
        memcpy(out, c.h, 64);
    }
};
constexpr uint64_t Blake2b::IV[8];

// ---- host field helpers (Montgomery u256 over Fr / Fq from field.cuh) ---------------------------------------------------------------------------------
using Fe = u256;
inline Fe fe_from_u64(uint64_t v) { u256 x = Fr::zero(); x.v[0] = (uint32_t)v; x.v[1] = (uint32_t)(v >> 32); return Fr::to_mont(x); }
inline Fe fe_pow_u64(Fe a, uint64_t e) { Fe r = Fr::one(); while (e) { if (e & 1) r = Fr::mul(r, a); a = Fr::sqr(a); e >>= 1; } return r; }
inline bool canon_less(const u256& a, const u256& b) { for (int i = 7; i >= 0; i--) if (a.v[i] != b.v[i]) return a.v[i] < b.v[i]; return false; }
inline u256 load32(const void* p) { u256 o; memcpy(&o, p, 32); return o; }
struct CanonLess { bool operator()(const u256& a, const u256& b) const { return canon_less(a, b); } };

// ---- Keccak-256 (the original padding 0x01, as the EVM's KECCAK256), one shot --------------------------------------------------------------------------------
inline void keccak256(const uint8_t* data, size_t n, uint8_t out[32]) {
    static const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
                                    0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
                                    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
                                    0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};    // [x + 5 y]
    uint64_t a[25] = {0};
    auto permute = [&]() {
        for (int rd = 0; rd < 24; rd++) {
            uint64_t c[5], d[5], b[25];
            for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
            for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ ((c[(x + 1) % 5] << 1) | (c[(x + 1) % 5] >> 63));
            for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
            for (int x = 0; x < 5; x++)
                for (int y = 0; y < 5; y++) {
                    const int r = ROT[x + 5 * y];
                    const uint64_t v = a[x + 5 * y];
                    b[y + 5 * ((2 * x + 3 * y) % 5)] = r ? (v << r) | (v >> (64 - r)) : v;
                }
            for (int y = 0; y < 5; y++)
                for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
            a[0] ^= RC[rd];
        }
    };
    const size_t rate = 136;
    std::vector<uint8_t> m(data, data + n);
    m.push_back(0x01);
    while (m.size() % rate) m.push_back(0);
    m.back() |= 0x80;
    for (size_t off = 0; off < m.size(); off += rate) {
        for (size_t i = 0; i < rate / 8; i++) { uint64_t w; memcpy(&w, &m[off + 8 * i], 8); a[i] ^= w; }
        permute();
    }
    memcpy(out, a, 32);
}

// ---- Poseidon over Fr as snark-verifier's transcript uses it: T = 3, RATE = 2, R_F = 8, R_P = 57; constants from the Grain LFSR of the Poseidon paper ----------------
struct PoseidonSpec {
    static constexpr int T = 3, RATE = 2, RF = 8, RP = 57, BITS = 254;
    Fe rc[RF + RP][T], mds[T][T];
    PoseidonSpec() {
        uint8_t st[80];
        int pos = 0;
        auto put = [&](uint32_t value, int len) { for (int i = 0; i < len; i++) st[pos++] = (value >> (len - 1 - i)) & 1; };      // MSB first
        put(1, 2); put(0, 4); put(BITS, 12); put(T, 12); put(RF, 10); put(RP, 10);                                                  // prime field, x^alpha s-box
        for (int i = 0; i < 30; i++) st[pos++] = 1;
        auto raw = [&]() { const uint8_t nb = st[62] ^ st[51] ^ st[38] ^ st[23] ^ st[13] ^ st[0]; memmove(st, st + 1, 79); st[79] = nb; return nb; };
        for (int i = 0; i < 160; i++) raw();
        auto bit = [&]() { for (;;) { const uint8_t b1 = raw(), b2 = raw(); if (b1) return b2; } };                                // self-shrinking
        auto integer = [&]() { u256 v = Fr::zero(); for (int i = 0; i < BITS; i++) { for (int l = 7; l > 0; l--) v.v[l] = (v.v[l] << 1) | (v.v[l - 1] >> 31); v.v[0] = (v.v[0] << 1) | bit(); } return v; };
        for (int r = 0; r < RF + RP; r++)
            for (int i = 0; i < T; i++) {
                u256 v;
                do v = integer(); while (!Fr::eq(Fr::reduce_once(v), v));                  // round constants: rejection sampling below r
                rc[r][i] = Fr::to_mont(v);
            }
        for (;;) {                                                                          // MDS: Cauchy matrix 1 / (x_i + y_j) of 2 T distinct samples (taken mod r)
            Fe v[2 * T];
            bool distinct = true;
            for (int i = 0; i < 2 * T; i++) {
                v[i] = Fr::to_mont(Fr::reduce_once(integer()));
                for (int j = 0; j < i; j++) distinct &= !Fr::eq(v[i], v[j]);
            }
            if (!distinct) continue;
            for (int i = 0; i < T; i++)
                for (int j = 0; j < T; j++) mds[i][j] = Fr::inv(Fr::add(v[i], v[T + j]));
            break;
        }
    }
    void permute(Fe s[T]) const {
        for (int r = 0; r < RF + RP; r++) {
            for (int i = 0; i < T; i++) s[i] = Fr::add(s[i], rc[r][i]);
            const bool full = r < RF / 2 || r >= RF / 2 + RP;
            for (int i = 0; i < (full ? T : 1); i++) { const Fe x2 = Fr::sqr(s[i]); s[i] = Fr::mul(Fr::sqr(x2), s[i]); }
            Fe o[T];
            for (int i = 0; i < T; i++) { o[i] = Fr::zero(); for (int j = 0; j < T; j++) o[i] = Fr::add(o[i], Fr::mul(mds[i][j], s[j])); }
            for (int i = 0; i < T; i++) s[i] = o[i];
        }
    }
};
const PoseidonSpec& poseidon_spec() { static const PoseidonSpec spec; return spec; }

// The transcript of create_proof, three flavours (zk_plonk_pk_desc.transcript):
//   0  Blake2bWrite<_, G1Affine, Challenge255<_>>                      stack A (sgx_dcap_verifier.rs:813): points compressed with the y-parity flag in bit 255
//   1  snark-verifier PoseidonTranscript<G1Affine, NativeLoader, _>    stack B gen_proof (base.rs:200-212): a point is absorbed as its coordinates taken mod r,
//                                                                       a squeeze is one sponge squeeze; points compressed with the flag in bit 254 (halo2curves-axiom)
//   2  snark-verifier EvmTranscript<G1Affine, NativeLoader, _, _>      stack B gen_evm_proof_shplonk (base.rs:193-199): 32-byte BIG-endian words, Keccak-256
struct Transcript {
    int kind = 0;
    bool bad_point = false;                                           // no flavour can absorb the identity (halo2's common_point: "cannot write points at infinity to the transcript"): create_proof returns ZK_ERR_ARG
    Blake2b st{"Halo2-Transcript"};
    Fe sponge[3];
    std::vector<Fe> pending;
    std::vector<uint8_t> evm;
    std::vector<uint8_t> out;
    explicit Transcript(int kind_) : kind(kind_) {
        sponge[0] = Fr::to_mont([] { u256 x = Fr::zero(); x.v[2] = 1; return x; }());     // 2^64
        sponge[1] = sponge[2] = Fr::zero();
    }
    static void be32(const u256& c, uint8_t o[32]) { for (int i = 0; i < 32; i++) o[i] = (uint8_t)(c.v[(31 - i) >> 2] >> (8 * ((31 - i) & 3))); }
    void absorb(const Fe* chunk, int n) {
        for (int i = 0; i < n; i++) sponge[1 + i] = Fr::add(sponge[1 + i], chunk[i]);
        if (n < PoseidonSpec::RATE) sponge[1 + n] = Fr::add(sponge[1 + n], Fr::one());
        poseidon_spec().permute(sponge);
    }
    Fe squeeze() {
        if (kind == 1) {
            std::vector<Fe> buf;
            buf.swap(pending);
            for (size_t i = 0; i < buf.size(); i += PoseidonSpec::RATE) absorb(&buf[i], (int)std::min<size_t>(PoseidonSpec::RATE, buf.size() - i));
            if (buf.size() % PoseidonSpec::RATE == 0) absorb(nullptr, 0);
            return sponge[1];
        }
        if (kind == 2) {
            std::vector<uint8_t> data = evm;
            if (evm.size() == 32) data.push_back(0x01);
            uint8_t h[32];
            keccak256(data.data(), data.size(), h);
            evm.assign(h, h + 32);
            u256 v;
            for (int i = 0; i < 32; i++) ((uint8_t*)v.v)[i] = h[31 - i];                   // big-endian integer -> little-endian limbs
            return Fr::to_mont(v);                                                         // (v < 2^256: the Montgomery product reduces it mod r)
        }
        const uint8_t pre = 0;                                        // Challenge255: the 64-byte digest as a little-endian integer mod r
        st.update(&pre, 1);
        uint8_t d[64];
        st.digest(d);
        const u256 lo = load32(d), hi = load32(d + 32), r2 = Fr::R2();
        return Fr::add(Fr::mul(lo, r2), Fr::mul(Fr::mul(hi, r2), r2));
    }
    void common_scalar(const Fe& s) {
        if (kind == 1) { pending.push_back(s); return; }
        const u256 c = Fr::from_mont(s);
        if (kind == 2) { uint8_t b[32]; be32(c, b); evm.insert(evm.end(), b, b + 32); return; }
        const uint8_t pre = 2;
        st.update(&pre, 1); st.update(c.v, 32);
    }
    void write_scalar(const Fe& s) {
        common_scalar(s);
        const u256 c = Fr::from_mont(s);
        if (kind == 2) { uint8_t b[32]; be32(c, b); out.insert(out.end(), b, b + 32); return; }
        out.insert(out.end(), (const uint8_t*)c.v, (const uint8_t*)c.v + 32);
    }
    void write_point(const uint64_t jac[12]) {                        // normalised {x, y, z}: z = mont(1), or all zero for the identity
        u256 x = Fq::zero(), y = Fq::zero();
        bool ident = true;
        for (int i = 8; i < 12; i++) ident &= jac[i] == 0;
        if (!ident) { x = Fq::from_mont(load32(jac)); y = Fq::from_mont(load32(jac + 4)); }
        if (ident) { bad_point = true; return; }
        if (kind == 1) {
            pending.push_back(Fr::to_mont(Fr::reduce_once(x)));       // fe_to_fe: the coordinate as an integer, mod r (q < 2 r)
            pending.push_back(Fr::to_mont(Fr::reduce_once(y)));
            uint8_t enc[32];
            memcpy(enc, x.v, 32);
            enc[31] |= (uint8_t)((y.v[0] & 1) << 6);
            out.insert(out.end(), enc, enc + 32);
            return;
        }
        if (kind == 2) {
            uint8_t b[64];
            be32(x, b); be32(y, b + 32);
            evm.insert(evm.end(), b, b + 64);
            out.insert(out.end(), b, b + 64);
            return;
        }
        const uint8_t pre = 1;
        st.update(&pre, 1); st.update(x.v, 32); st.update(y.v, 32);
        uint8_t enc[32];
        memcpy(enc, x.v, 32);
        enc[31] |= (uint8_t)((y.v[0] & 1) << 7);
        out.insert(out.end(), enc, enc + 32);
    }
};

// ---- device memory of one proof: size-keyed free lists kept per context across proofs ------------------------------------------------------------------
struct Pool {
    std::mutex mu;
    std::map<size_t, std::vector<void*>> free_;
    std::vector<std::vector<uint64_t>> draw_bufs;                     // host buffers of the rng draws of finished proofs (one per proof in flight on the context), reused: a proof
                                                                      // draws n + O(columns) field elements — 16 MiB at k = 19 — and fresh pages every proof cost mmap churn
    bool retired = false;                                             // zk_plonk_trim ran while a proof of this context still held buffers: they are freed as they come back
};
std::mutex g_pools_mu;
std::map<zk_ctx*, std::shared_ptr<Pool>> g_pools;
std::shared_ptr<Pool> pool_of(zk_ctx* ctx) {
    std::lock_guard<std::mutex> lk(g_pools_mu);
    std::shared_ptr<Pool>& p = g_pools[ctx];
    if (!p) p = std::make_shared<Pool>();
    return p;
}
struct Arena {                                                        // everything a proof allocates goes back to the pool when it ends
    zk_ctx* ctx; std::shared_ptr<Pool> pool; std::vector<std::pair<void*, size_t>> held;
    explicit Arena(zk_ctx* c) : ctx(c), pool(pool_of(c)) {}
    void* get(size_t bytes) {
        void* p = nullptr;
        {
            std::lock_guard<std::mutex> lk(pool->mu);
            auto& v = pool->free_[bytes];
            if (!v.empty()) { p = v.back(); v.pop_back(); }
        }
        if (!p && zk_dev_alloc(ctx, bytes, &p) != ZK_OK) return nullptr;
        try { held.push_back({p, bytes}); } catch (...) { put(p, bytes); throw; }
        return p;
    }
    void put(void* p, size_t bytes) noexcept {                         // (runs in destructors: a buffer the free list cannot take is freed instead)
        bool keep = false;
        try { std::lock_guard<std::mutex> lk(pool->mu); if (!pool->retired) { pool->free_[bytes].push_back(p); keep = true; } } catch (...) {}
        if (!keep) (void)zk_dev_free(ctx, p);
    }
    void give_back(void* p) {
        for (auto& h : held) if (h.first == p) { put(p, h.second); h.first = nullptr; return; }
    }
    ~Arena() { for (auto& h : held) if (h.first) put(h.first, h.second); }
};

// ---- transforms that need no challenge, beside the commitments that do ------------------------------------------------------------------------------------------------
// A proof alone on the GPU spends a third of phases 2-5 in the MSM's sort, its reduction tail and the host's folds — latency, not arithmetic — and every commitment waits for
// the transcript.  The coefficient and extended forms of a phase's columns depend on no challenge: as soon as a phase's values are final they go to the context's HELPER
// context (own stream, workspaces, lock: ctx.h) from a helper host thread, jobs in submission order, while the proof's own thread commits the same columns; phase 6 waits
// for them instead of transforming everything at once.  Same field elements, same bytes.
struct SideLane {
    zk_ctx* h = nullptr;
    std::thread th;
    std::mutex mu; std::condition_variable cv;
    std::deque<std::function<int()>> q;
    size_t open = 0; bool closing = false; int rc = ZK_OK;
    void start(zk_ctx* helper) {
        h = helper;
        fault_thread_tick();
        th = std::thread([this]() {
            std::unique_lock<std::mutex> lk(mu);
            while (true) {
                cv.wait(lk, [&] { return closing || !q.empty(); });
                if (q.empty()) break;
                std::function<int()> f = std::move(q.front());
                q.pop_front();
                const int before = rc;
                lk.unlock();
                int r = before;                                       // (after an error the remaining jobs are dropped)
                if (!r) { try { r = f(); } catch (...) { r = abi_exception(h, "zk_plonk_create_proof (helper thread)"); } }      // nothing may leave a thread's entry function: it becomes the lane's rc
                lk.lock();
                if (r && !rc) rc = r;
                open--;
                cv.notify_all();
            }
        });
    }
    void submit(std::function<int()> f) { { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); open++; } cv.notify_all(); }
    int wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return open == 0; }); return rc; }
    ~SideLane() {
        if (!th.joinable()) return;
        { std::lock_guard<std::mutex> lk(mu); closing = true; }
        cv.notify_all();
        th.join();                                                     // (jobs in flight finish: they write buffers of the proof's arena, which outlives this object)
    }
};

// ---- the caller's Fr::random draws, made on a helper thread in the order the phases consume them ---------------------------------------------------------
struct Draws {
    std::vector<uint64_t> buf;                                        // all items back to back (count x 4 limbs each), borrowed from the context's pool
    std::vector<size_t> counts, offs;
    std::shared_ptr<Pool> pool;
    std::mutex mu; std::condition_variable cv; size_t done = 0;
    std::thread th;
    std::atomic<bool> abandoned{false};                                // the proof ended early (an error): stop asking the caller for randomness nobody will use
    int failed = 0;                                                    // the helper thread could not get its buffer (mu): every take() from then on throws the code to the proof's thread
    void start(zk_rng_fn rng, void* user, std::shared_ptr<Pool> p) {
        pool = std::move(p);
        size_t total = 4;
        for (size_t c : counts) { offs.push_back(total); total += c * 4; }
        {
            std::lock_guard<std::mutex> lk(pool->mu);
            if (!pool->draw_bufs.empty()) { buf.swap(pool->draw_bufs.back()); pool->draw_bufs.pop_back(); }
        }
        fault_thread_tick();
        th = std::thread([this, rng, user, total]() {
            try {
                if (buf.size() < total) buf.resize(total);             // (first proof of a context only; on the helper thread, off the proof's critical path)
            } catch (...) {
                { std::lock_guard<std::mutex> lk(mu); failed = abi_exception(nullptr, "zk_plonk_create_proof (rng thread)"); }
                cv.notify_all();
                return;
            }
            for (size_t i = 0; i < counts.size() && !abandoned.load(); i++) {
                if (counts[i]) rng(user, counts[i], buf.data() + offs[i]);
                { std::lock_guard<std::mutex> lk(mu); done = i + 1; }
                cv.notify_all();
            }
        });
    }
    const uint64_t* take(size_t i) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done > i || failed; });
        if (failed) throw AbiError{failed, "the rng helper thread could not allocate its draw buffer"};
        return buf.data() + offs[i];
    }
    void finish() { if (!counts.empty()) (void)take(counts.size() - 1); }     // a successful proof leaves the caller's stream where halo2 would: every planned draw made
    ~Draws() {
        abandoned.store(true);
        if (th.joinable()) th.join();
        try {
            if (pool && !buf.empty()) { std::lock_guard<std::mutex> lk(pool->mu); if (!pool->retired && pool->draw_bufs.size() < 8) pool->draw_bufs.emplace_back(std::move(buf)); }
        } catch (...) {}                                              // (a destructor: the buffer is simply not kept)
    }
};

struct Query { const void* poly; Fe point; Fe eval; };                 // ProverQuery { point, poly } + its evaluation

// coefficients of the Lagrange basis polynomials of a point set: every commitment of a rotation set is interpolated over the SAME points, so the products
// and the field inversions (one Fermat exponentiation each on the host) are done once per set
std::vector<std::vector<Fe>> lagrange_basis(const std::vector<Fe>& pts) {
    const size_t n = pts.size();
    std::vector<std::vector<Fe>> basis(n);
    for (size_t j = 0; j < n; j++) {
        std::vector<Fe> num{Fr::one()};
        Fe den = Fr::one();
        for (size_t m = 0; m < n; m++) {
            if (m == j) continue;
            std::vector<Fe> nx(num.size() + 1);
            nx[0] = Fr::neg(Fr::mul(pts[m], num[0]));
            for (size_t i = 1; i < num.size(); i++) nx[i] = Fr::sub(num[i - 1], Fr::mul(pts[m], num[i]));
            nx[num.size()] = num.back();
            num.swap(nx);
            den = Fr::mul(den, Fr::sub(pts[j], pts[m]));
        }
        const Fe sc = Fr::inv(den);
        basis[j].resize(n);
        for (size_t i = 0; i < n; i++) basis[j][i] = Fr::mul(num[i], sc);
    }
    return basis;
}
std::vector<Fe> interpolate_with_basis(const std::vector<std::vector<Fe>>& basis, const std::vector<Fe>& evals) {
    const size_t n = basis.size();
    std::vector<Fe> coeffs(n, Fr::zero());
    for (size_t j = 0; j < n; j++) for (size_t i = 0; i < n; i++) coeffs[i] = Fr::add(coeffs[i], Fr::mul(evals[j], basis[j][i]));
    return coeffs;
}
Fe eval_small(const std::vector<Fe>& c, const Fe& x) { Fe acc = Fr::zero(); for (size_t i = c.size(); i-- > 0;) acc = Fr::add(Fr::mul(acc, x), c[i]); return acc; }
Fe vanishing_at(const std::vector<Fe>& roots, const Fe& z) { Fe acc = Fr::one(); for (auto& r : roots) acc = Fr::mul(acc, Fr::sub(z, r)); return acc; }

#define PK(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

thread_local double g_phase_ms[9];                                    // wall time of the phases of the calling thread's last proof (zk_plonk_last_phase_ms)
struct PhaseClock {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(int phase) {
        const auto now = std::chrono::steady_clock::now();
        g_phase_ms[phase] += std::chrono::duration<double, std::milli>(now - t).count();
        t = now;
    }
};

}  // namespace

// One proof over several ranks: the ranks run in lockstep from collective to collective, so a rank that fails on its own (out of memory, a HIP error, its witness
// outside a lookup table, an rng callback error) must not simply return — the others would wait in the next all-gather for ever.  It enters that NEXT exchange once more
// with a poisoned block (first 32 bytes 0xFF: no field element and no point coordinate has that value) and returns its error; every other rank finds the mark in
// the gathered blocks and returns ZK_ERR_COMM from the same exchange.  Not covered: a failing collective itself, and a rank that cannot even allocate its exchange
// buffers — the callback must enforce a timeout for those (include/zkmi355.h, zk_allgather_fn).
struct ShardSignal {
    std::vector<size_t> sizes;                                         // bytes of every exchange of this proof, in order (the same list on every rank)
    size_t next = 0;                                                   // the exchange every healthy rank enters next
    void* xsend = nullptr; void* xrecv = nullptr;
    bool armed = false;
    Arena* xmem = nullptr;                                             // the wrapper's arena: library-owned exchange buffers must outlive the body, whose failure they announce
};
static const uint64_t POISON[4] = {~0ull, ~0ull, ~0ull, ~0ull};

static int create_proof_body(zk_ctx* ctx, const zk_plonk_pk_desc* pk, const void* const* advice, int advice_on_device, const void* const* instances,
                             const uint32_t* instance_lens, zk_rng_fn rng, void* rng_user, void* proof_out, size_t proof_cap, size_t* proof_len, ShardSignal& sig);

// proofs this process has in flight (every device together: a prover process drives one GPU): the side lane fills a lone proof's idle issue slots — with three and more
// in flight the other proofs do that already, and a helper context per proof only adds kernels to the crowd (measured: DESIGN 3.7)
static std::atomic<int> g_proofs_in_flight{0};
struct InFlight { InFlight() { g_proofs_in_flight.fetch_add(1); } ~InFlight() { g_proofs_in_flight.fetch_sub(1); } };

extern "C" int zk_plonk_create_proof(zk_ctx* ctx, const zk_plonk_pk_desc* pk, const void* const* advice, int advice_on_device, const void* const* instances,
                                     const uint32_t* instance_lens, zk_rng_fn rng, void* rng_user, void* proof_out, size_t proof_cap, size_t* proof_len) ZK_ABI_TRY {
    if (!ctx || !pk) return ZK_ERR_ARG;
    InFlight counted;
    Arena xmem(ctx);                                                   // (before the body's own arena: a failing body has returned everything else when the poisoned block travels)
    ShardSignal sig;
    sig.xmem = &xmem;
    int rc;
    try { rc = create_proof_body(ctx, pk, advice, advice_on_device, instances, instance_lens, rng, rng_user, proof_out, proof_cap, proof_len, sig); }
    catch (...) { rc = abi_exception(ctx, "zk_plonk_create_proof"); }   // (here rather than at the barrier below: the other ranks of a sharded proof are told first)
    if (rc != ZK_OK && rc != ZK_ERR_COMM && sig.armed && sig.next < sig.sizes.size()) {
        std::string why = zk_last_error(ctx) ? zk_last_error(ctx) : "";
        if (zk_dev_upload(ctx, sig.xsend, POISON, 32) == ZK_OK && zk_dev_sync(ctx) == ZK_OK)
            (void)pk->allgather(pk->allgather_user, sig.xsend, sig.xrecv, sig.sizes[sig.next]);
        pk_fail(ctx, rc, "%s [rank %u of a sharded proof: failure signalled to the other ranks in exchange %zu]", why.c_str(), pk->shard_rank, sig.next);
    }
    return rc;
} ZK_ABI_CATCH(ctx)

static int create_proof_body(zk_ctx* ctx, const zk_plonk_pk_desc* pk, const void* const* advice, int advice_on_device, const void* const* instances,
                             const uint32_t* instance_lens, zk_rng_fn rng, void* rng_user, void* proof_out, size_t proof_cap, size_t* proof_len, ShardSignal& sig) {
    if (!ctx || !pk) return ZK_ERR_ARG;
    if (pk->struct_size != sizeof(zk_plonk_pk_desc))
        return pk_fail(ctx, ZK_ERR_ARG, "zk_plonk_create_proof: zk_plonk_pk_desc.struct_size %u, expected %zu (ABI version %u)", pk->struct_size, sizeof(zk_plonk_pk_desc), ZK_ABI_VERSION);
    if (!rng || !proof_len || (pk->n_advice && !advice)) return ZK_ERR_ARG;
    const uint32_t k = pk->k, ek = pk->extended_k, bf = pk->blinding_factors, L = pk->n_lookups;
    const size_t n = (size_t)1 << k, en = (size_t)1 << ek, col_bytes = n * 32;
    if (k < 1 || ek < k || ek > 27 || bf + 2 >= n || pk->cs_degree < 3) return ZK_ERR_ARG;
    const size_t usable = n - (bf + 1);
    const uint32_t chunk = pk->cs_degree - 2;
    const uint32_t n_sets = pk->n_perm_columns ? (pk->n_perm_columns + chunk - 1) / chunk : 0;
    const uint32_t n_pieces = pk->cs_degree - 1;
    // the descriptor is the caller's: refuse indices that would read outside its arrays
    const uint32_t world = pk->shard_world > 1 ? pk->shard_world : 1, rank = world > 1 ? pk->shard_rank : 0;
    const bool sharded = world > 1;
    if ((pk->n_fixed && (!pk->fixed_values || !pk->fixed_polys)) || (pk->n_perm_columns && (!pk->perm_columns || !pk->sigma_values || !pk->sigma_polys)) ||
        (L && (!pk->lookup_input_programs || !pk->lookup_table_programs || !pk->lookup_table_key)) || (pk->n_advice_queries && !pk->advice_queries) ||
        (pk->n_fixed_queries && !pk->fixed_queries) || !pk->transcript_repr)
        return ZK_ERR_ARG;
    // a single-GPU key that holds cosets 0 .. n_pieces-1 instead of the extended domain (zk_plonk_pk_build does when cs_degree - 1 is not a power of two): the quotient is
    // evaluated on those cosets only and the pieces of h(X) come from zk_cosets_to_pieces_dev
    const bool by_cosets = !sharded && pk->coset_l && pk->coset_l[0] && n_pieces < (1u << (ek - k)) && n_pieces <= 8;
    if (!sharded && !by_cosets && ((pk->n_fixed && !pk->fixed_cosets) || (pk->n_perm_columns && !pk->sigma_cosets) || !pk->l0 || !pk->l_last || !pk->l_active_row)) return ZK_ERR_ARG;
    if (by_cosets && ((pk->n_fixed && !pk->coset_fixed) || (pk->n_perm_columns && !pk->coset_sigma))) return ZK_ERR_ARG;
    if (sharded && (rank >= world || n % world || !pk->allgather || (pk->n_fixed && !pk->coset_fixed) || (pk->n_perm_columns && !pk->coset_sigma) || !pk->coset_l)) return ZK_ERR_ARG;
    // the quotient's units of this rank (see zk_plonk_pk_desc): (coset, first row, rows)
    struct Unit { uint32_t coset; uint64_t lo, rows; };
    std::vector<Unit> units;
    std::vector<uint32_t> my_cosets;
    const uint32_t n_cosets = 1u << (ek - k);
    uint32_t parts = 1;
    size_t slots = 0, unit_rows = n;
    if (sharded) {
        if (world > n_cosets && world % n_cosets == 0) { const uint32_t p = world / n_cosets; if ((p & (p - 1)) == 0 && n % p == 0) parts = p; }
        const size_t n_units = (size_t)n_cosets * parts;
        unit_rows = n / parts;
        slots = (n_units + world - 1) / world;
        for (size_t u = (size_t)rank * slots; u < (size_t)(rank + 1) * slots && u < n_units; u++) {
            units.push_back({(uint32_t)(u / parts), (u % parts) * unit_rows, unit_rows});
            if (my_cosets.empty() || my_cosets.back() != units.back().coset) my_cosets.push_back(units.back().coset);
        }
    }
    const size_t n_loc = n / world, shard_lo = (size_t)rank * n_loc;
    for (uint32_t j = 0; j < pk->n_perm_columns; j++) {
        const uint32_t ty = pk->perm_columns[2 * j], ix = pk->perm_columns[2 * j + 1];
        if (ty > 2 || ix >= (ty == 0 ? pk->n_advice : ty == 1 ? pk->n_fixed : pk->n_instance)) return ZK_ERR_ARG;
    }
    for (uint32_t i = 0; i < pk->n_advice_queries; i++) if (pk->advice_queries[2 * i] >= pk->n_advice) return ZK_ERR_ARG;
    for (uint32_t i = 0; i < pk->n_fixed_queries; i++) if (pk->fixed_queries[2 * i] >= pk->n_fixed) return ZK_ERR_ARG;
    for (uint32_t i = 0; i < pk->n_advice; i++) if (!advice[i]) return ZK_ERR_ARG;
    Arena mem(ctx);
    SideLane lane;                                                     // (after the arena: its destructor joins the helper thread before the buffers go back to the pool)
    if (pk->transcript > 2) return ZK_ERR_ARG;
    // exchange buffers of a sharded proof: the caller's (e.g. two torch tensors, so that its callback can hand RCCL tensors) or the proof's own.  First thing of all:
    // from here on this rank can tell the others about a failure of its own (ShardSignal)
    void* xsend = nullptr; void* xrecv = nullptr;
    if (sharded) {
        const size_t most_cols = std::max<size_t>({pk->n_advice, 2 * (size_t)L, (size_t)n_sets + L, n_pieces, 1});
        const size_t need = std::max(slots * unit_rows * 32, most_cols * 128);
        if (pk->xchg_send && pk->xchg_recv) { if (pk->xchg_cap < need) return ZK_ERR_LIMIT; xsend = pk->xchg_send; xrecv = pk->xchg_recv; }
        else { xsend = sig.xmem->get(need); xrecv = sig.xmem->get(need * world); if (!xsend || !xrecv) return ZK_ERR_HIP; }
        for (size_t cols : {(size_t)pk->n_advice, 2 * (size_t)L, (size_t)n_sets + L, (size_t)1}) if (cols) sig.sizes.push_back(cols * 128);   // advice, permuted pairs, grand products, random poly
        sig.sizes.push_back(slots * unit_rows * 32);                                                                                       // the quotient's numerators
        for (size_t cols : {(size_t)n_pieces, (size_t)1, (size_t)1}) sig.sizes.push_back(cols * 128);                                        // h pieces, SHPLONK h(X) and quotient
        sig.xsend = xsend; sig.xrecv = xrecv; sig.armed = true;
    }
    // one exchange: the caller's collective (the library's stream is idle when it runs), then the other ranks' failure marks — the first 32 bytes of every rank's block,
    // read from `gathered_host` when the caller has downloaded the blocks anyway
    auto exchange = [&](size_t bytes, uint64_t* gathered_host) -> int {
        if (sig.next >= sig.sizes.size() || sig.sizes[sig.next] != bytes) return pk_fail(ctx, ZK_ERR_ARG, "zk_plonk_create_proof: exchange %zu of %zu bytes is not in the proof's schedule", sig.next, bytes);
        PK(zk_dev_sync(ctx));
        const size_t ex = sig.next;
        sig.next = sig.sizes.size();                                   // (no signalling after a failure in here: the collective itself is in doubt)
        if (pk->allgather(pk->allgather_user, xsend, xrecv, bytes)) return pk_fail(ctx, ZK_ERR_COMM, "zk_plonk_create_proof: the caller's all-gather failed in exchange %zu", ex);
        if (gathered_host) PK(zk_dev_download(ctx, gathered_host, xrecv, bytes * world));
        for (uint32_t r = 0; r < world; r++) {
            uint64_t head[4];
            if (gathered_host) memcpy(head, gathered_host + (size_t)r * bytes / 8, 32);
            else PK(zk_dev_download(ctx, head, (const char*)xrecv + (size_t)r * bytes, 32));
            if (!memcmp(head, POISON, 32)) return pk_fail(ctx, ZK_ERR_COMM, "zk_plonk_create_proof: rank %u reported a failure of its own in exchange %zu", r, ex);
        }
        sig.next = ex + 1;
        return ZK_OK;
    };
    Transcript tr((int)pk->transcript);
    Draws draws;
    for (double& v : g_phase_ms) v = 0;
    PhaseClock clk;
    // The caller's `&mut rng` is consumed in halo2's order (plonk/prover.rs and the argument provers it calls; [3P-MEM]).  draw_schedule 1 (the only one:
    // every binding sets it): the blinding rows of every advice column, then one Blind(Fr::random) per advice column (KZG ignores the value, the stream advances); per lookup, in order:
    // permute_expression_pair's input rows then table rows, then commit_values' two Blinds; per permutation set its rows + one Blind; per lookup product its rows + one Blind; the
    // vanishing argument's n coefficients + one Blind; one Blind per h(X) piece.
    if (pk->draw_schedule != 1) return pk_fail(ctx, ZK_ERR_ARG, "zk_plonk_create_proof: draw_schedule %u (1 = halo2_proofs v2023_01_20, the only schedule this build knows)", pk->draw_schedule);
    auto plan = [&](size_t count) { draws.counts.push_back(count); return draws.counts.size() - 1; };
    std::vector<size_t> d_bi(L), d_bt(L), d_pb(n_sets), d_lb(L);
    for (uint32_t i = 0; i < pk->n_advice; i++) plan(n - usable);                                  // items [0, n_advice)
    for (uint32_t i = 0; i < pk->n_advice; i++) plan(1);
    for (uint32_t l = 0; l < L; l++) { d_bi[l] = plan(bf + 1); d_bt[l] = plan(bf + 1); plan(1); plan(1); }
    for (uint32_t s = 0; s < n_sets; s++) { d_pb[s] = plan(bf); plan(1); }
    for (uint32_t l = 0; l < L; l++) { d_lb[l] = plan(bf); plan(1); }
    const size_t d_rp = plan(n);
    for (uint32_t i = 0; i < 1 + n_pieces; i++) plan(1);
    draws.start(rng, rng_user, mem.pool);

    // ---- 1. vk, instances ----------------------------------------------------------------------------------------------------------------------------
    tr.common_scalar(Fr::to_mont(load32(pk->transcript_repr)));
    std::vector<void*> inst_values;
    for (uint32_t c = 0; c < pk->n_instance; c++) {
        const uint32_t len = instance_lens ? instance_lens[c] : 0;
        if (len > usable || (len && (!instances || !instances[c]))) return ZK_ERR_ARG;
        std::vector<uint64_t> col((size_t)len * 4);
        for (uint32_t i = 0; i < len; i++) {
            const u256 canon = load32((const char*)instances[c] + 32 * i);
            if (!Fr::eq(Fr::reduce_once(canon), canon)) return ZK_ERR_ARG;          // not a canonical scalar (Fr::from_repr would refuse it)
            const Fe v = Fr::to_mont(canon);
            tr.common_scalar(v);
            memcpy(&col[4 * (size_t)i], v.v, 32);
        }
        void* d = mem.get(col_bytes);
        if (!d) return ZK_ERR_HIP;
        PK(zk_dev_zero(ctx, d, col_bytes));
        if (len) PK(zk_dev_upload(ctx, d, col.data(), (size_t)len * 32));
        inst_values.push_back(d);
    }
    clk.lap(0);
    // ---- 2. advice: upload (host columns), blind, commit ------------------------------------------------------------------------------------------------
    std::vector<void*> adv(pk->n_advice);
    {
        std::vector<void*> dst;
        std::vector<const void*> src;
        for (uint32_t i = 0; i < pk->n_advice; i++) {
            if (advice_on_device) adv[i] = (void*)advice[i];
            else { adv[i] = mem.get(col_bytes); if (!adv[i]) return ZK_ERR_HIP; dst.push_back(adv[i]); src.push_back(advice[i]); }
        }
        if (!dst.empty()) PK(zk_dev_upload_batch(ctx, dst.data(), src.data(), dst.size(), col_bytes));
        std::vector<void*> bdst(pk->n_advice);
        std::vector<const void*> bsrc(pk->n_advice);
        for (uint32_t i = 0; i < pk->n_advice; i++) { bdst[i] = (char*)adv[i] + usable * 32; bsrc[i] = draws.take(i); }
        if (pk->n_advice) PK(zk_dev_upload_batch(ctx, bdst.data(), bsrc.data(), pk->n_advice, (n - usable) * 32));
    }
    // the side lane (SideLane above): a single-GPU proof on the extended domain hands every phase's columns to the helper context as soon as their values are final
    bool side = false;
    {
        int want = 0;
        if (!sharded && !by_cosets && zk_tune_get(ctx, "prover_side_lane", &want) == ZK_OK && (want >= 2 || (want == 1 && g_proofs_in_flight.load() <= 2))) {
            zk_ctx* h = zk_internal_helper_ctx(ctx);
            if (h) { try { lane.start(h); side = true; } catch (const std::system_error&) { side = false; } }      // no thread to be had: the proof runs in one lane, as with three proofs in flight
        }
    }
    std::vector<void*> coefA, extA, coefB, extB, coefC, extC;         // coefficient and extended forms of: advice + instance | permuted pairs | grand products
    auto early = [&](const std::vector<void*>& vals, std::vector<void*>& coefs, std::vector<void*>& exts) -> int {
        coefs.resize(vals.size()); exts.resize(vals.size());
        for (size_t i = 0; i < vals.size(); i++) { coefs[i] = mem.get(col_bytes); exts[i] = mem.get(en * 32); if (!coefs[i] || !exts[i]) return ZK_ERR_HIP; }
        if (vals.empty()) return ZK_OK;
        PK(zk_dev_sync(ctx));                                         // the values are final: everything that wrote them ran on this context's stream
        zk_ctx* h = lane.h;
        lane.submit([h, vals, coefs, exts, col_bytes, k, ek]() -> int {
            for (size_t i = 0; i < vals.size(); i++) { const int r = zk_dev_copy(h, coefs[i], vals[i], col_bytes); if (r) return r; }
            int r = zk_lagrange_to_coeff_batch_dev(h, coefs.data(), coefs.size(), k);
            if (!r) r = zk_coeff_to_extended_batch_dev(h, (const void* const*)coefs.data(), exts.data(), coefs.size(), k, ek);
            return r ? r : zk_dev_sync(h);
        });
        return ZK_OK;
    };
    if (side) {
        std::vector<void*> vals(adv);
        vals.insert(vals.end(), inst_values.begin(), inst_values.end());
        PK(early(vals, coefA, extA));
    }
    auto commit = [&](uint64_t table, const std::vector<void*>& cols) -> int {
        if (cols.empty()) return ZK_OK;
        std::vector<uint64_t> out(cols.size() * 12);
        if (!sharded) PK(zk_msm_batch_dev(ctx, table, (const void* const*)cols.data(), cols.size(), n, out.data()));
        else {
            // this rank's index range of every column against its slice of the table; the 128-byte partial points of the phase travel in ONE all-gather
            std::vector<const void*> slice(cols.size());
            for (size_t i = 0; i < cols.size(); i++) slice[i] = (const char*)cols[i] + shard_lo * 32;
            const size_t bytes = cols.size() * 128;
            std::vector<uint64_t> part(cols.size() * 16), all((size_t)world * cols.size() * 16);
            PK(zk_msm_batch_partial_dev(ctx, table, slice.data(), cols.size(), n_loc, part.data()));
            PK(zk_dev_upload(ctx, xsend, part.data(), bytes));
            PK(exchange(bytes, all.data()));
            PK(zk_g1_sum_xyzz_batch(all.data(), world, cols.size(), out.data()));
        }
        for (size_t i = 0; i < cols.size(); i++) tr.write_point(&out[12 * i]);
        return ZK_OK;
    };
    PK(commit(pk->srs_g_lagrange, adv));
    clk.lap(1);
    // ---- 3. theta; lookups: compress, permute, commit ---------------------------------------------------------------------------------------------------
    const Fe theta = tr.squeeze();
    const Fe one = Fr::one();
    std::vector<void*> cin(L), ctab(L);
    {
        const void* anycol = pk->n_fixed ? pk->fixed_values[0] : (pk->n_advice ? adv[0] : nullptr);
        std::map<uint32_t, void*> table_cache;
        auto compress = [&](uint64_t prog, void** out) -> int {
            *out = mem.get(col_bytes);
            if (!*out) return ZK_ERR_HIP;
            zk_quotient_args a;
            ZK_STRUCT_INIT(a);
            a.fixed = pk->fixed_values; a.advice = (const void* const*)adv.data(); a.instance = (const void* const*)inst_values.data();
            a.l0 = a.l_last = a.l_active_row = anycol;
            a.beta = a.gamma = a.y = one.v; a.theta = theta.v; a.challenges = one.v;
            a.out = *out;
            return zk_quotient_run_dev(ctx, prog, &a);
        };
        for (uint32_t l = 0; l < L; l++) {
            auto it = table_cache.find(pk->lookup_table_key[l]);
            if (it == table_cache.end()) { void* t = nullptr; PK(compress(pk->lookup_table_programs[l], &t)); it = table_cache.emplace(pk->lookup_table_key[l], t).first; }
            PK(compress(pk->lookup_input_programs[l], &cin[l]));
            ctab[l] = it->second;
        }
    }
    std::vector<void*> pin(L), ptab(L);
    if (L) {
        std::vector<uint64_t> bi((size_t)L * (bf + 1) * 4), bt((size_t)L * (bf + 1) * 4);
        for (uint32_t l = 0; l < L; l++) {
            memcpy(&bi[(size_t)l * (bf + 1) * 4], draws.take(d_bi[l]), (bf + 1) * 32);
            memcpy(&bt[(size_t)l * (bf + 1) * 4], draws.take(d_bt[l]), (bf + 1) * 32);
        }
        for (uint32_t l = 0; l < L; l++) { pin[l] = mem.get(col_bytes); ptab[l] = mem.get(col_bytes); if (!pin[l] || !ptab[l]) return ZK_ERR_HIP; }
        PK(zk_lookup_permute_batch_dev(ctx, (const void* const*)cin.data(), (const void* const*)ctab.data(), L, k, bf, bi.data(), bt.data(), pin.data(), ptab.data()));
        std::vector<void*> flat;
        for (uint32_t l = 0; l < L; l++) { flat.push_back(pin[l]); flat.push_back(ptab[l]); }
        if (side) PK(early(flat, coefB, extB));
        PK(commit(pk->srs_g_lagrange, flat));
    }
    clk.lap(2);
    // ---- 4. beta, gamma; grand products -------------------------------------------------------------------------------------------------------------------
    const Fe beta = tr.squeeze(), gamma = tr.squeeze();
    std::vector<void*> zs(n_sets), lzs(L);
    if (n_sets) {
        std::vector<const void*> vals(pk->n_perm_columns);
        for (uint32_t j = 0; j < pk->n_perm_columns; j++) {
            const uint32_t ty = pk->perm_columns[2 * j], ix = pk->perm_columns[2 * j + 1];
            vals[j] = ty == 0 ? adv[ix] : ty == 1 ? pk->fixed_values[ix] : inst_values[ix];
        }
        std::vector<uint64_t> blind((size_t)n_sets * bf * 4);
        for (uint32_t s = 0; s < n_sets; s++) memcpy(&blind[(size_t)s * bf * 4], draws.take(d_pb[s]), bf * 32);
        for (uint32_t s = 0; s < n_sets; s++) { zs[s] = mem.get(col_bytes); if (!zs[s]) return ZK_ERR_HIP; }
        PK(zk_permutation_product_all_dev(ctx, vals.data(), pk->sigma_values, pk->n_perm_columns, chunk, k, beta.v, gamma.v, blind.data(), bf, zs.data()));
    }
    if (L) {
        std::vector<const void*> quads;
        for (uint32_t l = 0; l < L; l++) { quads.push_back(cin[l]); quads.push_back(ctab[l]); quads.push_back(pin[l]); quads.push_back(ptab[l]); }
        std::vector<uint64_t> blind((size_t)L * bf * 4);
        for (uint32_t l = 0; l < L; l++) memcpy(&blind[(size_t)l * bf * 4], draws.take(d_lb[l]), bf * 32);
        for (uint32_t l = 0; l < L; l++) { lzs[l] = mem.get(col_bytes); if (!lzs[l]) return ZK_ERR_HIP; }
        PK(zk_lookup_product_batch_dev(ctx, quads.data(), L, k, beta.v, gamma.v, blind.data(), bf, lzs.data()));
    }
    {
        std::vector<void*> both(zs);
        both.insert(both.end(), lzs.begin(), lzs.end());
        if (side) PK(early(both, coefC, extC));
        PK(commit(pk->srs_g_lagrange, both));
    }
    clk.lap(3);
    // ---- 5. vanishing argument: random polynomial ------------------------------------------------------------------------------------------------------------
    void* random_poly = mem.get(col_bytes);
    if (!random_poly) return ZK_ERR_HIP;
    PK(zk_dev_upload(ctx, random_poly, draws.take(d_rp), col_bytes));
    PK(commit(pk->srs_g, {random_poly}));
    clk.lap(4);
    // ---- 6. y; coefficient form; extended cosets; h(X) numerator ----------------------------------------------------------------------------------------------
    const Fe y = tr.squeeze();
    std::vector<void*> side_ext;
    if (side) {                                                       // the helper context has brought every column to both forms: from here on the vectors name the coefficient forms
        const int rc_side = lane.wait();
        if (rc_side) return pk_fail(ctx, rc_side, "zk_plonk_create_proof: transforms on the helper context: %s", zk_last_error(lane.h));
        // the Lagrange forms have been committed (this context, synchronous calls) and copied (the lane, waited for above): nothing reads them again.  Back to the pool now
        // rather than at the end of the proof — the lane's copies would otherwise double the columns a proof holds through its quotient phase.  Not the caller's own
        // advice_on_device columns (give_back only knows what the arena handed out).
        for (void* p : adv) mem.give_back(p);
        for (void* p : inst_values) mem.give_back(p);
        for (void* p : zs) mem.give_back(p);
        for (void* p : lzs) mem.give_back(p);
        for (uint32_t l = 0; l < L; l++) { mem.give_back(pin[l]); mem.give_back(ptab[l]); mem.give_back(cin[l]); mem.give_back(ctab[l]); }
        for (uint32_t i = 0; i < pk->n_advice; i++) { adv[i] = coefA[i]; side_ext.push_back(extA[i]); }
        for (uint32_t c = 0; c < pk->n_instance; c++) { inst_values[c] = coefA[pk->n_advice + c]; side_ext.push_back(extA[pk->n_advice + c]); }
        for (uint32_t s_ = 0; s_ < n_sets; s_++) { zs[s_] = coefC[s_]; side_ext.push_back(extC[s_]); }
        for (uint32_t l = 0; l < L; l++) { lzs[l] = coefC[n_sets + l]; side_ext.push_back(extC[n_sets + l]); }
        for (uint32_t l = 0; l < L; l++) { pin[l] = coefB[2 * l]; ptab[l] = coefB[2 * l + 1]; side_ext.push_back(extB[2 * l]); side_ext.push_back(extB[2 * l + 1]); }
    }
    std::vector<void*> lag(adv);
    lag.insert(lag.end(), inst_values.begin(), inst_values.end());
    lag.insert(lag.end(), zs.begin(), zs.end());
    lag.insert(lag.end(), lzs.begin(), lzs.end());
    for (uint32_t l = 0; l < L; l++) { lag.push_back(pin[l]); lag.push_back(ptab[l]); }
    if (!side) PK(zk_lagrange_to_coeff_batch_dev(ctx, lag.data(), lag.size(), k));
    void* h_ext = by_cosets ? nullptr : mem.get(en * 32);
    if (!by_cosets && !h_ext) return ZK_ERR_HIP;
    std::vector<void*> numer(by_cosets ? n_pieces : 0);
    auto quotient_args = [&](zk_quotient_args& a, void* const* ext, std::vector<const void*>& e_in, std::vector<const void*>& e_tab) {
        const size_t nA = pk->n_advice, nI = pk->n_instance;
        e_in.clear(); e_tab.clear();
        for (uint32_t l = 0; l < L; l++) { e_in.push_back(ext[nA + nI + n_sets + L + 2 * l]); e_tab.push_back(ext[nA + nI + n_sets + L + 2 * l + 1]); }
        ZK_STRUCT_INIT(a);
        a.advice = (const void* const*)ext; a.instance = (const void* const*)ext + nA;
        a.perm_products = (const void* const*)ext + nA + nI; a.n_sets = n_sets;
        a.lookup_product = (const void* const*)ext + nA + nI + n_sets; a.lookup_input = e_in.data(); a.lookup_table = e_tab.data();
        a.challenges = one.v; a.beta = beta.v; a.gamma = gamma.v; a.theta = theta.v; a.y = y.v;
    };
    // Degree split (zkmi355.h, zk_quotient_program_split): the identities of degree <= 3 — about half of the sgx-shaped program's arithmetic — are evaluated on
    // low_cosets = 2 cosets only; their share of h(X) has degree below 2 n and is added to the first two pieces.  Single-GPU proofs only (a sharded proof keeps the whole program).
    uint32_t low_cosets = 0;
    {
        int want = 0;
        if (!sharded && zk_tune_get(ctx, "quot_degree_split", &want) == ZK_OK && want) PK(zk_quotient_program_split(ctx, pk->program, &low_cosets, nullptr, nullptr));
        if (low_cosets >= n_pieces || low_cosets > n_cosets) low_cosets = 0;
    }
    std::vector<void*> numer_low(low_cosets);
    if (low_cosets) {
        void* blk = mem.get(low_cosets * col_bytes);                  // (one block: zk_quotient_run_low_dev writes the cosets back to back)
        if (!blk) return ZK_ERR_HIP;
        for (uint32_t j = 0; j < low_cosets; j++) numer_low[j] = (char*)blk + (size_t)j * col_bytes;
    }
    if (by_cosets) {
        std::vector<void*> cols(lag.size());
        for (auto& e : cols) { e = mem.get(col_bytes); if (!e) return ZK_ERR_HIP; }
        for (auto& e : numer) { e = mem.get(col_bytes); if (!e) return ZK_ERR_HIP; }
        for (uint32_t j = 0; j < n_pieces; j++) {
            PK(zk_coeff_to_coset_batch_dev(ctx, (const void* const*)lag.data(), cols.data(), cols.size(), k, ek, j));
            std::vector<const void*> e_in, e_tab;
            zk_quotient_args a;
            quotient_args(a, cols.data(), e_in, e_tab);
            a.fixed = pk->coset_fixed + (size_t)j * pk->n_fixed; a.perm_cosets = pk->coset_sigma + (size_t)j * pk->n_perm_columns;
            a.l0 = pk->coset_l[3 * j]; a.l_last = pk->coset_l[3 * j + 1]; a.l_active_row = pk->coset_l[3 * j + 2];
            a.out = numer[j];
            if (!low_cosets) PK(zk_quotient_run_coset_dev(ctx, pk->program, &a, j));
            else {
                PK(zk_quotient_run_coset_part_dev(ctx, pk->program, &a, j, 1));
                if (j < low_cosets) { a.out = numer_low[j]; PK(zk_quotient_run_coset_part_dev(ctx, pk->program, &a, j, 2)); }
            }
        }
        for (auto e : cols) mem.give_back(e);
    } else if (!sharded) {
        std::vector<void*> ext(lag.size());
        if (side) ext = side_ext;
        else {
            for (auto& e : ext) { e = mem.get(en * 32); if (!e) return ZK_ERR_HIP; }
            PK(zk_coeff_to_extended_batch_dev(ctx, (const void* const*)lag.data(), ext.data(), ext.size(), k, ek));
        }
        std::vector<const void*> e_in, e_tab;
        zk_quotient_args a;
        quotient_args(a, ext.data(), e_in, e_tab);
        a.fixed = pk->fixed_cosets; a.l0 = pk->l0; a.l_last = pk->l_last; a.l_active_row = pk->l_active_row; a.perm_cosets = pk->sigma_cosets; a.out = h_ext;
        if (!low_cosets) PK(zk_quotient_run_dev(ctx, pk->program, &a));
        else {
            PK(zk_quotient_run_high_dev(ctx, pk->program, &a));
            a.out = numer_low[0];
            PK(zk_quotient_run_low_dev(ctx, pk->program, &a, low_cosets));
        }
        for (auto e : ext) mem.give_back(e);
    } else {
        // this rank brings the columns to ITS cosets only (size-n NTTs), evaluates the numerator on its units straight into the send buffer; one all-gather
        // carries every rank's numerators (unit order = coset order, rows ascending: rank r's block starts at unit r * slots), then the cosets are interleaved
        std::vector<void*> cols(lag.size());
        for (auto& e : cols) { e = mem.get(col_bytes); if (!e) return ZK_ERR_HIP; }
        int at = -1;
        for (size_t s_ = 0; s_ < units.size(); s_++) {
            const Unit& u = units[s_];
            size_t ci = 0;
            while (my_cosets[ci] != u.coset) ci++;
            if ((int)u.coset != at) { PK(zk_coeff_to_coset_batch_dev(ctx, (const void* const*)lag.data(), cols.data(), cols.size(), k, ek, u.coset)); at = (int)u.coset; }
            std::vector<const void*> e_in, e_tab;
            zk_quotient_args a;
            quotient_args(a, cols.data(), e_in, e_tab);
            a.fixed = pk->coset_fixed + ci * pk->n_fixed; a.perm_cosets = pk->coset_sigma + ci * pk->n_perm_columns;
            a.l0 = pk->coset_l[3 * ci]; a.l_last = pk->coset_l[3 * ci + 1]; a.l_active_row = pk->coset_l[3 * ci + 2];
            a.out = (char*)xsend + s_ * unit_rows * 32;
            if (parts == 1) PK(zk_quotient_run_coset_dev(ctx, pk->program, &a, u.coset));
            else PK(zk_quotient_run_coset_rows_dev(ctx, pk->program, &a, u.coset, u.lo, u.rows));
        }
        if (units.empty()) PK(zk_dev_zero(ctx, xsend, 32));          // (more ranks than units: nothing of this rank's travels, but its block's head is read as a status)
        PK(exchange(slots * unit_rows * 32, nullptr));
        std::vector<const void*> srcs(n_cosets);
        for (uint32_t j = 0; j < n_cosets; j++) srcs[j] = (const char*)xrecv + (size_t)j * col_bytes;
        PK(zk_fr_interleave_dev(ctx, srcs.data(), n_cosets, n, h_ext));
        for (auto e : cols) mem.give_back(e);
    }
    clk.lap(5);
    // ---- 7. divide, back to coefficients, commit the pieces ---------------------------------------------------------------------------------------------------------
    std::vector<void*> pieces(n_pieces);
    if (by_cosets) {
        for (auto& e : pieces) { e = mem.get(col_bytes); if (!e) return ZK_ERR_HIP; }
        PK(zk_cosets_to_pieces_dev(ctx, numer.data(), n_pieces, k, ek, pieces.data()));
        for (auto e : numer) mem.give_back(e);
    } else {
        PK(zk_divide_by_vanishing_poly_dev(ctx, h_ext, k, ek));
        PK(zk_extended_to_coeff_dev(ctx, h_ext, k, ek));
        for (uint32_t i = 0; i < n_pieces; i++) pieces[i] = (char*)h_ext + (size_t)i * col_bytes;
    }
    if (low_cosets) {                                                 // h = (the high part's pieces) + (the low part's two pieces)
        std::vector<void*> lowp(low_cosets);
        for (auto& e : lowp) { e = mem.get(col_bytes); if (!e) return ZK_ERR_HIP; }
        PK(zk_cosets_to_pieces_dev(ctx, numer_low.data(), low_cosets, k, ek, lowp.data()));
        const Fe ones[2] = {one, one};
        for (uint32_t i = 0; i < low_cosets; i++) {
            const void* two[2] = {pieces[i], lowp[i]};
            PK(zk_fr_lincomb_dev(ctx, two, ones, 2, n, pieces[i]));
        }
        for (auto e : lowp) mem.give_back(e);
        mem.give_back(numer_low[0]);
    }
    PK(commit(pk->srs_g, pieces));
    clk.lap(6);
    // ---- 8. x; evaluations ------------------------------------------------------------------------------------------------------------------------------------------------
    const Fe x = tr.squeeze();
    Fe xn = x;
    for (uint32_t i = 0; i < k; i++) xn = Fr::sqr(xn);
    Fe omega;
    {
        const uint64_t rl[4] = BN254_FR_ROOT_OF_UNITY_M;
        for (int i = 0; i < 8; i++) omega.v[i] = (uint32_t)(rl[i >> 1] >> (32 * (i & 1)));
        for (uint32_t i = k; i < 28; i++) omega = Fr::sqr(omega);
    }
    const Fe omega_inv = Fr::inv(omega);
    auto rot = [&](int32_t r) { return Fr::mul(x, r >= 0 ? fe_pow_u64(omega, (uint64_t)r) : fe_pow_u64(omega_inv, (uint64_t)(-(int64_t)r))); };
    void* h_poly = mem.get(col_bytes);
    if (!h_poly) return ZK_ERR_HIP;
    {
        std::vector<uint64_t> sc((size_t)n_pieces * 4);
        Fe p = Fr::one();
        for (uint32_t i = 0; i < n_pieces; i++) { memcpy(&sc[4 * i], p.v, 32); p = Fr::mul(p, xn); }
        PK(zk_fr_lincomb_dev(ctx, (const void* const*)pieces.data(), sc.data(), n_pieces, n, h_poly));
    }
    const Fe x_last = rot(-(int32_t)(bf + 1)), x_next = rot(1), x_prev = rot(-1);
    std::vector<Query> q;
    for (uint32_t i = 0; i < pk->n_advice_queries; i++) q.push_back({adv[pk->advice_queries[2 * i]], rot((int32_t)pk->advice_queries[2 * i + 1]), Fr::zero()});
    for (uint32_t i = 0; i < pk->n_fixed_queries; i++) q.push_back({pk->fixed_polys[pk->fixed_queries[2 * i]], rot((int32_t)pk->fixed_queries[2 * i + 1]), Fr::zero()});
    q.push_back({random_poly, x, Fr::zero()});
    for (uint32_t j = 0; j < pk->n_perm_columns; j++) q.push_back({pk->sigma_polys[j], x, Fr::zero()});
    for (uint32_t s = 0; s < n_sets; s++) {
        q.push_back({zs[s], x, Fr::zero()});
        q.push_back({zs[s], x_next, Fr::zero()});
        if (s + 1 < n_sets) q.push_back({zs[s], x_last, Fr::zero()});
    }
    for (uint32_t l = 0; l < L; l++) {
        q.push_back({lzs[l], x, Fr::zero()}); q.push_back({lzs[l], x_next, Fr::zero()});
        q.push_back({pin[l], x, Fr::zero()}); q.push_back({pin[l], x_prev, Fr::zero()});
        q.push_back({ptab[l], x, Fr::zero()});
    }
    q.push_back({h_poly, x, Fr::zero()});
    {
        std::vector<const void*> polys(q.size());
        std::vector<uint64_t> pts(q.size() * 4), ev(q.size() * 4);
        for (size_t i = 0; i < q.size(); i++) { polys[i] = q[i].poly; memcpy(&pts[4 * i], q[i].point.v, 32); }
        PK(zk_eval_polynomial_batch_dev(ctx, polys.data(), q.size(), n, pts.data(), ev.data()));
        for (size_t i = 0; i < q.size(); i++) q[i].eval = load32(&ev[4 * i]);
        for (size_t i = 0; i + 1 < q.size(); i++) tr.write_scalar(q[i].eval);            // h's evaluation is the verifier's to derive
    }
    clk.lap(7);
    // ---- 9. ProverSHPLONK: queries in the multi-open order ---------------------------------------------------------------------------------------------------------------
    std::vector<Query> mq;
    {
        size_t it = 0;
        std::vector<Query> q_adv(q.begin(), q.begin() + pk->n_advice_queries); it += pk->n_advice_queries;
        std::vector<Query> q_fix(q.begin() + it, q.begin() + it + pk->n_fixed_queries); it += pk->n_fixed_queries;
        const Query q_rand = q[it++];
        std::vector<Query> q_sig(q.begin() + it, q.begin() + it + pk->n_perm_columns); it += pk->n_perm_columns;
        std::vector<Query> q_pa, q_pl;
        for (uint32_t s = 0; s < n_sets; s++) { q_pa.push_back(q[it++]); q_pa.push_back(q[it++]); if (s + 1 < n_sets) q_pl.push_back(q[it++]); }
        std::vector<Query> q_lk;
        for (uint32_t l = 0; l < L; l++) {
            const Query pz = q[it], pzn = q[it + 1], pa = q[it + 2], pai = q[it + 3], ps = q[it + 4];
            it += 5;
            q_lk.push_back(pz); q_lk.push_back(pa); q_lk.push_back(ps); q_lk.push_back(pai); q_lk.push_back(pzn);      // lookup::Evaluated::open order
        }
        const Query q_h = q[it++];
        mq = q_adv;
        mq.insert(mq.end(), q_pa.begin(), q_pa.end());
        mq.insert(mq.end(), q_pl.rbegin(), q_pl.rend());
        mq.insert(mq.end(), q_lk.begin(), q_lk.end());
        mq.insert(mq.end(), q_fix.begin(), q_fix.end());
        mq.insert(mq.end(), q_sig.begin(), q_sig.end());
        mq.push_back(q_h); mq.push_back(q_rand);
    }
    const Fe yy = tr.squeeze();
    // construct_intermediate_sets: commitments (by polynomial) in first-appearance order, their point sets ascending by canonical value, sets in first-appearance order
    struct Com { const void* poly; std::map<u256, Fe, CanonLess> pts; };          // canonical point -> eval
    std::vector<Com> coms;
    std::map<u256, Fe, CanonLess> super;                                          // canonical -> Montgomery point
    for (auto& qq : mq) {
        const u256 cp = Fr::from_mont(qq.point);
        super.emplace(cp, qq.point);
        size_t ci = 0;
        while (ci < coms.size() && coms[ci].poly != qq.poly) ci++;
        if (ci == coms.size()) coms.push_back(Com{qq.poly, {}});
        coms[ci].pts.emplace(cp, qq.eval);
    }
    struct RSet { std::vector<u256> keys; std::vector<size_t> members; };
    std::vector<RSet> sets;
    for (size_t ci = 0; ci < coms.size(); ci++) {
        std::vector<u256> keys;
        for (auto& kv : coms[ci].pts) keys.push_back(kv.first);
        size_t si = 0;
        for (; si < sets.size(); si++) {
            if (sets[si].keys.size() != keys.size()) continue;
            bool same = true;
            for (size_t i = 0; i < keys.size() && same; i++) same = Fr::eq(sets[si].keys[i], keys[i]);
            if (same) break;
        }
        if (si == sets.size()) sets.push_back(RSet{keys, {}});
        sets[si].members.push_back(ci);
    }
    const Fe v = tr.squeeze();
    size_t pad = 1;
    for (auto& s : sets) pad = std::max(pad, s.keys.size());
    void* rbuf = mem.get(col_bytes);
    void* tmp0 = mem.get(col_bytes);
    void* tmp1 = mem.get(col_bytes);
    if (!rbuf || !tmp0 || !tmp1) return ZK_ERR_HIP;
    PK(zk_dev_zero(ctx, rbuf, col_bytes));
    std::vector<void*> quotients;
    std::vector<std::vector<std::vector<Fe>>> low(sets.size());                    // per set, per member: r(X) coefficients
    for (size_t si = 0; si < sets.size(); si++) {
        const RSet& s = sets[si];
        std::vector<Fe> pts;
        for (auto& key : s.keys) pts.push_back(super[key]);
        std::vector<Fe> rsum(pts.size(), Fr::zero());
        std::vector<const void*> polys;
        std::vector<uint64_t> scal;
        Fe ypow = Fr::one();
        const std::vector<std::vector<Fe>> basis = lagrange_basis(pts);
        for (size_t ci : s.members) {
            std::vector<Fe> evals;
            for (auto& key : s.keys) evals.push_back(coms[ci].pts[key]);
            const std::vector<Fe> r = interpolate_with_basis(basis, evals);
            for (size_t i = 0; i < r.size(); i++) rsum[i] = Fr::sub(rsum[i], Fr::mul(ypow, r[i]));
            low[si].push_back(r);
            polys.push_back(coms[ci].poly);
            scal.insert(scal.end(), (const uint64_t*)ypow.v, (const uint64_t*)ypow.v + 4);
            ypow = Fr::mul(ypow, yy);
        }
        std::vector<uint64_t> rs(pad * 4, 0);
        for (size_t i = 0; i < rsum.size(); i++) memcpy(&rs[4 * i], rsum[i].v, 32);
        PK(zk_dev_upload(ctx, rbuf, rs.data(), pad * 32));
        polys.push_back(rbuf);
        scal.insert(scal.end(), (const uint64_t*)one.v, (const uint64_t*)one.v + 4);
        PK(zk_fr_lincomb_dev(ctx, polys.data(), scal.data(), polys.size(), n, tmp0));
        void* cur = tmp0; void* oth = tmp1;
        size_t ln = n;
        for (auto& p : pts) { PK(zk_kate_division_dev(ctx, cur, ln, p.v, oth)); std::swap(cur, oth); ln--; }
        void* qi = mem.get(col_bytes);
        if (!qi) return ZK_ERR_HIP;
        PK(zk_fr_scale_dev(ctx, cur, one.v, qi, ln));
        if (n > ln) PK(zk_dev_zero(ctx, (char*)qi + ln * 32, (n - ln) * 32));
        quotients.push_back(qi);
    }
    std::vector<Fe> vp(sets.size());
    { Fe p = Fr::one(); for (size_t i = 0; i < sets.size(); i++) { vp[i] = p; p = Fr::mul(p, v); } }
    void* h_x = mem.get(col_bytes);
    if (!h_x) return ZK_ERR_HIP;
    {
        std::vector<uint64_t> sc(sets.size() * 4);
        for (size_t i = 0; i < sets.size(); i++) memcpy(&sc[4 * i], vp[i].v, 32);
        PK(zk_fr_lincomb_dev(ctx, (const void* const*)quotients.data(), sc.data(), quotients.size(), n, h_x));
    }
    PK(commit(pk->srs_g, {h_x}));
    const Fe u = tr.squeeze();
    {
        std::vector<Fe> super_pts;
        for (auto& kv : super) super_pts.push_back(kv.second);
        std::vector<Fe> z_diffs(sets.size());
        std::vector<const void*> polys;
        std::vector<Fe> scal;
        Fe cst = Fr::zero();
        for (size_t si = 0; si < sets.size(); si++) {
            std::vector<Fe> diffs;
            for (auto& kv : super) {
                bool in_set = false;
                for (auto& key : sets[si].keys) in_set |= Fr::eq(key, kv.first);
                if (!in_set) diffs.push_back(kv.second);
            }
            z_diffs[si] = vanishing_at(diffs, u);
            Fe ypow = Fr::one();
            for (size_t m = 0; m < sets[si].members.size(); m++) {
                const Fe w = Fr::mul(Fr::mul(vp[si], z_diffs[si]), ypow);
                polys.push_back(coms[sets[si].members[m]].poly);
                scal.push_back(w);
                cst = Fr::sub(cst, Fr::mul(w, eval_small(low[si][m], u)));
                ypow = Fr::mul(ypow, yy);
            }
        }
        const Fe zt = vanishing_at(super_pts, u), z0_inv = Fr::inv(z_diffs[0]);
        polys.push_back(h_x);
        scal.push_back(Fr::neg(zt));
        std::vector<uint64_t> rs(pad * 4, 0);
        const Fe c0 = Fr::mul(cst, z0_inv);
        memcpy(rs.data(), c0.v, 32);
        PK(zk_dev_upload(ctx, rbuf, rs.data(), pad * 32));
        polys.push_back(rbuf);
        std::vector<uint64_t> sc(polys.size() * 4);
        for (size_t i = 0; i + 1 < polys.size(); i++) { const Fe w = Fr::mul(scal[i], z0_inv); memcpy(&sc[4 * i], w.v, 32); }
        memcpy(&sc[4 * (polys.size() - 1)], one.v, 32);
        PK(zk_fr_lincomb_dev(ctx, polys.data(), sc.data(), polys.size(), n, tmp0));
        PK(zk_kate_division_dev(ctx, tmp0, n, u.v, tmp1));
        PK(zk_dev_zero(ctx, (char*)tmp1 + (n - 1) * 32, 32));
        PK(commit(pk->srs_g, {tmp1}));
    }
    clk.lap(8);
    if (tr.bad_point) return ZK_ERR_ARG;                              // a commitment to the zero polynomial under a transcript that cannot encode the identity
    draws.finish();
    *proof_len = tr.out.size();
    if (!proof_out || proof_cap < tr.out.size()) return ZK_ERR_LIMIT;
    memcpy(proof_out, tr.out.data(), tr.out.size());
    return ZK_OK;
}


// ---- the proving key as a library object (zk_plonk_pk_build / share / release / prove) -------------------------------------------------------------------------
// The device half of keygen_pk, written — like create_proof above — as a CLIENT of the public entry points: upload the Lagrange columns, lagrange_to_coeff,
// coeff_to_extended, l0 / l_last / l_active_row, load the ZKQ1 programs.  One PkMem per key per process (columns + the host arrays the descriptor points into);
// every holding context has a PkHandle with its own program handles (zk_quotient_program_share) and SRS handles.
namespace {
struct PkMem {
    std::vector<void*> owned;                                         // device allocations, freed by whoever drops the last handle
    std::vector<const void*> fixed_values, fixed_polys, fixed_cosets, sigma_values, sigma_polys, sigma_cosets;
    void* l[3] = {nullptr, nullptr, nullptr};
    std::vector<const void*> coset_fixed, coset_sigma, coset_l;      // a sharded key: [this rank's cosets][columns], n values each
    std::vector<uint32_t> perm_columns, advice_queries, fixed_queries, table_key;
    uint8_t transcript_repr[32];
    int holders = 0;
};
struct PkHandle {
    PkMem* mem = nullptr;
    zk_plonk_pk_desc desc;
    uint64_t program = 0;
    std::vector<uint64_t> in_prog, tab_prog;
    int in_use = 0;              // zk_plonk_prove calls running on this handle (g_pk_mu)
    bool released = false;       // zk_plonk_pk_release / zk_ctx_destroy arrived meanwhile: the last of those calls drops the handle
};
std::mutex g_pk_mu;
std::map<std::pair<zk_ctx*, uint64_t>, PkHandle*> g_pk_handles;
uint64_t g_pk_next = 1;

void pk_fill_desc(PkHandle* h, const zk_plonk_pk_desc& shape, uint64_t srs_g, uint64_t srs_g_lagrange) {
    PkMem* m = h->mem;
    zk_plonk_pk_desc& d = h->desc;
    d = shape;
    d.perm_columns = m->perm_columns.data(); d.advice_queries = m->advice_queries.data(); d.fixed_queries = m->fixed_queries.data();
    d.srs_g = srs_g; d.srs_g_lagrange = srs_g_lagrange; d.program = h->program;
    d.lookup_input_programs = h->in_prog.data(); d.lookup_table_programs = h->tab_prog.data(); d.lookup_table_key = m->table_key.data();
    d.fixed_values = m->fixed_values.data(); d.fixed_polys = m->fixed_polys.data(); d.fixed_cosets = m->fixed_cosets.data();
    d.sigma_values = m->sigma_values.data(); d.sigma_polys = m->sigma_polys.data(); d.sigma_cosets = m->sigma_cosets.data();
    d.l0 = m->l[0]; d.l_last = m->l[1]; d.l_active_row = m->l[2];
    d.coset_fixed = m->coset_fixed.data(); d.coset_sigma = m->coset_sigma.data(); d.coset_l = m->coset_l.data();
    d.transcript_repr = m->transcript_repr;
}
void pk_drop(zk_ctx* ctx, PkHandle* h) {                              // g_pk_mu held
    if (h->program) (void)zk_quotient_program_release(ctx, h->program);
    for (uint64_t p : h->in_prog) if (p) (void)zk_quotient_program_release(ctx, p);
    for (uint64_t p : h->tab_prog) if (p) (void)zk_quotient_program_release(ctx, p);
    if (h->mem && --h->mem->holders == 0) {
        for (void* p : h->mem->owned) (void)zk_dev_free(ctx, p);
        delete h->mem;
    }
    delete h;
}
}  // namespace

extern "C" int zk_plonk_pk_build(zk_ctx* ctx, const zk_plonk_pk_host* host, uint64_t srs_g, uint64_t srs_g_lagrange, uint64_t* pk) ZK_ABI_TRY {
    if (!ctx || !host || !pk) return ZK_ERR_ARG;
    if (host->struct_size != sizeof(zk_plonk_pk_host))
        return pk_fail(ctx, ZK_ERR_ARG, "zk_plonk_pk_build: zk_plonk_pk_host.struct_size %u, expected %zu (ABI version %u)", host->struct_size, sizeof(zk_plonk_pk_host), ZK_ABI_VERSION);
    const uint32_t k = host->k, L = host->n_lookups;
    if (k < 1 || k > 27 || host->cs_degree < 3 || host->transcript > 2 || host->draw_schedule != 1 || !host->transcript_repr || !host->evaluator_zkq1) return ZK_ERR_ARG;
    if ((host->n_fixed && !host->fixed_values) || (host->n_perm_columns && (!host->sigma_values || !host->perm_columns)) || (host->n_advice_queries && !host->advice_queries) ||
        (host->n_fixed_queries && !host->fixed_queries) ||
        (L && (!host->lookup_input_zkq1 || !host->lookup_input_zkq1_len || !host->lookup_table_zkq1 || !host->lookup_table_zkq1_len || !host->lookup_table_key)))
        return ZK_ERR_ARG;
    const size_t n = (size_t)1 << k, col_bytes = n * 32;
    if ((size_t)host->blinding_factors + 2 >= n) return ZK_ERR_ARG;
    uint32_t ek = k;                                                  // EvaluationDomain::new(j, k): the smallest extended domain that holds a quotient of degree (j - 1) n
    while (((size_t)1 << ek) < n * (host->cs_degree - 1)) ek++;
    if (ek > 27) return ZK_ERR_LIMIT;
    const size_t ext_bytes = (size_t)32 << ek;
    PkHandle* h = new PkHandle();
    struct Undo { zk_ctx* ctx; PkHandle* h; ~Undo() { if (h) { std::lock_guard<std::mutex> lk(g_pk_mu); pk_drop(ctx, h); } } } undo{ctx, h};      // every way out but the last line drops the half-built key
    h->mem = new PkMem();
    h->mem->holders = 1;
    PkMem* m = h->mem;
    auto fail = [&](int rc) { return rc; };
    auto alloc = [&](size_t bytes) -> void* {
        m->owned.reserve(m->owned.size() + 1);                         // (the slot first: a buffer is never allocated without an owner to free it)
        void* p = nullptr;
        if (zk_dev_alloc(ctx, bytes, &p) != ZK_OK) return nullptr;
        m->owned.push_back(p);
        return p;
    };
    // values -> (values, polys, cosets); `src` are host columns, or device columns that are borrowed as they are
    // a sharded key keeps only the cosets this rank's quotient units live on (the unit rule of zk_plonk_pk_desc), n values per column
    const uint32_t world = host->shard_world > 1 ? host->shard_world : 1;
    std::vector<uint32_t> my_cosets;
    if (world > 1) {
        if (host->shard_rank >= world || n % world || !host->allgather) return fail(ZK_ERR_ARG);
        const uint32_t n_cosets = 1u << (ek - k);
        uint32_t parts = 1;
        if (world > n_cosets && world % n_cosets == 0) { const uint32_t p = world / n_cosets; if ((p & (p - 1)) == 0 && n % p == 0) parts = p; }
        const size_t n_units = (size_t)n_cosets * parts, slots = (n_units + world - 1) / world;
        for (size_t u = (size_t)host->shard_rank * slots; u < (size_t)(host->shard_rank + 1) * slots && u < n_units; u++)
            if (my_cosets.empty() || my_cosets.back() != u / parts) my_cosets.push_back((uint32_t)(u / parts));
    }
    // a single GPU needs h(X)'s numerator on cs_degree - 1 cosets only (zk_cosets_to_pieces_dev): when that is fewer than the 2^(ek - k) of the extended domain the key
    // keeps cosets 0 .. cs_degree-2, n values per column, and no extended form at all (tunable "quot_piece_cosets", default on)
    bool whole_domain = world == 1;
    if (world == 1 && host->cs_degree - 1 < (1u << (ek - k)) && host->cs_degree - 1 <= 8) {
        int on = 1;
        (void)zk_tune_get(ctx, "quot_piece_cosets", &on);
        if (on) { whole_domain = false; for (uint32_t j = 0; j + 1 < host->cs_degree; j++) my_cosets.push_back(j); }
    }
    // coeffs -> extended cosets (`cosets`) or the cosets this key keeps (`by_coset`, [coset][column])
    auto to_cosets = [&](std::vector<void*>& pl, std::vector<const void*>& cosets, std::vector<const void*>& by_coset) -> int {
        const size_t count = pl.size();
        if (whole_domain) {
            std::vector<void*> cs(count);
            for (auto& c : cs) { c = alloc(ext_bytes); if (!c) return ZK_ERR_HIP; }
            if (count) PK(zk_coeff_to_extended_batch_dev(ctx, (const void* const*)pl.data(), cs.data(), count, k, ek));
            cosets.assign(cs.begin(), cs.end());
            return ZK_OK;
        }
        for (uint32_t j : my_cosets) {
            std::vector<void*> cs(count);
            for (auto& c : cs) { c = alloc(col_bytes); if (!c) return ZK_ERR_HIP; }
            if (count) PK(zk_coeff_to_coset_batch_dev(ctx, (const void* const*)pl.data(), cs.data(), count, k, ek, j));
            by_coset.insert(by_coset.end(), cs.begin(), cs.end());
        }
        return ZK_OK;
    };
    auto three_forms = [&](const void* const* src, size_t count, bool on_device, std::vector<const void*>& values, std::vector<const void*>& polys, std::vector<const void*>& cosets,
                           std::vector<const void*>& by_coset) -> int {
        std::vector<void*> dst, pl;
        std::vector<const void*> hs;
        for (size_t i = 0; i < count; i++) {
            if (!src[i]) return ZK_ERR_ARG;
            if (on_device) values.push_back(src[i]);
            else { void* v = alloc(col_bytes); if (!v) return ZK_ERR_HIP; values.push_back(v); dst.push_back(v); hs.push_back(src[i]); }
            void* p = alloc(col_bytes);
            if (!p) return ZK_ERR_HIP;
            pl.push_back(p);
        }
        if (!dst.empty()) PK(zk_dev_upload_batch(ctx, dst.data(), hs.data(), dst.size(), col_bytes));
        for (size_t i = 0; i < count; i++) PK(zk_dev_copy(ctx, pl[i], values[i], col_bytes));
        if (count) PK(zk_lagrange_to_coeff_batch_dev(ctx, pl.data(), count, k));
        polys.assign(pl.begin(), pl.end());
        return to_cosets(pl, cosets, by_coset);
    };
    int rc = three_forms(host->fixed_values, host->n_fixed, host->values_on_device != 0, m->fixed_values, m->fixed_polys, m->fixed_cosets, m->coset_fixed);
    if (!rc) rc = three_forms(host->sigma_values, host->n_perm_columns, host->values_on_device != 0, m->sigma_values, m->sigma_polys, m->sigma_cosets, m->coset_sigma);
    if (rc) return fail(rc);
    {   // l0 = [row 0], l_last = [row n - bf - 1], l_active_row = [rows below it]: Lagrange columns -> extended cosets (keygen.rs)
        const size_t last = n - host->blinding_factors - 1;
        std::vector<uint64_t> col(3 * n * 4, 0);
        const Fe one = Fr::one();
        memcpy(&col[0], one.v, 32);
        memcpy(&col[(n + last) * 4], one.v, 32);
        for (size_t i = 0; i < last; i++) memcpy(&col[(2 * n + i) * 4], one.v, 32);
        struct Tmp { zk_ctx* ctx; std::vector<void*> v; ~Tmp() { for (void* p : v) if (p) (void)zk_dev_free(ctx, p); } } t{ctx, std::vector<void*>(3, nullptr)};
        std::vector<void*>& tmp = t.v;
        const void* hs[3];
        for (int i = 0; i < 3; i++) {
            if (zk_dev_alloc(ctx, col_bytes, &tmp[i]) != ZK_OK) return fail(ZK_ERR_HIP);
            hs[i] = &col[(size_t)i * n * 4];
        }
        rc = zk_dev_upload_batch(ctx, tmp.data(), hs, 3, col_bytes);
        if (!rc) rc = zk_lagrange_to_coeff_batch_dev(ctx, tmp.data(), 3, k);
        std::vector<const void*> ext;
        if (!rc) rc = to_cosets(tmp, ext, m->coset_l);
        if (rc) return fail(rc);
        for (int i = 0; i < 3 && whole_domain; i++) m->l[i] = (void*)ext[i];
    }
    rc = zk_quotient_program_load(ctx, host->evaluator_zkq1, host->evaluator_zkq1_len, &h->program);
    h->in_prog.assign(L, 0); h->tab_prog.assign(L, 0);
    for (uint32_t l = 0; l < L && !rc; l++) {
        rc = zk_quotient_program_load(ctx, host->lookup_input_zkq1[l], host->lookup_input_zkq1_len[l], &h->in_prog[l]);
        if (!rc) rc = zk_quotient_program_load(ctx, host->lookup_table_zkq1[l], host->lookup_table_zkq1_len[l], &h->tab_prog[l]);
    }
    if (rc) return fail(rc);
    m->perm_columns.assign(host->perm_columns, host->perm_columns + 2 * (size_t)host->n_perm_columns);
    m->advice_queries.assign(host->advice_queries, host->advice_queries + 2 * (size_t)host->n_advice_queries);
    m->fixed_queries.assign(host->fixed_queries, host->fixed_queries + 2 * (size_t)host->n_fixed_queries);
    m->table_key.assign(host->lookup_table_key, host->lookup_table_key + L);
    m->perm_columns.push_back(0); m->advice_queries.push_back(0); m->fixed_queries.push_back(0); m->table_key.push_back(0);      // .data() of an empty vector may be null: the prover refuses null arrays
    h->in_prog.push_back(0); h->tab_prog.push_back(0);
    for (auto* v : {&m->fixed_values, &m->fixed_polys, &m->fixed_cosets, &m->sigma_values, &m->sigma_polys, &m->sigma_cosets, &m->coset_fixed, &m->coset_sigma, &m->coset_l}) v->push_back(nullptr);
    memcpy(m->transcript_repr, host->transcript_repr, 32);
    zk_plonk_pk_desc shape;
    ZK_STRUCT_INIT(shape);
    shape.k = k; shape.extended_k = ek; shape.cs_degree = host->cs_degree; shape.blinding_factors = host->blinding_factors;
    shape.n_fixed = host->n_fixed; shape.n_advice = host->n_advice; shape.n_instance = host->n_instance; shape.n_lookups = L; shape.n_perm_columns = host->n_perm_columns;
    shape.n_advice_queries = host->n_advice_queries; shape.n_fixed_queries = host->n_fixed_queries;
    shape.transcript = host->transcript; shape.draw_schedule = host->draw_schedule;
    shape.shard_world = host->shard_world; shape.shard_rank = host->shard_rank; shape.allgather = host->allgather; shape.allgather_user = host->allgather_user;
    pk_fill_desc(h, shape, srs_g, srs_g_lagrange);
    {
        std::lock_guard<std::mutex> lk(g_pk_mu);
        g_pk_handles[{ctx, g_pk_next}] = h;
        *pk = g_pk_next++;
    }
    undo.h = nullptr;
    return ZK_OK;
} ZK_ABI_CATCH(ctx)

extern "C" int zk_plonk_pk_share(zk_ctx* ctx, zk_ctx* owner, uint64_t owner_pk, uint64_t srs_g, uint64_t srs_g_lagrange, uint64_t* pk) ZK_ABI_TRY {
    if (!ctx || !owner || !pk) return ZK_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_pk_mu);
    auto it = g_pk_handles.find({owner, owner_pk});
    if (it == g_pk_handles.end()) return ZK_ERR_ARG;
    const PkHandle* src = it->second;
    PkHandle* h = new PkHandle();
    struct Undo { zk_ctx* ctx; PkHandle* h; ~Undo() { if (h) pk_drop(ctx, h); } } undo{ctx, h};      // (g_pk_mu is held for the whole function)
    h->mem = src->mem;
    h->mem->holders++;
    const size_t L = src->desc.n_lookups;
    h->in_prog.assign(L + 1, 0); h->tab_prog.assign(L + 1, 0);
    int rc = zk_quotient_program_share(ctx, owner, src->program, &h->program);                  // (refuses contexts on different devices)
    for (size_t l = 0; l < L && !rc; l++) {
        rc = zk_quotient_program_share(ctx, owner, src->in_prog[l], &h->in_prog[l]);
        if (!rc) rc = zk_quotient_program_share(ctx, owner, src->tab_prog[l], &h->tab_prog[l]);
    }
    if (rc) return rc;
    pk_fill_desc(h, src->desc, srs_g, srs_g_lagrange);
    g_pk_handles[{ctx, g_pk_next}] = h;
    *pk = g_pk_next++;
    undo.h = nullptr;
    return ZK_OK;
} ZK_ABI_CATCH(ctx)

extern "C" int zk_plonk_pk_release(zk_ctx* ctx, uint64_t pk) ZK_ABI_TRY {
    if (!ctx) return ZK_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_pk_mu);
    auto it = g_pk_handles.find({ctx, pk});
    if (it == g_pk_handles.end()) return ZK_ERR_ARG;
    if (it->second->in_use) it->second->released = true;             // a proof is running through this handle on another thread: it drops the handle when it returns
    else pk_drop(ctx, it->second);
    g_pk_handles.erase(it);
    return ZK_OK;
} ZK_ABI_CATCH(ctx)

extern "C" int zk_plonk_pk_descriptor(zk_ctx* ctx, uint64_t pk, const zk_plonk_pk_desc** desc) ZK_ABI_TRY {
    if (!ctx || !desc) return ZK_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_pk_mu);
    auto it = g_pk_handles.find({ctx, pk});
    if (it == g_pk_handles.end()) return ZK_ERR_ARG;
    *desc = &it->second->desc;
    return ZK_OK;
} ZK_ABI_CATCH(ctx)

extern "C" int zk_plonk_prove(zk_ctx* ctx, uint64_t pk, const void* const* advice, int advice_on_device, const void* const* instances, const uint32_t* instance_lens,
                              zk_rng_fn rng, void* rng_user, void* proof_out, size_t proof_cap, size_t* proof_len) ZK_ABI_TRY {
    if (!ctx) return ZK_ERR_ARG;
    PkHandle* h = nullptr;
    {   // the handle (descriptor, programs, its share of the columns) stays alive for the whole proof whatever other threads release meanwhile
        std::lock_guard<std::mutex> lk(g_pk_mu);
        auto it = g_pk_handles.find({ctx, pk});
        if (it == g_pk_handles.end()) return pk_fail(ctx, ZK_ERR_ARG, "zk_plonk_prove: unknown key %llu", (unsigned long long)pk);
        h = it->second;
        h->in_use++;
    }
    struct Done { zk_ctx* ctx; PkHandle* h; ~Done() { std::lock_guard<std::mutex> lk(g_pk_mu); if (--h->in_use == 0 && h->released) pk_drop(ctx, h); } } done{ctx, h};
    return zk_plonk_create_proof(ctx, &h->desc, advice, advice_on_device, instances, instance_lens, rng, rng_user, proof_out, proof_cap, proof_len);
} ZK_ABI_CATCH(ctx)

// zk_ctx_destroy (capi.hip): the keys this context still holds go with it (before its programs are released)
void zk_internal_plonk_ctx_destroyed(zk_ctx* ctx) {
    std::lock_guard<std::mutex> lk(g_pk_mu);
    for (auto it = g_pk_handles.begin(); it != g_pk_handles.end();) {
        if (it->first.first == ctx) { if (it->second->in_use) it->second->released = true; else pk_drop(ctx, it->second); it = g_pk_handles.erase(it); }
        else ++it;
    }
}

extern "C" int zk_plonk_last_phase_ms(double out[9]) ZK_ABI_TRY {
    if (!out) return ZK_ERR_ARG;
    for (int i = 0; i < 9; i++) out[i] = g_phase_ms[i];
    return ZK_OK;
} ZK_ABI_CATCH(nullptr)
extern "C" int zk_plonk_trim(zk_ctx* ctx) ZK_ABI_TRY {
    if (!ctx) return ZK_ERR_ARG;
    std::shared_ptr<Pool> p;
    {
        std::lock_guard<std::mutex> lk(g_pools_mu);
        auto it = g_pools.find(ctx);
        if (it == g_pools.end()) return ZK_OK;
        p = it->second;
        g_pools.erase(it);
    }
    std::map<size_t, std::vector<void*>> idle;
    { std::lock_guard<std::mutex> lk(p->mu); p->retired = true; idle.swap(p->free_); }      // a proof still running on this context keeps its Arena's reference: its buffers are freed as it returns them
    for (auto& kv : idle) for (void* d : kv.second) (void)zk_dev_free(ctx, d);
    zk_internal_trim_helper(ctx);                                      // the side lane's context: its transform workspaces, twiddle and coset tables (rebuilt at the next lone proof)
    return ZK_OK;
} ZK_ABI_CATCH(ctx)
