// Library context shared by the MSM / NTT / quotient translation units.
#pragma once
#include <stdarg.h>
#include <string.h>
#include <stdio.h>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "ec.cuh"
#include "../../include/zkmi355.h"
#include "abi_guard.h"

namespace zk {

// Tunables (settable through zk_tune_set; defaults chosen for MI355X)
struct Tune {
    int msm_c = 0;               // 0 = pick from n
    int msm_sort_wgs = 256;      // workgroups of the counting-sort kernels (one per CU)
    int msm_sort_threads = 1024;
    int msm_bsort_threads = 1024;    // workgroup size of the two-level sort's second level ...
    int msm_bsort_chunk = 8192;      // ... and the entries of a bin one of its workgroups sorts: one round of eight entries per lane — the kernel is a chain of load -> LDS atomic -> scattered store,
                                     // and its time follows the rounds a lane runs one after the other (2^24 scalars, c = 20: 256 threads x 65536 entries 5.8 ms of sort, this 3.6: profiles/r05/run35x)
    int msm_part_threads = 1024, msm_part_pairs = 16384; // two-level sort, first level: workgroup size and the pairs a workgroup stages in LDS (8 bytes each): a workgroup pays a scan and one global
                                                         // atomic per bin whatever it stages (2^24, c = 20: 256 x 8192 3.64 ms of sort, 1024 x 16384 3.09, 256 x 4096 5.72: profiles/r05/run357)
    int msm_wide_bins_log = 9;       // windows wider than 16 bits: log2 of the bins of the two-level sort (8 .. 11; 9 measured best at 2^24)
    int msm_two_level_sort = 0;      // 1: take the two-level bucket sort even where the one-level sort applies (tests, measurements)
    int msm_sort_batch_wgs = 2048;   // total sort workgroups aimed for by a batched call
    int msm_target_threads = 1 << 19;  // sub-bucket count the accumulate launch aims for
    int msm_min_chunk = 16;      // min pairs per accumulate thread
    int msm_max_chunk = 48;        // fixed-size sub-buckets keep all 64 lanes of a wave equally loaded (profiles/r01); 32 -> 48 with the 29-bit chain, whose per-chain entry and exit cost 6 products (profiles/r03: -0.6 %)
    int msm_max_chunk_wide = 128;  // ... of tables with windows wider than 16 bits
    int msm_merge_fanin = 8;
    int msm_tree_fanin = 2;
    int msm_block = 128;         // threads per workgroup of the curve-arithmetic kernels
    int prover_lane_priority = 1;  // the helper context's stream: 1 = the device's lowest stream priority, 2 = its highest, 0 = the default priority (capi.hip, zk_internal_helper_ctx: a queue of its own)
    int prover_side_lane = 1;    // zk_plonk_create_proof (single-GPU keys on the extended domain): lagrange_to_coeff + coeff_to_extended of a phase's columns on the helper context while the phase's commitments run; 1 = when at most two proofs are in flight in the process, 2 = always, 0 = never
    int msm_runs = 1;            // commit run-heavy columns through adjacent differences against the prefix-sum table (when the table has one)
    int ntt_tile_log = 10;       // log2(elements) of the LDS tile of one NTT workgroup (sweep: profiles/r01/run6_ntt_plan_sweep.txt)
    int ntt_threads = 256;
    int ntt_max_radix_log = 8;
    int ntt_plan = 0;            // measurement knob: three radices as decimal digits for the transforms they fit (0 = balanced)
    int ntt_col_major = 1;               // strided passes of a batch walk the columns of one tile before the next tile (NttPassArgs::col_major)
    int ntt_coset_table = 1;             // coset transforms of two passes and more: the pre-scaling ZETA^(m mod 3) * ext_omega^(coset * m) from ONE table per coset (32 B/element, one product) instead of two-level powers (up to 2.67 products)
    int ntt_full_twiddle_max_log = 24;   // up to this size inter-pass twiddles come from full HBM tables (32 B/element/pass)
    int ntt_ws_limit_mb = 24576; // a batched transform's out-of-place workspace (columns x N x 32 B) is capped here: larger batches run in slices of columns (k >= 22)
    int ntt_fuse_scale = 1;      // the 1/n of an inverse transform rides on the last strided pass's inter-pass twiddle table (one product per element fewer in the final pass)
    int ntt_quarter_input = 1;   // coeff_to_extended: skip the arithmetic of the first two stages when 3/4 of the input is the zero padding
    int vec_block = 256;
    int quot_threads = 128;
    int quot_piece_cosets = 1;   // zk_plonk_pk_build on one GPU: keep cosets 0 .. cs_degree-2 of the key's columns instead of their extended forms when cs_degree - 1 < 2^(extended_k - k) (zk_cosets_to_pieces_dev)
    int quot_degree_split = 1;   // quotient compiler + zk_plonk_create_proof: identities of degree <= 3 are evaluated on two cosets of the extended domain only and join h(X) through
                                 // zk_cosets_to_pieces_dev; 0 = every identity on every row, halo2's bytes also for a witness that violates its circuit
    int quot_group_factors = 1;  // quotient compiler: the identities of the permutation and lookup arguments are summed per common factor (l_0, l_last, l_active_row) and multiplied by it once
    int quot_factor_horner = 1;  // quotient compiler: q * Horner([a_j], theta) for a theta-compression whose parts all carry the factor q (selector-switched lookups): m - 1 products fewer per row
    int quot_remat_ops = 4;      // quotient compiler: a shared sub-expression of at most this many operations ...
    int quot_remat_distance = 24;   // ... is recomputed when its previous copy lies further back than this many micro-ops
    int quot_jit = 0;            // zk_quotient_program_load: also generate straight-line kernels for the program with hiprtc (quotient_jit.hip) and run those instead of the interpreter; 0 = interpreter only; 1 = the kernels a single-GPU proof launches (the degree parts of a split program, else the whole program); 2 = whole program and parts
    int quot_jit_waves = 0;      // ... amdgpu_waves_per_eu of the generated kernels (0 = the compiler's choice; 4 = the interpreter's budget of 128 VGPRs, which costs some kernels a few spills)
    int quot_jit_group = 200;    // ... products per generated kernel (swept 24 .. 400 on the sgx-shaped program, profiles/r05/run303: straight-line code streams well past the instruction cache; 200 = three to four kernels per program)
    int lookup_force_generic_sort = 0;   // tests: take the every-digit sort of permute_expression_pair even when the 64-bit window sort is exact
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 4096;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct DevTmp {               // a device allocation that lives until the end of its scope unless release() hands it on (error returns and exceptions free it)
    void* p = nullptr;
    DevTmp() = default;
    DevTmp(const DevTmp&) = delete;
    DevTmp& operator=(const DevTmp&) = delete;
    ~DevTmp() { if (p) (void)hipFree(p); }
    void* release() { void* q = p; p = nullptr; return q; }
    void reset() { if (p) (void)hipFree(p); p = nullptr; }
};

struct TableMem {             // the window-expanded table in HBM; shared (refcounted) by every context of the process that registered or was lent it
    void* p = nullptr;
    int device = 0;
    ~TableMem() { if (p) { (void)hipSetDevice(device); (void)hipFree(p); } }
};
struct BaseTable {
    void* d_table = nullptr;  // [W][n] affine points: table[j*n + i] = 2^(c*j) * P_i
    size_t n = 0;
    int c = 0, W = 0;
    std::shared_ptr<TableMem> mem;
    // optional (zk_bases_enable_runs): the same expansion of the PREFIX SUMS S_i = P_0 + ... + P_i.  sum_i a_i P_i = sum_i (a_i - a_(i+1)) S_i, and
    // zero digits cost nothing, so a column with long runs of equal values (a sorted lookup column, a constant column) is committed through its
    // adjacent differences at the cost of its run boundaries
    void* d_runs_table = nullptr;
    std::shared_ptr<TableMem> runs_mem;
};

struct TwiddleSet {           // per (omega, log_n)
    uint32_t log_n = 0;
    u256 omega;
    void* d_lo = nullptr;     // omega^e, e < 2^lo_bits
    void* d_hi = nullptr;     // omega^(e << lo_bits)
    uint32_t lo_bits = 0;
    void* d_stage[3] = {nullptr, nullptr, nullptr};  // per pass: omega_R^k, k < R/2
    void* d_stage_sh[3] = {nullptr, nullptr, nullptr};   // per pass: the same twiddles as Shoup pairs (w, floor(w 2^261 / p)), 80 bytes each
    void* d_stage29[3] = {nullptr, nullptr, nullptr};    // per pass: the same twiddles x 2^261 (Montgomery operands of the 29-bit form)
    void* d_full[3] = {nullptr, nullptr, nullptr};   // per non-final pass: inter-pass twiddles in store order
    uint32_t radix_log[3] = {0, 0, 0};
    int passes = 0;
    // a transform that ends in a multiplication of every output by one constant (lagrange_to_coeff's 1/n, extended_to_coeff's 1/2^extended_k) takes it through the
    // inter-pass twiddles of its LAST strided pass instead (that pass multiplies every element anyway): such a set's d_full[passes - 2] holds twiddle * scale
    bool scale_fused = false;
    u256 fused_scale;
    uint64_t stamp = 0;       // zk_ctx::twiddle_clock at the last use (ntt.hip keeps the 16 most recently used sets)
};

struct QuotProgram;  // quotient.hip

}  // namespace zk

struct zk_ctx {
    int device = 0;
    std::mutex mu;            // calls are serialised per context (thread-safe, blocking)
    char err[640] = {0};      // zk_last_error: a fixed buffer, so that reporting a failure (an exhausted heap included) allocates nothing
    hipStream_t stream = nullptr;
    zk::Tune tune;
    uint64_t next_handle = 1;
    std::map<uint64_t, zk::BaseTable> bases;
    std::list<zk::TwiddleSet> twiddles;    // (a list: a set's address survives the eviction of another)
    uint64_t twiddle_clock = 0;
    std::map<uint64_t, void*> coset_tables;                           // ntt.hip: (k, extended_k, coset) -> pre-scaling table of a coset transform
    std::map<uint64_t, std::shared_ptr<zk::QuotProgram>> programs;   // compiled micro-programs are immutable once loaded: contexts of one device may share them (zk_quotient_program_share)
    std::map<uint64_t, std::vector<uint32_t>> lookup_tie_hint;   // lookupperm.hip: columns whose rows tied on the sort window in the previous call of the same shape
    // workspaces (grow-only)
    zk::DevBuf ws_scalars, ws_sorted, ws_mid, ws_small, ws_sub0, ws_sub1, ws_cls0, ws_cls1, ws_tmp, ws_ntt, ws_ntt_in, ws_pts, ws_runs, ws_quot, ws_quot_state;
    // last-call kernel timing (ms), filled when timing is enabled
    bool timing = false;
    std::map<std::string, double> last_ms;
    struct PendingTimer { const char* label; hipEvent_t a, b; };
    std::vector<PendingTimer> pending_timers;   // event pairs of asynchronous entry points, read by zk_timing_get
    // a second context of the same device that belongs to this one (created on first use, destroyed with it): zk_plonk_create_proof runs the transforms of columns whose
    // values are final there — its own stream, workspaces and lock — while this context commits them (prover.hip SideLane); timings and tunables pass through
    zk_ctx* helper = nullptr;

    int fail(int code, const char* fmt, ...) {
        char buf[sizeof err];                  // (callers pass zk_last_error's own text back in as an argument)
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        memcpy(err, buf, sizeof err);
        return code;
    }
};

#define ZK_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess)                                                                \
            return ctx->fail(ZK_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)
#define ZK_CHECK_LAUNCH() ZK_HIP(hipGetLastError())

namespace zk {
inline void resolve_pending_timers(zk_ctx* ctx);
struct EvTimer {
    zk_ctx* ctx; const char* label; hipEvent_t a = nullptr, b = nullptr; bool on;
    EvTimer(zk_ctx* c, const char* l) : ctx(c), label(l), on(c->timing) {
        if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, ctx->stream); }
    }
    void stop() { if (on) (void)hipEventRecord(b, ctx->stream); }
    void resolve() {
        if (!on) return;
        (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        on = false;
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        ctx->last_ms[label] += ms; ctx->last_ms[std::string(label) + "#n"] += 1.0;
    }
    // for entry points that return without waiting for their kernels (the NTT passes): the event pair is read at the next zk_timing_get — or here, when a long timed run
    // that never asks has let the list grow (one pair per pass otherwise, without bound)
    void defer() {
        if (!on) return;
        if (ctx->pending_timers.size() >= 4096) resolve_pending_timers(ctx);
        ctx->pending_timers.push_back({label, a, b});
        on = false;
    }
    ~EvTimer() { if (on) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } }       // an early return between the constructor and resolve() / defer()
    EvTimer(const EvTimer&) = delete;
    EvTimer& operator=(const EvTimer&) = delete;
};
inline void resolve_pending_timers(zk_ctx* ctx) {
    std::vector<zk_ctx::PendingTimer> list;
    list.swap(ctx->pending_timers);                                   // (the events are destroyed whatever the bookkeeping below does)
    struct Drop { std::vector<zk_ctx::PendingTimer>& l; ~Drop() { for (auto& t : l) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); } } } drop{list};
    for (auto& t : list) {
        (void)hipEventSynchronize(t.b);
        float ms = 0; (void)hipEventElapsedTime(&ms, t.a, t.b);
        ctx->last_ms[t.label] += ms; ctx->last_ms[std::string(t.label) + "#n"] += 1.0;
    }
}
inline void drop_pending_timers(zk_ctx* ctx) {                        // zk_ctx_destroy
    for (auto& t : ctx->pending_timers) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    ctx->pending_timers.clear();
}

// implemented in the respective translation units
int msm_register(zk_ctx* ctx, const void* pts, size_t n, bool on_device, uint64_t* handle);
int msm_release(zk_ctx* ctx, uint64_t handle);
int msm_share(zk_ctx* dst, const BaseTable& bt, uint64_t* handle);
int msm_enable_runs(zk_ctx* ctx, uint64_t handle);
int msm_run(zk_ctx* ctx, uint64_t handle, const void* scalars, size_t n, bool on_device, void* out, int partial);
int msm_run_batch(zk_ctx* ctx, uint64_t handle, const void* const* scalars, size_t nb, size_t n, bool on_device, void* out, int partial);
int g1_sum_xyzz_host(const void* xyzz, size_t count, void* out_jac);
int g1_fixed_base_mul(zk_ctx* ctx, const void* d_scalars, size_t n, void* d_out_affine);
void release_gtab(zk_ctx* ctx);
int msm_set_lds_attr();
struct NttFuse {              // optional fused pre/post operations (EvaluationDomain wrappers)
    const void* src = nullptr; // read the input from here instead of `a` (device)
    uint32_t n_valid = 0;      // inputs with index >= n_valid are zero (0 = all valid)
    int pre_zeta = 0;          // input i *= ZETA^(i mod 3)            (coeff_to_extended)
    int post_scale = 0;        // output *= scale
    u256 scale;
    int post_zeta_inv = 0;     // output i *= ZETA^-(i mod 3)          (extended_to_coeff)
    const void* cs_lo = nullptr; const void* cs_hi = nullptr; uint32_t cs_lo_bits = 0, cs_stride = 0, cs_log = 0;   // coset pre-scaling (ntt.hip)
    const void* pre_full = nullptr;   // the same pre-scaling AND pre_zeta as one table of operands x 2^261 (transforms whose first pass runs on 29-bit limbs take it instead)
};
int ntt_dev(zk_ctx* ctx, void* d_a, uint32_t log_n, const u256& omega, const NttFuse* fuse);
void release_twiddles(zk_ctx* ctx);
void release_programs(zk_ctx* ctx);
int quotient_program_share(zk_ctx* ctx, zk_ctx* owner, uint64_t owner_prog, uint64_t* prog);
}  // namespace zk
