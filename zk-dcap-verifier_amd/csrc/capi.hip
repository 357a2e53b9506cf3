// extern "C" surface of libzkmi355.so (see include/zkmi355.h for the contract and the reference
// call sites each entry point stands in for).
#include <string.h>
#include <new>
#include <system_error>
#include "ctx.h"

namespace zk {
int ntt_set_lds_attr();
int quotient_set_lds_attr();
int domain_lagrange_to_coeff(zk_ctx* ctx, void* d_a, uint32_t k);
int domain_coeff_to_lagrange(zk_ctx* ctx, void* d_a, uint32_t k);
int domain_lagrange_to_coeff_batch(zk_ctx* ctx, void* const* cols, size_t count, uint32_t k);
int domain_coeff_to_extended_batch(zk_ctx* ctx, const void* const* coeffs, void* const* outs, size_t count, uint32_t k, uint32_t ek);
int ntt_dev_batch(zk_ctx* ctx, void* const* h_cols, const void* const* h_srcs, size_t count, uint32_t log_n, const u256& omega, const NttFuse* fuse);
int domain_coeff_to_extended(zk_ctx* ctx, const void* d_coeff, uint32_t k, uint32_t ek, void* d_out);
int domain_extended_to_coeff(zk_ctx* ctx, void* d_a, uint32_t k, uint32_t ek);
int domain_divide_by_vanishing(zk_ctx* ctx, void* d_a, uint32_t k, uint32_t ek);
int fr_vec_op(zk_ctx* ctx, int op, const void* a, const void* b, void* out, size_t n, const u256* scalar);
int permutation_product(zk_ctx* ctx, const void* const* values, const void* const* sigmas, size_t count, uint32_t k, const void* beta, const void* gamma,
                        const void* delta_start, const void* z_init, const void* blinding, uint32_t bf, void* d_z, void* h_last_z);
int lookup_product(zk_ctx* ctx, const void* cin, const void* ctab, const void* pin, const void* ptab, uint32_t k, const void* beta, const void* gamma,
                   const void* blinding, uint32_t bf, void* d_z);
int permutation_product_all(zk_ctx* ctx, const void* const* values, const void* const* sigmas, size_t m, uint32_t chunk_len, uint32_t k, const void* beta,
                            const void* gamma, const void* blinding, uint32_t bf, void* const* d_zs);
int lookup_product_batch(zk_ctx* ctx, const void* const* cols4, size_t count, uint32_t k, const void* beta, const void* gamma, const void* blinding,
                         uint32_t bf, void* const* d_zs);
int eval_polynomial_batch(zk_ctx* ctx, const void* const* polys, size_t count, size_t n, const void* points, void* out);
int kate_division(zk_ctx* ctx, const void* d_a, size_t n, const void* b_host, void* d_q);
int fr_lincomb(zk_ctx* ctx, const void* const* polys, const void* scalars, size_t count, size_t n, void* d_out);
int pk_load(zk_ctx* ctx, uint64_t prog, const void* const* fixed, const void* const* sigma, const void* l0, const void* l_last, const void* l_active,
            int form, uint64_t* handle);
int pk_release(zk_ctx* ctx, uint64_t h);
void release_pks(zk_ctx* ctx);
int evaluate_h_host(zk_ctx* ctx, uint64_t pkh, const void* const* advice, const void* const* instance, const void* const* perm_products,
                    const void* const* lk_product, const void* const* lk_input, const void* const* lk_table, const void* challenges,
                    const void* beta, const void* gamma, const void* theta, const void* y, int finish, void* out);
int lookup_permute(zk_ctx* ctx, const void* d_input, const void* d_table, uint32_t k, uint32_t blinding_factors, const void* h_blind_input,
                   const void* h_blind_table, void* d_out_input, void* d_out_table);
int lookup_permute_batch(zk_ctx* ctx, const void* const* d_inputs, const void* const* d_tables, size_t count, uint32_t k, uint32_t blinding_factors,
                         const void* h_blind_inputs, const void* h_blind_tables, void* const* d_out_inputs, void* const* d_out_tables);
int g1_decompress(zk_ctx* ctx, const void* d_bytes, size_t n, uint32_t sign_bit, void* d_out_affine, uint32_t* n_bad_host);
int g1_compress(zk_ctx* ctx, const void* d_affine, size_t n, uint32_t sign_bit, void* d_bytes);
int g1_ntt(zk_ctx* ctx, const void* d_affine_in, uint32_t log_n, const void* omega_host, const void* scale_host, void* d_affine_out);
int quotient_program_load(zk_ctx* ctx, const void* blob, size_t len, uint64_t* prog);
int quotient_program_release(zk_ctx* ctx, uint64_t prog);
int quotient_program_info(zk_ctx* ctx, uint64_t prog, uint32_t* n_instr, uint32_t* n_slots, uint32_t* n_columns);
int quotient_program_kernels(zk_ctx* ctx, uint64_t prog, uint32_t* n_kernels);
int quotient_program_opmix(zk_ctx* ctx, uint64_t prog, uint32_t part, uint32_t counts[9]);
int quotient_run(zk_ctx* ctx, uint64_t prog, const zk_quotient_args* args, int coset, uint64_t row_lo, uint64_t row_count, int part, uint32_t low_cosets);
int quotient_program_split(zk_ctx* ctx, uint64_t prog, uint32_t* low_cosets, uint32_t* n_instr_high, uint32_t* n_instr_low);
int domain_coeff_to_coset_batch(zk_ctx* ctx, const void* const* coeffs, void* const* outs, size_t count, uint32_t k, uint32_t ek, uint32_t coset);
int fr_interleave(zk_ctx* ctx, const void* const* h_cosets, size_t count, size_t n, void* d_out);
int domain_cosets_to_pieces(zk_ctx* ctx, void* const* h_numer, uint32_t q, uint32_t k, uint32_t ek, void* const* h_pieces);
}  // namespace zk
using namespace zk;

// What an exception that reached the C ABI becomes (abi_guard.h).  Called inside a catch (...) handler: rethrows to classify, never throws itself.
int zk::abi_exception(zk_ctx* ctx, const char* fn) noexcept {
    int code = ZK_ERR_HIP;
    char msg[sizeof ctx->err];
    try { throw; }
    catch (const zk::AbiError& e) { code = e.code; snprintf(msg, sizeof msg, "%s: %s", fn, e.what ? e.what : "internal error"); }
    catch (const std::bad_alloc&) { code = ZK_ERR_LIMIT; snprintf(msg, sizeof msg, "%s: out of host memory (std::bad_alloc)", fn); }
    catch (const std::system_error& e) { snprintf(msg, sizeof msg, "%s: %s (std::system_error %d: no thread or lock to be had)", fn, e.what(), e.code().value()); }
    catch (const std::exception& e) { snprintf(msg, sizeof msg, "%s: unexpected C++ exception: %s", fn, e.what()); }
    catch (...) { snprintf(msg, sizeof msg, "%s: unexpected C++ exception", fn); }
    if (ctx) {
        try { std::lock_guard<std::mutex> lk(ctx->mu); memcpy(ctx->err, msg, sizeof msg); }
        catch (...) { memcpy(ctx->err, msg, sizeof msg); }            // (the lock itself failed: the text still goes out)
    }
    return code;
}

#ifdef ZK_FAULT_INJECT
// TEST-ONLY (emulator build): the n-th host allocation / the next thread start of the calling thread fails.  Replacing operator new inside this test library is
// benign for the rest of the process: both sides end in malloc / free.
#include <atomic>
namespace {
thread_local long g_fail_alloc_in = 0;      // > 0: allocations of THIS thread left until one throws
thread_local long g_alloc_count = 0;
std::atomic<long> g_fail_alloc_any{0};      // > 0: allocations of ANY thread of the library left until one throws (reaches the prover's helper threads)
thread_local int g_fail_thread = 0;         // > 0: thread starts of this thread left until one throws
}
// (not under ASan / TSan: their runtimes own operator new and delete — a second replacement would pair their new with this delete; the sanitizer runs keep the
//  thread-start hook and the device-buffer census, the allocation ladder belongs to the plain emulator build)
#if defined(__has_feature)
#if __has_feature(address_sanitizer) || __has_feature(thread_sanitizer)
#define ZK_NO_NEW_HOOK 1
#endif
#endif
#ifndef ZK_NO_NEW_HOOK
void* operator new(size_t n) {
    g_alloc_count++;
    if (g_fail_alloc_in > 0 && --g_fail_alloc_in == 0) throw std::bad_alloc();
    if (g_fail_alloc_any.load(std::memory_order_relaxed) > 0 && g_fail_alloc_any.fetch_sub(1) == 1) throw std::bad_alloc();
    void* p = malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void* operator new[](size_t n) { return operator new(n); }
void operator delete(void* p) noexcept { free(p); }
void operator delete[](void* p) noexcept { free(p); }
void operator delete(void* p, size_t) noexcept { free(p); }
void operator delete[](void* p, size_t) noexcept { free(p); }
#endif
void zk::fault_thread_tick() { if (g_fail_thread > 0 && --g_fail_thread == 0) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again), "std::thread"); }
extern "C" {
// not in include/zkmi355.h: test hooks of the emulator build (tests/test_abi_no_throw.py)
int zk_test_alloc_hook_present(void) {
#ifdef ZK_NO_NEW_HOOK
    return 0;
#else
    return 1;
#endif
}
void zk_test_fail_alloc(long nth) { g_fail_alloc_in = nth; g_alloc_count = 0; }          // the nth host allocation of the calling thread from now on throws (0 = disarm)
void zk_test_fail_alloc_any_thread(long nth) { g_fail_alloc_any.store(nth); }            // ... of any thread
long zk_test_alloc_count(void) { return g_alloc_count; }
void zk_test_fail_thread(int nth) { g_fail_thread = nth; }                              // the nth std::thread the calling thread starts from now on fails
#ifdef ZK_EMU
long zk_test_live_device_allocs(void) { return emu_live_device_allocs.load(); }
#endif
}  // extern "C"
#endif

void zk_internal_plonk_ctx_destroyed(zk_ctx* ctx);   // prover.hip
// prover.hip is a client of the public ABI and does not see zk_ctx's members: its own argument errors reach zk_last_error through this
int zk_internal_fail(zk_ctx* ctx, int code, const char* msg) { return ctx ? ctx->fail(code, "%s", msg) : code; }
#define LOCK std::lock_guard<std::mutex> lk__(ctx->mu)
#define NEED_CTX if (!ctx) return ZK_ERR_ARG

extern "C" {

const char* zk_version(void) ZK_ABI_TRY {
#ifdef ZK_EMU
    return "zkmi355 0.1 (EMULATED kernels - test build, not the product)";
#else
    return "zkmi355 0.1 gfx950";
#endif
} ZK_ABI_CATCH_VALUE(nullptr, "zkmi355: internal error")

uint32_t zk_abi_version(void) ZK_ABI_TRY { return ZK_ABI_VERSION; } ZK_ABI_CATCH_VALUE(nullptr, 0u)
uint32_t zk_abi_struct_size(const char* name) ZK_ABI_TRY {
    if (!name) return 0;
    if (!strcmp(name, "zk_quotient_args")) return (uint32_t)sizeof(zk_quotient_args);
    if (!strcmp(name, "zk_plonk_pk_desc")) return (uint32_t)sizeof(zk_plonk_pk_desc);
    if (!strcmp(name, "zk_plonk_pk_host")) return (uint32_t)sizeof(zk_plonk_pk_host);
    return 0;
} ZK_ABI_CATCH_VALUE(nullptr, 0u)

int zk_ctx_create(int device_id, zk_ctx** out) ZK_ABI_TRY {
    if (!out) return ZK_ERR_ARG;
    *out = nullptr;
#ifndef ZK_EMU
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return ZK_ERR_NODEV;
    if (hipSetDevice(device_id) != hipSuccess) return ZK_ERR_NODEV;
#endif
    std::unique_ptr<zk_ctx> ctx(new zk_ctx());
    ctx->device = device_id;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return ZK_ERR_HIP;
    msm_set_lds_attr();
    ntt_set_lds_attr();
    quotient_set_lds_attr();
    *out = ctx.release();
    return ZK_OK;
} ZK_ABI_CATCH(nullptr)

}  // extern "C"
// the helper context of `ctx` (ctx.h): same device, made on first use, the caller's tunables as they are now (internal: prover.hip)
zk_ctx* zk_internal_helper_ctx(zk_ctx* ctx) {
    if (!ctx) return nullptr;
    zk_ctx* h;
    {
        LOCK;
        h = ctx->helper;
    }
    if (!h) {                                                          // made outside the lock (a context is a stream and three attribute calls); of two proofs that start together one wins, the other's goes
        zk_ctx* made = nullptr;
        if (zk_ctx_create(ctx->device, &made) != ZK_OK) return nullptr;
#ifndef ZK_EMU
        {   // The lane only helps if its stream runs on ANOTHER hardware queue than the proof's: the runtime spreads a process's streams of one priority over four queues
            // (GPU_MAX_HW_QUEUES), so beside four proving contexts the fifth stream shares one — with its own proof's stream one time in four, and the two then run in
            // turn (profiles/r05/run360: 63.8 ms for the proof alone, 57.5 when the queues differ).  Streams of another priority get queues of their own.
            int lane_prio;
            { LOCK; lane_prio = ctx->tune.prover_lane_priority; }
            int least = 0, greatest = 0;
            hipStream_t s = nullptr;
            if (lane_prio && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest &&
                hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lane_prio == 1 ? least : greatest) == hipSuccess) {
                (void)hipStreamDestroy(made->stream);
                made->stream = s;
            }
        }
#endif
        {
            LOCK;
            if (!ctx->helper) { ctx->helper = made; made = nullptr; }
            h = ctx->helper;
        }
        if (made) zk_ctx_destroy(made);
    }
    zk::Tune t;
    bool timing;
    {
        LOCK;
        t = ctx->tune; timing = ctx->timing;
    }
    std::lock_guard<std::mutex> lk(h->mu);
    h->tune = t;
    h->timing = timing;
    return h;
}
static void release_workspaces(zk_ctx* ctx) {                          // ctx->mu held
    zk::DevBuf* bufs[] = {&ctx->ws_scalars, &ctx->ws_sorted, &ctx->ws_mid, &ctx->ws_small, &ctx->ws_sub0, &ctx->ws_sub1, &ctx->ws_cls0,
                          &ctx->ws_cls1, &ctx->ws_tmp, &ctx->ws_ntt, &ctx->ws_ntt_in, &ctx->ws_pts, &ctx->ws_runs, &ctx->ws_quot, &ctx->ws_quot_state};
    for (auto* b : bufs) b->release();
}
void zk_internal_trim_helper(zk_ctx* ctx) {
    if (!ctx) return;
    zk_ctx* h;
    { LOCK; h = ctx->helper; }
    if (!h) return;
    std::lock_guard<std::mutex> lk(h->mu);                             // (a side lane still running on it holds this lock call by call: the trim falls between two of its calls, and every entry point re-grows what it needs)
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    release_workspaces(h);
    release_twiddles(h);
}
extern "C" {
void zk_ctx_destroy(zk_ctx* ctx) ZK_ABI_TRY {
    if (!ctx) return;
    zk_ctx* helper;
    { LOCK; helper = ctx->helper; ctx->helper = nullptr; }
    if (helper) zk_ctx_destroy(helper);
    zk_internal_plonk_ctx_destroyed(ctx);       // proving keys built on / shared to this context (prover.hip)
    (void)zk_plonk_trim(ctx);                   // device buffers zk_plonk_create_proof kept for reuse on this context (a later context at the same address must not inherit them)
    {
        LOCK;
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        ctx->bases.clear();                     // shared tables are freed with their last holder
        drop_pending_timers(ctx);
        release_twiddles(ctx);
        release_pks(ctx);
        release_programs(ctx);
        release_gtab(ctx);
        release_workspaces(ctx);
        (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
} ZK_ABI_CATCH_VOID(ctx)

const char* zk_last_error(zk_ctx* ctx) ZK_ABI_TRY { return ctx ? ctx->err : "null context"; } ZK_ABI_CATCH_VALUE(ctx, "zkmi355: internal error")

static int* tune_slot(zk_ctx* ctx, const char* key) {
    zk::Tune& t = ctx->tune;
    struct { const char* k; int* v; } tab[] = {
        {"msm_c", &t.msm_c}, {"msm_sort_wgs", &t.msm_sort_wgs}, {"msm_sort_threads", &t.msm_sort_threads}, {"msm_sort_batch_wgs", &t.msm_sort_batch_wgs}, {"msm_bsort_threads", &t.msm_bsort_threads}, {"msm_bsort_chunk", &t.msm_bsort_chunk}, {"msm_wide_bins_log", &t.msm_wide_bins_log}, {"msm_part_threads", &t.msm_part_threads}, {"msm_part_pairs", &t.msm_part_pairs}, {"msm_two_level_sort", &t.msm_two_level_sort},
        {"msm_target_threads", &t.msm_target_threads}, {"msm_min_chunk", &t.msm_min_chunk}, {"msm_max_chunk", &t.msm_max_chunk},
        {"msm_max_chunk_wide", &t.msm_max_chunk_wide}, {"msm_merge_fanin", &t.msm_merge_fanin}, {"msm_tree_fanin", &t.msm_tree_fanin}, {"msm_block", &t.msm_block}, {"msm_runs", &t.msm_runs}, {"prover_side_lane", &t.prover_side_lane}, {"prover_lane_priority", &t.prover_lane_priority},
        {"ntt_tile_log", &t.ntt_tile_log}, {"ntt_threads", &t.ntt_threads}, {"ntt_max_radix_log", &t.ntt_max_radix_log}, {"ntt_plan", &t.ntt_plan}, {"ntt_full_twiddle_max_log", &t.ntt_full_twiddle_max_log}, {"ntt_coset_table", &t.ntt_coset_table}, {"ntt_col_major", &t.ntt_col_major},
        {"vec_block", &t.vec_block}, {"quot_threads", &t.quot_threads}, {"lookup_force_generic_sort", &t.lookup_force_generic_sort},
        {"ntt_quarter_input", &t.ntt_quarter_input}, {"ntt_fuse_scale", &t.ntt_fuse_scale}, {"quot_piece_cosets", &t.quot_piece_cosets}, {"quot_factor_horner", &t.quot_factor_horner}, {"quot_degree_split", &t.quot_degree_split}, {"quot_group_factors", &t.quot_group_factors}, {"ntt_ws_limit_mb", &t.ntt_ws_limit_mb}, 
        {"quot_jit", &t.quot_jit}, {"quot_jit_group", &t.quot_jit_group}, {"quot_jit_waves", &t.quot_jit_waves}, {"quot_remat_ops", &t.quot_remat_ops}, {"quot_remat_distance", &t.quot_remat_distance}};
    for (auto& e : tab) if (!strcmp(e.k, key)) return e.v;
    return nullptr;
}
int zk_tune_set(zk_ctx* ctx, const char* key, int value) ZK_ABI_TRY {
    NEED_CTX; LOCK;
    int* s = key ? tune_slot(ctx, key) : nullptr;
    if (!s) return ctx->fail(ZK_ERR_ARG, "zk_tune_set: unknown key '%s'", key ? key : "(null)");
    if (value < 0) return ctx->fail(ZK_ERR_ARG, "zk_tune_set: negative value");
    *s = value;
    return ZK_OK;
} ZK_ABI_CATCH(ctx)
int zk_tune_get(zk_ctx* ctx, const char* key, int* value) ZK_ABI_TRY {
    NEED_CTX; LOCK;
    int* s = key ? tune_slot(ctx, key) : nullptr;
    if (!s || !value) return ctx->fail(ZK_ERR_ARG, "zk_tune_get: unknown key '%s'", key ? key : "(null)");
    *value = *s;
    return ZK_OK;
} ZK_ABI_CATCH(ctx)
int zk_timing_enable(zk_ctx* ctx, int on) ZK_ABI_TRY {
    NEED_CTX;
    zk_ctx* helper;
    { LOCK; helper = ctx->helper; }
    if (helper) (void)zk_timing_enable(helper, on);
    LOCK;
    resolve_pending_timers(ctx);                                     // (also when the timing goes OFF: the event pairs are read — zk_timing_get finds the sums afterwards — and destroyed)
    if (on) ctx->last_ms.clear();
    ctx->timing = on != 0;
    return ZK_OK;
} ZK_ABI_CATCH(ctx)
double zk_timing_get(zk_ctx* ctx, const char* label) ZK_ABI_TRY {
    if (!ctx || !label) return -1.0;
    zk_ctx* helper;
    { LOCK; helper = ctx->helper; }
    const double side = helper ? zk_timing_get(helper, label) : -1.0;      // what the helper context ran for this one counts as this one's
    LOCK;
    resolve_pending_timers(ctx);
    auto it = ctx->last_ms.find(label);
    if (it == ctx->last_ms.end()) return side;
    return it->second + (side > 0 ? side : 0.0);
} ZK_ABI_CATCH_VALUE(ctx, -1.0)

#define ENTER NEED_CTX; LOCK; ZK_HIP(hipSetDevice(ctx->device))

int zk_dev_alloc(zk_ctx* ctx, size_t bytes, void** dptr) ZK_ABI_TRY { ENTER; if (!dptr) return ctx->fail(ZK_ERR_ARG, "null"); ZK_HIP(hipMalloc(dptr, bytes ? bytes : 32)); return ZK_OK; } ZK_ABI_CATCH(ctx)
int zk_dev_free(zk_ctx* ctx, void* dptr) ZK_ABI_TRY { ENTER; ZK_HIP(hipFree(dptr)); return ZK_OK; } ZK_ABI_CATCH(ctx)
int zk_dev_upload(zk_ctx* ctx, void* dptr, const void* host, size_t bytes) ZK_ABI_TRY {
    ENTER; if ((!dptr || !host) && bytes) return ctx->fail(ZK_ERR_ARG, "zk_dev_upload: null");
    ZK_HIP(hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream)); ZK_HIP(hipStreamSynchronize(ctx->stream)); return ZK_OK;
} ZK_ABI_CATCH(ctx)
int zk_dev_download(zk_ctx* ctx, void* host, const void* dptr, size_t bytes) ZK_ABI_TRY {
    ENTER; if ((!dptr || !host) && bytes) return ctx->fail(ZK_ERR_ARG, "zk_dev_download: null");
    ZK_HIP(hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream)); ZK_HIP(hipStreamSynchronize(ctx->stream)); return ZK_OK;
} ZK_ABI_CATCH(ctx)
int zk_dev_copy(zk_ctx* ctx, void* dst, const void* src, size_t bytes) ZK_ABI_TRY {
    ENTER; if ((!dst || !src) && bytes) return ctx->fail(ZK_ERR_ARG, "zk_dev_copy: null");
    ZK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream)); ZK_HIP(hipStreamSynchronize(ctx->stream)); return ZK_OK;
} ZK_ABI_CATCH(ctx)
int zk_dev_zero(zk_ctx* ctx, void* dptr, size_t bytes) ZK_ABI_TRY {
    ENTER; if (!dptr && bytes) return ctx->fail(ZK_ERR_ARG, "zk_dev_zero: null");
    ZK_HIP(hipMemsetAsync(dptr, 0, bytes, ctx->stream)); return ZK_OK;          // stream-ordered like every kernel of the context
} ZK_ABI_CATCH(ctx)
int zk_dev_sync(zk_ctx* ctx) ZK_ABI_TRY { ENTER; ZK_HIP(hipStreamSynchronize(ctx->stream)); return ZK_OK; } ZK_ABI_CATCH(ctx)
// page-locked staging memory for the columns the caller hands over every proof (the witness): DMA reads it at link rate, no bounce buffer
int zk_host_alloc(zk_ctx* ctx, size_t bytes, void** hptr) ZK_ABI_TRY {
    ENTER; if (!hptr) return ctx->fail(ZK_ERR_ARG, "zk_host_alloc: null");
    ZK_HIP(hipHostMalloc(hptr, bytes ? bytes : 32, 0)); return ZK_OK;
} ZK_ABI_CATCH(ctx)
int zk_host_free(zk_ctx* ctx, void* hptr) ZK_ABI_TRY { ENTER; ZK_HIP(hipHostFree(hptr)); return ZK_OK; } ZK_ABI_CATCH(ctx)
int zk_dev_upload_batch(zk_ctx* ctx, void* const* dptrs, const void* const* hosts, size_t count, size_t bytes_each) ZK_ABI_TRY {
    ENTER; if ((!dptrs || !hosts) && count) return ctx->fail(ZK_ERR_ARG, "zk_dev_upload_batch: null");
    for (size_t i = 0; i < count; i++) {
        if (!dptrs[i] || !hosts[i]) return ctx->fail(ZK_ERR_ARG, "zk_dev_upload_batch: null column %zu", i);
        ZK_HIP(hipMemcpyAsync(dptrs[i], hosts[i], bytes_each, hipMemcpyHostToDevice, ctx->stream));
    }
    ZK_HIP(hipStreamSynchronize(ctx->stream)); return ZK_OK;
} ZK_ABI_CATCH(ctx)

// ---- MSM ----------------------------------------------------------------------------------------
int zk_bases_register(zk_ctx* ctx, const void* p, size_t n, uint64_t* h) ZK_ABI_TRY { ENTER; return msm_register(ctx, p, n, false, h); } ZK_ABI_CATCH(ctx)
int zk_bases_register_dev(zk_ctx* ctx, const void* p, size_t n, uint64_t* h) ZK_ABI_TRY { ENTER; return msm_register(ctx, p, n, true, h); } ZK_ABI_CATCH(ctx)
int zk_bases_release(zk_ctx* ctx, uint64_t h) ZK_ABI_TRY { ENTER; return msm_release(ctx, h); } ZK_ABI_CATCH(ctx)
int zk_bases_enable_runs(zk_ctx* ctx, uint64_t h) ZK_ABI_TRY { ENTER; return msm_enable_runs(ctx, h); } ZK_ABI_CATCH(ctx)
int zk_bases_share(zk_ctx* ctx, zk_ctx* owner, uint64_t owner_handle, uint64_t* h) ZK_ABI_TRY {
    if (!ctx || !owner || !h) return ZK_ERR_ARG;
    zk::BaseTable bt;
    {
        std::lock_guard<std::mutex> lk(owner->mu);
        auto it = owner->bases.find(owner_handle);
        if (it == owner->bases.end()) return ZK_ERR_ARG;
        bt = it->second;
    }
    LOCK;
    if (owner->device != ctx->device) return ctx->fail(ZK_ERR_ARG, "zk_bases_share: the contexts are on different devices (%d, %d)", owner->device, ctx->device);
    return msm_share(ctx, bt, h);
} ZK_ABI_CATCH(ctx)
int zk_msm(zk_ctx* ctx, uint64_t b, const void* s, size_t n, void* out) ZK_ABI_TRY { ENTER; return msm_run(ctx, b, s, n, false, out, 0); } ZK_ABI_CATCH(ctx)
int zk_msm_dev(zk_ctx* ctx, uint64_t b, const void* s, size_t n, void* out) ZK_ABI_TRY { ENTER; return msm_run(ctx, b, s, n, true, out, 0); } ZK_ABI_CATCH(ctx)
int zk_msm_batch(zk_ctx* ctx, uint64_t b, const void* const* s, size_t count, size_t n, void* out) ZK_ABI_TRY { ENTER; return msm_run_batch(ctx, b, s, count, n, false, out, 0); } ZK_ABI_CATCH(ctx)
int zk_msm_batch_dev(zk_ctx* ctx, uint64_t b, const void* const* s, size_t count, size_t n, void* out) ZK_ABI_TRY { ENTER; return msm_run_batch(ctx, b, s, count, n, true, out, 0); } ZK_ABI_CATCH(ctx)
int zk_msm_partial_dev(zk_ctx* ctx, uint64_t b, const void* s, size_t n, void* out) ZK_ABI_TRY { ENTER; return msm_run(ctx, b, s, n, true, out, 1); } ZK_ABI_CATCH(ctx)
int zk_msm_batch_partial_dev(zk_ctx* ctx, uint64_t b, const void* const* s, size_t count, size_t n, void* out) ZK_ABI_TRY { ENTER; return msm_run_batch(ctx, b, s, count, n, true, out, 1); } ZK_ABI_CATCH(ctx)
int zk_g1_sum_xyzz_batch(const void* xyzz, size_t parts, size_t count, void* out) ZK_ABI_TRY {
    if (!xyzz || !out) return ZK_ERR_ARG;
    for (size_t c = 0; c < count; c++) {   // column c of every part: xyzz[(p * count + c)]
        std::vector<unsigned char> col(parts * 128);
        for (size_t p = 0; p < parts; p++) memcpy(&col[p * 128], (const char*)xyzz + (p * count + c) * 128, 128);
        int rc = g1_sum_xyzz_host(col.data(), parts, (char*)out + c * 96);
        if (rc) return rc;
    }
    return ZK_OK;
} ZK_ABI_CATCH(nullptr)
int zk_g1_sum_xyzz(const void* xyzz, size_t count, void* out) ZK_ABI_TRY { if (!xyzz || !out) return ZK_ERR_ARG; return g1_sum_xyzz_host(xyzz, count, out); } ZK_ABI_CATCH(nullptr)
int zk_g1_fixed_base_mul_dev(zk_ctx* ctx, const void* s, size_t n, void* out) ZK_ABI_TRY { ENTER; return g1_fixed_base_mul(ctx, s, n, out); } ZK_ABI_CATCH(ctx)

int zk_g1_ntt_dev(zk_ctx* ctx, const void* in, uint32_t log_n, const void* omega, const void* scale, void* out) ZK_ABI_TRY { ENTER; return g1_ntt(ctx, in, log_n, omega, scale, out); } ZK_ABI_CATCH(ctx)

int zk_g1_decompress_dev(zk_ctx* ctx, const void* bytes_dev, size_t n, uint32_t sign_bit, void* out_affine_dev, uint32_t* n_invalid) ZK_ABI_TRY { ENTER; return g1_decompress(ctx, bytes_dev, n, sign_bit, out_affine_dev, n_invalid); } ZK_ABI_CATCH(ctx)
int zk_g1_compress_dev(zk_ctx* ctx, const void* affine_dev, size_t n, uint32_t sign_bit, void* bytes_dev) ZK_ABI_TRY { ENTER; return g1_compress(ctx, affine_dev, n, sign_bit, bytes_dev); } ZK_ABI_CATCH(ctx)

// ---- NTT / domain -------------------------------------------------------------------------------
static int with_host_buffer(zk_ctx* ctx, void* host_in_out, size_t in_bytes, size_t buf_bytes, size_t out_bytes, void** dbuf) {
    ZK_HIP(ctx->ws_ntt_in.ensure(buf_bytes));
    *dbuf = ctx->ws_ntt_in.p;
    (void)out_bytes;
    ZK_HIP(hipMemcpyAsync(*dbuf, host_in_out, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    return ZK_OK;
}
static int finish_host(zk_ctx* ctx, void* host, const void* dbuf, size_t bytes) {
    ZK_HIP(hipMemcpyAsync(host, dbuf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}
static u256 load_host_fr(const void* p) { u256 o; memcpy(&o, p, 32); return o; }

int zk_ntt_dev(zk_ctx* ctx, void* a, uint32_t log_n, const void* omega) ZK_ABI_TRY {
    ENTER; if (!omega) return ctx->fail(ZK_ERR_ARG, "zk_ntt: null omega");
    int rc = ntt_dev(ctx, a, log_n, load_host_fr(omega), nullptr); if (rc) return rc;
    ZK_HIP(hipStreamSynchronize(ctx->stream)); return ZK_OK;
} ZK_ABI_CATCH(ctx)
int zk_ntt(zk_ctx* ctx, void* a, uint32_t log_n, const void* omega) ZK_ABI_TRY {
    ENTER; if (!omega || !a) return ctx->fail(ZK_ERR_ARG, "zk_ntt: null pointer");
    if (log_n > 27) return ctx->fail(ZK_ERR_LIMIT, "zk_ntt: log_n = %u > 27", log_n);
    void* d; size_t bytes = (size_t)32 << log_n;
    int rc = with_host_buffer(ctx, a, bytes, bytes, bytes, &d); if (rc) return rc;
    rc = ntt_dev(ctx, d, log_n, load_host_fr(omega), nullptr); if (rc) return rc;
    return finish_host(ctx, a, d, bytes);
} ZK_ABI_CATCH(ctx)
int zk_ntt_batch_dev(zk_ctx* ctx, void* const* cols, size_t count, uint32_t log_n, const void* omega) ZK_ABI_TRY {
    ENTER; if (!omega) return ctx->fail(ZK_ERR_ARG, "zk_ntt_batch_dev: null omega");
    int rc = ntt_dev_batch(ctx, cols, nullptr, count, log_n, load_host_fr(omega), nullptr); if (rc) return rc;
    ZK_HIP(hipStreamSynchronize(ctx->stream)); return ZK_OK;
} ZK_ABI_CATCH(ctx)
int zk_lagrange_to_coeff_batch_dev(zk_ctx* ctx, void* const* cols, size_t count, uint32_t k) ZK_ABI_TRY {
    ENTER; int rc = domain_lagrange_to_coeff_batch(ctx, cols, count, k); if (rc) return rc; ZK_HIP(hipStreamSynchronize(ctx->stream)); return ZK_OK;
} ZK_ABI_CATCH(ctx)
int zk_coeff_to_extended_batch_dev(zk_ctx* ctx, const void* const* coeffs, void* const* outs, size_t count, uint32_t k, uint32_t ek) ZK_ABI_TRY {
    ENTER; int rc = domain_coeff_to_extended_batch(ctx, coeffs, outs, count, k, ek); if (rc) return rc; ZK_HIP(hipStreamSynchronize(ctx->stream)); return ZK_OK;
} ZK_ABI_CATCH(ctx)
int zk_coeff_to_coset_batch_dev(zk_ctx* ctx, const void* const* coeffs, void* const* outs, size_t count, uint32_t k, uint32_t ek, uint32_t coset) ZK_ABI_TRY {
    ENTER; int rc = domain_coeff_to_coset_batch(ctx, coeffs, outs, count, k, ek, coset); if (rc) return rc; ZK_HIP(hipStreamSynchronize(ctx->stream)); return ZK_OK;
} ZK_ABI_CATCH(ctx)
int zk_fr_interleave_dev(zk_ctx* ctx, const void* const* cosets, size_t count, size_t n, void* out) ZK_ABI_TRY { ENTER; return fr_interleave(ctx, cosets, count, n, out); } ZK_ABI_CATCH(ctx)
int zk_cosets_to_pieces_dev(zk_ctx* ctx, void* const* numer, uint32_t pieces, uint32_t k, uint32_t ek, void* const* out) ZK_ABI_TRY { ENTER; return domain_cosets_to_pieces(ctx, numer, pieces, k, ek, out); } ZK_ABI_CATCH(ctx)
int zk_lagrange_to_coeff_dev(zk_ctx* ctx, void* a, uint32_t k) ZK_ABI_TRY { ENTER; int rc = domain_lagrange_to_coeff(ctx, a, k); if (rc) return rc; ZK_HIP(hipStreamSynchronize(ctx->stream)); return ZK_OK; } ZK_ABI_CATCH(ctx)
int zk_coeff_to_lagrange_dev(zk_ctx* ctx, void* a, uint32_t k) ZK_ABI_TRY { ENTER; int rc = domain_coeff_to_lagrange(ctx, a, k); if (rc) return rc; ZK_HIP(hipStreamSynchronize(ctx->stream)); return ZK_OK; } ZK_ABI_CATCH(ctx)
int zk_coeff_to_extended_dev(zk_ctx* ctx, const void* c, uint32_t k, uint32_t ek, void* out) ZK_ABI_TRY { ENTER; int rc = domain_coeff_to_extended(ctx, c, k, ek, out); if (rc) return rc; ZK_HIP(hipStreamSynchronize(ctx->stream)); return ZK_OK; } ZK_ABI_CATCH(ctx)
int zk_extended_to_coeff_dev(zk_ctx* ctx, void* a, uint32_t k, uint32_t ek) ZK_ABI_TRY { ENTER; int rc = domain_extended_to_coeff(ctx, a, k, ek); if (rc) return rc; ZK_HIP(hipStreamSynchronize(ctx->stream)); return ZK_OK; } ZK_ABI_CATCH(ctx)
int zk_divide_by_vanishing_poly_dev(zk_ctx* ctx, void* a, uint32_t k, uint32_t ek) ZK_ABI_TRY { ENTER; return domain_divide_by_vanishing(ctx, a, k, ek); } ZK_ABI_CATCH(ctx)

int zk_lagrange_to_coeff(zk_ctx* ctx, void* a, uint32_t k) ZK_ABI_TRY {
    ENTER; if (!a || k > 27) return ctx->fail(ZK_ERR_ARG, "zk_lagrange_to_coeff: bad argument");
    void* d; size_t bytes = (size_t)32 << k;
    int rc = with_host_buffer(ctx, a, bytes, bytes, bytes, &d); if (rc) return rc;
    rc = domain_lagrange_to_coeff(ctx, d, k); if (rc) return rc;
    return finish_host(ctx, a, d, bytes);
} ZK_ABI_CATCH(ctx)
int zk_coeff_to_extended(zk_ctx* ctx, const void* c, uint32_t k, uint32_t ek, void* out) ZK_ABI_TRY {
    ENTER; if (!c || !out || k > ek || ek > 27) return ctx->fail(ZK_ERR_ARG, "zk_coeff_to_extended: bad argument");
    size_t inb = (size_t)32 << k, outb = (size_t)32 << ek;
    ZK_HIP(ctx->ws_ntt_in.ensure(inb + outb));
    void* din = ctx->ws_ntt_in.p; void* dout = (char*)din + inb;
    ZK_HIP(hipMemcpyAsync(din, c, inb, hipMemcpyHostToDevice, ctx->stream));
    int rc = domain_coeff_to_extended(ctx, din, k, ek, dout); if (rc) return rc;
    return finish_host(ctx, out, dout, outb);
} ZK_ABI_CATCH(ctx)
int zk_extended_to_coeff(zk_ctx* ctx, void* a, uint32_t k, uint32_t ek) ZK_ABI_TRY {
    ENTER; if (!a || k > ek || ek > 27) return ctx->fail(ZK_ERR_ARG, "zk_extended_to_coeff: bad argument");
    void* d; size_t bytes = (size_t)32 << ek;
    int rc = with_host_buffer(ctx, a, bytes, bytes, bytes, &d); if (rc) return rc;
    rc = domain_extended_to_coeff(ctx, d, k, ek); if (rc) return rc;
    return finish_host(ctx, a, d, bytes);
} ZK_ABI_CATCH(ctx)

// ---- vectors ------------------------------------------------------------------------------------
static int vec_sync(zk_ctx* ctx, int rc) { if (rc) return rc; ZK_HIP(hipStreamSynchronize(ctx->stream)); return ZK_OK; }
int zk_fr_mul_dev(zk_ctx* ctx, const void* a, const void* b, void* o, size_t n) ZK_ABI_TRY { ENTER; return vec_sync(ctx, fr_vec_op(ctx, 0, a, b, o, n, nullptr)); } ZK_ABI_CATCH(ctx)
int zk_fr_add_dev(zk_ctx* ctx, const void* a, const void* b, void* o, size_t n) ZK_ABI_TRY { ENTER; return vec_sync(ctx, fr_vec_op(ctx, 1, a, b, o, n, nullptr)); } ZK_ABI_CATCH(ctx)
int zk_fr_sub_dev(zk_ctx* ctx, const void* a, const void* b, void* o, size_t n) ZK_ABI_TRY { ENTER; return vec_sync(ctx, fr_vec_op(ctx, 2, a, b, o, n, nullptr)); } ZK_ABI_CATCH(ctx)
int zk_fr_scale_dev(zk_ctx* ctx, const void* a, const void* s, void* o, size_t n) ZK_ABI_TRY {
    ENTER; if (!s) return ctx->fail(ZK_ERR_ARG, "zk_fr_scale_dev: null scalar");
    u256 sc = load_host_fr(s); return vec_sync(ctx, fr_vec_op(ctx, 3, a, nullptr, o, n, &sc));
} ZK_ABI_CATCH(ctx)
int zk_fq_mul_dev(zk_ctx* ctx, const void* a, const void* b, void* o, size_t n) ZK_ABI_TRY { ENTER; return vec_sync(ctx, fr_vec_op(ctx, 4, a, b, o, n, nullptr)); } ZK_ABI_CATCH(ctx)

// ---- grand products -----------------------------------------------------------------------------
int zk_permutation_product_dev(zk_ctx* ctx, const void* const* values, const void* const* sigmas, size_t count, uint32_t k, const void* beta,
                               const void* gamma, const void* delta_start, const void* z_init, const void* blinding, uint32_t blinding_factors,
                               void* z_dev, void* last_z_out) ZK_ABI_TRY {
    ENTER; return permutation_product(ctx, values, sigmas, count, k, beta, gamma, delta_start, z_init, blinding, blinding_factors, z_dev, last_z_out);
} ZK_ABI_CATCH(ctx)
int zk_lookup_product_dev(zk_ctx* ctx, const void* cin, const void* ctab, const void* pin, const void* ptab, uint32_t k, const void* beta,
                          const void* gamma, const void* blinding, uint32_t blinding_factors, void* z_dev) ZK_ABI_TRY {
    ENTER; return lookup_product(ctx, cin, ctab, pin, ptab, k, beta, gamma, blinding, blinding_factors, z_dev);
} ZK_ABI_CATCH(ctx)

int zk_permutation_product_all_dev(zk_ctx* ctx, const void* const* values, const void* const* sigmas, size_t n_columns, uint32_t chunk_len, uint32_t k,
                                   const void* beta, const void* gamma, const void* blinding, uint32_t blinding_factors, void* const* z_devs) ZK_ABI_TRY {
    ENTER; return permutation_product_all(ctx, values, sigmas, n_columns, chunk_len, k, beta, gamma, blinding, blinding_factors, z_devs);
} ZK_ABI_CATCH(ctx)
int zk_lookup_product_batch_dev(zk_ctx* ctx, const void* const* cols4, size_t count, uint32_t k, const void* beta, const void* gamma, const void* blinding,
                                uint32_t blinding_factors, void* const* z_devs) ZK_ABI_TRY {
    ENTER; return lookup_product_batch(ctx, cols4, count, k, beta, gamma, blinding, blinding_factors, z_devs);
} ZK_ABI_CATCH(ctx)

int zk_lookup_permute_dev(zk_ctx* ctx, const void* input, const void* table, uint32_t k, uint32_t blinding_factors, const void* blind_input,
                          const void* blind_table, void* out_input, void* out_table) ZK_ABI_TRY {
    ENTER; return lookup_permute(ctx, input, table, k, blinding_factors, blind_input, blind_table, out_input, out_table);
} ZK_ABI_CATCH(ctx)

int zk_lookup_permute_batch_dev(zk_ctx* ctx, const void* const* inputs, const void* const* tables, size_t count, uint32_t k, uint32_t blinding_factors,
                                const void* blind_inputs, const void* blind_tables, void* const* out_inputs, void* const* out_tables) ZK_ABI_TRY {
    ENTER; return lookup_permute_batch(ctx, inputs, tables, count, k, blinding_factors, blind_inputs, blind_tables, out_inputs, out_tables);
} ZK_ABI_CATCH(ctx)

// ---- evaluation phase ---------------------------------------------------------------------------
int zk_eval_polynomial_batch_dev(zk_ctx* ctx, const void* const* polys, size_t count, size_t n, const void* points, void* out) ZK_ABI_TRY {
    ENTER; return eval_polynomial_batch(ctx, polys, count, n, points, out);
} ZK_ABI_CATCH(ctx)
int zk_kate_division_dev(zk_ctx* ctx, const void* a_dev, size_t n, const void* b, void* q_dev) ZK_ABI_TRY { ENTER; return kate_division(ctx, a_dev, n, b, q_dev); } ZK_ABI_CATCH(ctx)
int zk_fr_lincomb_dev(zk_ctx* ctx, const void* const* polys_dev, const void* scalars, size_t count, size_t n, void* out_dev) ZK_ABI_TRY { ENTER; return fr_lincomb(ctx, polys_dev, scalars, count, n, out_dev); } ZK_ABI_CATCH(ctx)

// ---- quotient -----------------------------------------------------------------------------------
int zk_quotient_program_load(zk_ctx* ctx, const void* blob, size_t len, uint64_t* prog) ZK_ABI_TRY { ENTER; return quotient_program_load(ctx, blob, len, prog); } ZK_ABI_CATCH(ctx)
int zk_quotient_program_info(zk_ctx* ctx, uint64_t prog, uint32_t* ni, uint32_t* ns, uint32_t* nc) ZK_ABI_TRY { ENTER; return quotient_program_info(ctx, prog, ni, ns, nc); } ZK_ABI_CATCH(ctx)
int zk_quotient_program_kernels(zk_ctx* ctx, uint64_t prog, uint32_t* n_kernels) ZK_ABI_TRY { ENTER; return quotient_program_kernels(ctx, prog, n_kernels); } ZK_ABI_CATCH(ctx)
int zk_quotient_program_opmix(zk_ctx* ctx, uint64_t prog, uint32_t counts[9]) ZK_ABI_TRY { ENTER; return quotient_program_opmix(ctx, prog, 0, counts); } ZK_ABI_CATCH(ctx)
int zk_quotient_program_part_opmix(zk_ctx* ctx, uint64_t prog, uint32_t part, uint32_t counts[9]) ZK_ABI_TRY { ENTER; return quotient_program_opmix(ctx, prog, part, counts); } ZK_ABI_CATCH(ctx)
int zk_quotient_program_release(zk_ctx* ctx, uint64_t prog) ZK_ABI_TRY { ENTER; return quotient_program_release(ctx, prog); } ZK_ABI_CATCH(ctx)
int zk_quotient_program_share(zk_ctx* ctx, zk_ctx* owner, uint64_t owner_prog, uint64_t* prog) ZK_ABI_TRY {
    if (!ctx || !owner || !prog) return ZK_ERR_ARG;
    return quotient_program_share(ctx, owner, owner_prog, prog);
} ZK_ABI_CATCH(ctx)
// ABI versioning (zkmi355.h): the caller's sizeof of a boundary struct must be this build's before any other field is read
#define ARGS_SIZE(fn) do { if (!args) return ctx->fail(ZK_ERR_ARG, fn ": null args"); \
        if (args->struct_size != sizeof(zk_quotient_args)) return ctx->fail(ZK_ERR_ARG, fn ": zk_quotient_args.struct_size %u, expected %zu (ABI version %u)", args->struct_size, sizeof(zk_quotient_args), ZK_ABI_VERSION); } while (0)
int zk_quotient_run_dev(zk_ctx* ctx, uint64_t prog, const zk_quotient_args* args) ZK_ABI_TRY { ENTER; ARGS_SIZE("zk_quotient_run_dev"); return quotient_run(ctx, prog, args, -1, 0, 0, 0, 0); } ZK_ABI_CATCH(ctx)
int zk_quotient_run_coset_dev(zk_ctx* ctx, uint64_t prog, const zk_quotient_args* args, uint32_t coset) ZK_ABI_TRY {
    ENTER;
    ARGS_SIZE("zk_quotient_run_coset_dev");
    if (coset >= (1u << 16)) return ctx->fail(ZK_ERR_ARG, "zk_quotient_run_coset_dev: coset %u out of range", coset);
    return quotient_run(ctx, prog, args, (int)coset, 0, 0, 0, 0);
} ZK_ABI_CATCH(ctx)
int zk_quotient_run_coset_rows_dev(zk_ctx* ctx, uint64_t prog, const zk_quotient_args* args, uint32_t coset, uint64_t row_lo, uint64_t row_count) ZK_ABI_TRY {
    ENTER;
    ARGS_SIZE("zk_quotient_run_coset_rows_dev");
    if (coset >= (1u << 16)) return ctx->fail(ZK_ERR_ARG, "zk_quotient_run_coset_rows_dev: coset %u out of range", coset);
    if (!row_count) return ctx->fail(ZK_ERR_ARG, "zk_quotient_run_coset_rows_dev: row_count = 0");
    return quotient_run(ctx, prog, args, (int)coset, row_lo, row_count, 0, 0);
} ZK_ABI_CATCH(ctx)
int zk_quotient_program_split(zk_ctx* ctx, uint64_t prog, uint32_t* low_cosets, uint32_t* n_instr_high, uint32_t* n_instr_low) ZK_ABI_TRY { ENTER; return quotient_program_split(ctx, prog, low_cosets, n_instr_high, n_instr_low); } ZK_ABI_CATCH(ctx)
int zk_quotient_run_high_dev(zk_ctx* ctx, uint64_t prog, const zk_quotient_args* args) ZK_ABI_TRY { ENTER; ARGS_SIZE("zk_quotient_run_high_dev"); return quotient_run(ctx, prog, args, -1, 0, 0, 1, 0); } ZK_ABI_CATCH(ctx)
int zk_quotient_run_low_dev(zk_ctx* ctx, uint64_t prog, const zk_quotient_args* args, uint32_t low_cosets) ZK_ABI_TRY {
    ENTER; ARGS_SIZE("zk_quotient_run_low_dev");
    if (!low_cosets) return ctx->fail(ZK_ERR_ARG, "zk_quotient_run_low_dev: low_cosets = 0");
    return quotient_run(ctx, prog, args, -1, 0, 0, 2, low_cosets);
} ZK_ABI_CATCH(ctx)
int zk_quotient_run_coset_part_dev(zk_ctx* ctx, uint64_t prog, const zk_quotient_args* args, uint32_t coset, uint32_t part) ZK_ABI_TRY {
    ENTER; ARGS_SIZE("zk_quotient_run_coset_part_dev");
    if (coset >= (1u << 16) || part < 1 || part > 2) return ctx->fail(ZK_ERR_ARG, "zk_quotient_run_coset_part_dev: coset %u / part %u out of range", coset, part);
    return quotient_run(ctx, prog, args, (int)coset, 0, 0, (int)part, 0);
} ZK_ABI_CATCH(ctx)

int zk_pk_load(zk_ctx* ctx, uint64_t prog, const void* const* fixed, const void* const* sigma, const void* l0, const void* l_last, const void* l_active,
               int form, uint64_t* pk) ZK_ABI_TRY { ENTER; return pk_load(ctx, prog, fixed, sigma, l0, l_last, l_active, form, pk); } ZK_ABI_CATCH(ctx)
int zk_pk_release(zk_ctx* ctx, uint64_t pk) ZK_ABI_TRY { ENTER; return pk_release(ctx, pk); } ZK_ABI_CATCH(ctx)
int zk_evaluate_h(zk_ctx* ctx, uint64_t pk, const void* const* advice, const void* const* instance, const void* const* perm_products,
                  const void* const* lookup_product, const void* const* lookup_input, const void* const* lookup_table, const void* challenges,
                  const void* beta, const void* gamma, const void* theta, const void* y, int finish, void* out) ZK_ABI_TRY {
    ENTER; return evaluate_h_host(ctx, pk, advice, instance, perm_products, lookup_product, lookup_input, lookup_table, challenges, beta, gamma, theta, y, finish, out);
} ZK_ABI_CATCH(ctx)

}  // extern "C"
