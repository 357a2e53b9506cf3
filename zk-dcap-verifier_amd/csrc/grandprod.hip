// Grand-product columns of the permutation and lookup arguments — SURVEY.md §8(f) "next 1".
//
// Replaces the row loops of halo2_proofs (zkwebauthn @ c254c75, Cargo.lock:1314-1327)
//   src/plonk/permutation/prover.rs  Argument::commit      (one call per column set / chunk)
//   src/plonk/lookup/prover.rs       Permuted::commit_product
// i.e. frac[i] = numerator_i / denominator_i (batch inversion), z[0] = z_init,
// z[i+1] = z[i] * frac[i], last `blinding_factors` rows replaced by caller-supplied randomness.
// The CPU code is a serial scan over n rows per column set; here it is a three-phase parallel prefix
// product (per-thread runs -> LDS scan per workgroup -> scan of workgroup totals), and the batch
// inversion is Montgomery's trick on per-thread chunks.  Outputs feed zk_msm / zk_lagrange_to_coeff
// directly, so the column never leaves HBM.
#include "ctx.h"
#include <algorithm>
#include <vector>

namespace zk {

int ntt_pow_tables(zk_ctx* ctx, uint32_t log_n, const u256& omega, const void** lo, const void** hi, uint32_t* lo_bits);
u256 domain_omega(uint32_t k);

constexpr int GP_MAX_COLS = 16;
constexpr uint32_t GP_E = 8;       // elements per thread in the scan
constexpr uint32_t GP_T = 256;     // threads per workgroup in the scan

struct GpPermArgs {
    const void* values[GP_MAX_COLS];
    const void* sigmas[GP_MAX_COLS];
    u256 delta_beta[GP_MAX_COLS];   // delta^j * beta for the j-th column of the set (global delta power included)
    uint32_t count;
    uint32_t n;
    u256 beta, gamma;
    const void* tw_lo; const void* tw_hi; uint32_t lo_bits;   // omega^i tables
    void* num; void* den;
};

// numerator and denominator of row i for one column set
ZK_KERNEL void gp_perm_fraction_kernel(GpPermArgs a) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    u256 w = load_u256(a.tw_lo, (uint32_t)i & ((1u << a.lo_bits) - 1u));
    const uint32_t h = (uint32_t)i >> a.lo_bits;
    if (h) w = Fr::mul(w, load_u256(a.tw_hi, h));      // omega^i
    u256 num = Fr::one(), den = Fr::one();
    for (uint32_t j = 0; j < a.count; j++) {
        const u256 v = load_u256(a.values[j], i);
        const u256 vg = Fr::add(v, a.gamma);
        den = Fr::mul(den, Fr::add(Fr::mul(a.beta, load_u256(a.sigmas[j], i)), vg));
        num = Fr::mul(num, Fr::add(Fr::mul(a.delta_beta[j], w), vg));
    }
    store_u256(a.num, i, num);
    store_u256(a.den, i, den);
}
// lookup: num = (compressed_input + beta)(compressed_table + gamma), den = (permuted_input + beta)(permuted_table + gamma)
ZK_KERNEL void gp_lookup_fraction_kernel(const void* cin, const void* ctab, const void* pin, const void* ptab, uint32_t n, u256 beta, u256 gamma,
                                         void* num, void* den) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_u256(num, i, Fr::mul(Fr::add(load_u256(cin, i), beta), Fr::add(load_u256(ctab, i), gamma)));
    store_u256(den, i, Fr::mul(Fr::add(load_u256(pin, i), beta), Fr::add(load_u256(ptab, i), gamma)));
}
// batched: blockIdx.y = lookup; cols = [cin_0, ctab_0, pin_0, ptab_0, cin_1, ...] (device array), num / den = [l][n]
ZK_KERNEL void gp_lookup_fraction_batch_kernel(const void* const* cols, uint32_t n, u256 beta, u256 gamma, void* num, void* den) {
    const uint32_t l = blockIdx.y;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const void* const* c = cols + 4 * (size_t)l;
    store_u256(num, (size_t)l * n + i, Fr::mul(Fr::add(load_u256(c[0], i), beta), Fr::add(load_u256(c[1], i), gamma)));
    store_u256(den, (size_t)l * n + i, Fr::mul(Fr::add(load_u256(c[2], i), beta), Fr::add(load_u256(c[3], i), gamma)));
}
// ---- prefix product ------------------------------------------------------------------------------
template <bool REVERSE = false>            // REVERSE: exclusive SUFFIX products (the scan runs over the threads in descending order)
__device__ __forceinline__ u256 gp_block_exclusive_scan(u256 mine, u256* total_out) {
    __shared__ uint4 slo[GP_T], shi[GP_T];
    const uint32_t tid = REVERSE ? blockDim.x - 1 - threadIdx.x : threadIdx.x;
    u256 incl = mine;
    slo[tid] = make_uint4(incl.v[0], incl.v[1], incl.v[2], incl.v[3]);
    shi[tid] = make_uint4(incl.v[4], incl.v[5], incl.v[6], incl.v[7]);
    __syncthreads();
    for (uint32_t d = 1; d < blockDim.x; d <<= 1) {
        u256 other = Fr::one();
        if (tid >= d) {
            uint4 l = slo[tid - d], h = shi[tid - d];
            other.v[0] = l.x; other.v[1] = l.y; other.v[2] = l.z; other.v[3] = l.w; other.v[4] = h.x; other.v[5] = h.y; other.v[6] = h.z; other.v[7] = h.w;
        }
        __syncthreads();
        if (tid >= d) {
            incl = Fr::mul(incl, other);
            slo[tid] = make_uint4(incl.v[0], incl.v[1], incl.v[2], incl.v[3]);
            shi[tid] = make_uint4(incl.v[4], incl.v[5], incl.v[6], incl.v[7]);
        }
        __syncthreads();
    }
    u256 excl = Fr::one();
    if (tid > 0) {
        uint4 l = slo[tid - 1], h = shi[tid - 1];
        excl.v[0] = l.x; excl.v[1] = l.y; excl.v[2] = l.z; excl.v[3] = l.w; excl.v[4] = h.x; excl.v[5] = h.y; excl.v[6] = h.z; excl.v[7] = h.w;
    }
    if (total_out) {
        uint4 l = slo[blockDim.x - 1], h = shi[blockDim.x - 1];
        total_out->v[0] = l.x; total_out->v[1] = l.y; total_out->v[2] = l.z; total_out->v[3] = l.w;
        total_out->v[4] = h.x; total_out->v[5] = h.y; total_out->v[6] = h.z; total_out->v[7] = h.w;
    }
    __syncthreads();
    return excl;
}
// frac[i] = num[i] / den[i] in place on num (0 denominators invert to 0, as batch_invert does): Montgomery's trick on two levels — every thread
// chains its `chunk` rows (prefix products to scratch), the workgroup chains its threads' totals (prefix and suffix scans in LDS), ONE field
// inversion per workgroup (a thread-level inversion per 32 rows was 9 of the 12 products a row cost); 1 / total_t = 1/T * prefix_t * suffix_t.
// The rows of a thread are strided by the thread count (row j * threads + g), so the lanes of a wave touch neighbouring 32-byte elements at every step.
ZK_KERNEL void gp_batch_divide_kernel(void* num, const void* den, uint32_t n, uint32_t chunk, void* scratch) {
    __shared__ uint4 binv[2];
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = ((uint64_t)n + chunk - 1) / chunk;
    uint32_t cnt = 0;
    u256 acc = Fr::one();
    if (g < nth)
        for (uint64_t k = g; k < n && cnt < chunk; k += nth, cnt++) {
            store_u256(scratch, k, acc);
            const u256 d = load_u256(den, k);
            if (!Fr::is_zero(d)) acc = Fr::mul(acc, d);
        }
    u256 total;
    const u256 pre = gp_block_exclusive_scan<false>(acc, &total);
    const u256 suf = gp_block_exclusive_scan<true>(acc, nullptr);
    if (threadIdx.x == 0) {
        const u256 t = Fr::inv(total);
        binv[0] = make_uint4(t.v[0], t.v[1], t.v[2], t.v[3]);
        binv[1] = make_uint4(t.v[4], t.v[5], t.v[6], t.v[7]);
    }
    __syncthreads();
    u256 inv;
    { const uint4 l = binv[0], h = binv[1]; inv.v[0] = l.x; inv.v[1] = l.y; inv.v[2] = l.z; inv.v[3] = l.w; inv.v[4] = h.x; inv.v[5] = h.y; inv.v[6] = h.z; inv.v[7] = h.w; }
    inv = Fr::mul(inv, Fr::mul(pre, suf));
    for (uint32_t j = cnt; j-- > 0;) {
        const uint64_t k = g + (uint64_t)j * nth;
        const u256 d = load_u256(den, k);
        u256 di = Fr::zero();
        if (!Fr::is_zero(d)) { di = Fr::mul(inv, load_u256(scratch, k)); inv = Fr::mul(inv, d); }
        store_u256(num, k, Fr::mul(load_u256(num, k), di));
    }
}

// phase A: x[i] <- inclusive prefix product inside the workgroup's span; totals[b] <- product of the span
ZK_KERNEL void gp_scan_local_kernel(void* x, uint32_t n, void* totals) {
    const uint32_t span = blockDim.x * GP_E;
    const uint32_t base = blockIdx.x * span + threadIdx.x * GP_E;
    u256 v[GP_E];
    u256 run = Fr::one();
#pragma unroll
    for (uint32_t e = 0; e < GP_E; e++) {
        v[e] = base + e < n ? load_u256(x, base + e) : Fr::one();
        run = Fr::mul(run, v[e]);
        v[e] = run;
    }
    u256 total;
    const u256 excl = gp_block_exclusive_scan(run, &total);
#pragma unroll
    for (uint32_t e = 0; e < GP_E; e++)
        if (base + e < n) store_u256(x, base + e, Fr::mul(v[e], excl));
    if (threadIdx.x == 0) store_u256(totals, blockIdx.x, total);
}
// phase B (single workgroup): totals[b] <- exclusive prefix product of the workgroup totals
ZK_KERNEL void gp_scan_totals_kernel(void* totals_all, uint32_t nblocks) {
    void* totals = (char*)totals_all + (size_t)blockIdx.x * nblocks * 32;      // one workgroup per column of a batch
    const uint32_t per = (nblocks + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = threadIdx.x * per < nblocks ? threadIdx.x * per : nblocks;
    const uint32_t hi = lo + per < nblocks ? lo + per : nblocks;
    u256 run = Fr::one();
    for (uint32_t b = lo; b < hi; b++) run = Fr::mul(run, load_u256(totals, b));
    u256 acc = gp_block_exclusive_scan(run, nullptr);
    for (uint32_t b = lo; b < hi; b++) {
        const u256 t = load_u256(totals, b);
        store_u256(totals, b, acc);
        acc = Fr::mul(acc, t);
    }
}
// phase C: z[0] = init; z[i+1] = init * blockprefix * local[i]  (i + 1 < n_keep); rows >= n_keep take the blinding values
ZK_KERNEL void gp_assemble_kernel(const void* local, const void* block_prefix, uint32_t n, uint32_t n_keep, u256 init, const void* blinding, void* z) {
    const uint32_t span = GP_T * GP_E;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // output row
    if (i >= n) return;
    u256 o;
    if (i >= n_keep) o = load_u256(blinding, i - n_keep);
    else if (i == 0) o = init;
    else {
        const uint32_t src = (uint32_t)i - 1;
        o = Fr::mul(Fr::mul(load_u256(local, src), load_u256(block_prefix, src / span)), init);
    }
    store_u256(z, i, o);
}

// batched phase C: blockIdx.y = column; local / block_prefix / z are [col][..]; inits: per-column Montgomery scalars (device); blinding [col][bf]
ZK_KERNEL void gp_assemble_batch_kernel(const void* local, const void* block_prefix, uint32_t n, uint32_t n_keep, const void* inits, const void* blinding,
                                        uint32_t bf, void* const* zs, uint32_t nblocks) {
    const uint32_t span = GP_T * GP_E, c = blockIdx.y;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u256 init = load_u256(inits, c);
    u256 o;
    if (i >= n_keep) o = load_u256(blinding, (size_t)c * bf + (i - n_keep));
    else if (i == 0) o = init;
    else {
        const uint32_t src = (uint32_t)i - 1;
        o = Fr::mul(Fr::mul(load_u256(local, (size_t)c * n + src), load_u256(block_prefix, (size_t)c * nblocks + src / span)), init);
    }
    store_u256(zs[c], i, o);
}

// ---- host ------------------------------------------------------------------------------------------
static int gp_finish(zk_ctx* ctx, void* d_frac, void* d_den_scratch, void* d_aux, uint32_t n, const u256& init, const void* h_blinding, uint32_t bf,
                     void* d_z, void* h_last_z) {
    hipStream_t st = ctx->stream;
    const int blk = ctx->tune.vec_block;
    const uint32_t chunk = 32;
    const int dblk = std::min<int>(ctx->tune.vec_block, (int)GP_T);   // the division kernel scans over its workgroup in LDS arrays of GP_T entries
    ZK_LAUNCH(gp_batch_divide_kernel, (uint32_t)(((n + chunk - 1) / chunk + dblk - 1) / dblk), dblk, 0, st, d_frac, (const void*)d_den_scratch, n, chunk, d_aux);
    ZK_CHECK_LAUNCH();
    const uint32_t span = GP_T * GP_E, nblocks = (n + span - 1) / span;
    void* d_tot = (char*)d_aux;                       // scratch is free again after the division
    ZK_LAUNCH(gp_scan_local_kernel, nblocks, GP_T, 0, st, d_frac, n, d_tot);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(gp_scan_totals_kernel, 1, GP_T, 0, st, d_tot, nblocks);
    ZK_CHECK_LAUNCH();
    void* d_blind = (char*)d_aux + (size_t)nblocks * 32;
    if (bf) ZK_HIP(hipMemcpyAsync(d_blind, h_blinding, (size_t)bf * 32, hipMemcpyHostToDevice, st));
    ZK_LAUNCH(gp_assemble_kernel, (n + blk - 1) / blk, blk, 0, st, (const void*)d_frac, (const void*)d_tot, n, n - bf, init, (const void*)d_blind, d_z);
    ZK_CHECK_LAUNCH();
    if (h_last_z) ZK_HIP(hipMemcpyAsync(h_last_z, (char*)d_z + (size_t)(n - bf - 1) * 32, 32, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    return ZK_OK;
}

int permutation_product(zk_ctx* ctx, const void* const* values, const void* const* sigmas, size_t count, uint32_t k, const void* beta, const void* gamma,
                        const void* delta_start, const void* z_init, const void* blinding, uint32_t bf, void* d_z, void* h_last_z) {
    if (!values || !sigmas || !beta || !gamma || !delta_start || !z_init || !d_z || (bf && !blinding))
        return ctx->fail(ZK_ERR_ARG, "zk_permutation_product_dev: null argument");
    if (count == 0 || count > GP_MAX_COLS) return ctx->fail(ZK_ERR_LIMIT, "zk_permutation_product_dev: %zu columns per set (max %d)", count, GP_MAX_COLS);
    if (k > 27 || k < 1) return ctx->fail(ZK_ERR_ARG, "zk_permutation_product_dev: k = %u out of range", k);
    const uint32_t n = 1u << k;
    if (bf + 1 >= n) return ctx->fail(ZK_ERR_ARG, "zk_permutation_product_dev: blinding_factors too large");
    auto rd = [](const void* p) { u256 o; memcpy(&o, p, 32); return o; };
    GpPermArgs a;
    memset(&a, 0, sizeof a);
    a.count = (uint32_t)count; a.n = n; a.beta = rd(beta); a.gamma = rd(gamma);
    const uint64_t dl[4] = BN254_FR_DELTA_M;
    u256 delta;
    for (int i = 0; i < 8; i++) delta.v[i] = (uint32_t)(dl[i >> 1] >> (32 * (i & 1)));
    u256 cur = Fr::mul(rd(delta_start), a.beta);
    for (size_t j = 0; j < count; j++) {
        if (!values[j] || !sigmas[j]) return ctx->fail(ZK_ERR_ARG, "zk_permutation_product_dev: null column %zu", j);
        a.values[j] = values[j]; a.sigmas[j] = sigmas[j];
        a.delta_beta[j] = cur;
        cur = Fr::mul(cur, delta);
    }
    int rc = ntt_pow_tables(ctx, k, domain_omega(k), &a.tw_lo, &a.tw_hi, &a.lo_bits);
    if (rc) return rc;
    ZK_HIP(ctx->ws_tmp.ensure((size_t)n * 96 + (size_t)(n / (GP_T * GP_E) + 2 + bf) * 32 + 4096));
    a.num = ctx->ws_tmp.p;
    a.den = (char*)ctx->ws_tmp.p + (size_t)n * 32;
    void* d_aux = (char*)ctx->ws_tmp.p + (size_t)n * 64;
    const int blk = ctx->tune.vec_block;
    ZK_LAUNCH(gp_perm_fraction_kernel, (n + blk - 1) / blk, blk, 0, ctx->stream, a);
    ZK_CHECK_LAUNCH();
    return gp_finish(ctx, a.num, a.den, d_aux, n, rd(z_init), blinding, bf, d_z, h_last_z);
}

int lookup_product(zk_ctx* ctx, const void* cin, const void* ctab, const void* pin, const void* ptab, uint32_t k, const void* beta, const void* gamma,
                   const void* blinding, uint32_t bf, void* d_z) {
    if (!cin || !ctab || !pin || !ptab || !beta || !gamma || !d_z || (bf && !blinding)) return ctx->fail(ZK_ERR_ARG, "zk_lookup_product_dev: null argument");
    if (k > 27 || k < 1) return ctx->fail(ZK_ERR_ARG, "zk_lookup_product_dev: k = %u out of range", k);
    const uint32_t n = 1u << k;
    if (bf + 1 >= n) return ctx->fail(ZK_ERR_ARG, "zk_lookup_product_dev: blinding_factors too large");
    auto rd = [](const void* p) { u256 o; memcpy(&o, p, 32); return o; };
    ZK_HIP(ctx->ws_tmp.ensure((size_t)n * 96 + (size_t)(n / (GP_T * GP_E) + 2 + bf) * 32 + 4096));
    void* num = ctx->ws_tmp.p;
    void* den = (char*)ctx->ws_tmp.p + (size_t)n * 32;
    void* d_aux = (char*)ctx->ws_tmp.p + (size_t)n * 64;
    const int blk = ctx->tune.vec_block;
    ZK_LAUNCH(gp_lookup_fraction_kernel, (n + blk - 1) / blk, blk, 0, ctx->stream, cin, ctab, pin, ptab, n, rd(beta), rd(gamma), num, den);
    ZK_CHECK_LAUNCH();
    return gp_finish(ctx, num, den, d_aux, n, Fr::one(), blinding, bf, d_z, nullptr);
}


// All lookup grand products of a proof in one launch sequence (Permuted::commit_product for every lookup): the batch
// inversion's one Fermat chain per thread is paid once, not once per lookup.  Needs n to be a multiple of the scan span.
int lookup_product_batch(zk_ctx* ctx, const void* const* cols4, size_t count, uint32_t k, const void* beta, const void* gamma, const void* blinding,
                         uint32_t bf, void* const* d_zs) {
    if (count == 0) return ZK_OK;
    if (!cols4 || !beta || !gamma || !d_zs || (bf && !blinding)) return ctx->fail(ZK_ERR_ARG, "zk_lookup_product_batch_dev: null argument");
    if (k > 27 || k < 1 || count > 4096) return ctx->fail(ZK_ERR_ARG, "zk_lookup_product_batch_dev: k / count out of range");
    const uint32_t n = 1u << k;
    if (bf + 1 >= n) return ctx->fail(ZK_ERR_ARG, "zk_lookup_product_batch_dev: blinding_factors too large");
    for (size_t i = 0; i < 4 * count; i++) if (!cols4[i]) return ctx->fail(ZK_ERR_ARG, "zk_lookup_product_batch_dev: null column");
    for (size_t i = 0; i < count; i++) if (!d_zs[i]) return ctx->fail(ZK_ERR_ARG, "zk_lookup_product_batch_dev: null output");
    auto rd = [](const void* p) { u256 o; memcpy(&o, p, 32); return o; };
    const uint32_t span = GP_T * GP_E;
    if (n % span) {   // tiny domains: the scan spans would straddle columns
        for (size_t l = 0; l < count; l++) {
            int rc = lookup_product(ctx, cols4[4 * l], cols4[4 * l + 1], cols4[4 * l + 2], cols4[4 * l + 3], k, beta, gamma,
                                    (const char*)blinding + l * (size_t)bf * 32, bf, d_zs[l]);
            if (rc) return rc;
        }
        return ZK_OK;
    }
    const uint32_t nblocks = n / span;
    const size_t N = (size_t)count * n;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_num = take(N * 32), o_den = take(N * 32), o_scr = take(N * 32), o_tot = take((size_t)count * nblocks * 32), o_bl = take((size_t)count * (bf + 1) * 32),
                 o_in = take(count * 32), o_cp = take(4 * count * sizeof(void*)), o_zp = take(count * sizeof(void*));
    ZK_HIP(ctx->ws_tmp.ensure(off + 256));
    char* base = (char*)ctx->ws_tmp.p;
    hipStream_t st = ctx->stream;
    const int blk = ctx->tune.vec_block;
    ZK_HIP(hipMemcpyAsync(base + o_cp, cols4, 4 * count * sizeof(void*), hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(base + o_zp, d_zs, count * sizeof(void*), hipMemcpyHostToDevice, st));
    if (bf) ZK_HIP(hipMemcpyAsync(base + o_bl, blinding, (size_t)count * bf * 32, hipMemcpyHostToDevice, st));
    std::vector<u256> ones(count, Fr::one());
    ZK_HIP(hipMemcpyAsync(base + o_in, ones.data(), count * 32, hipMemcpyHostToDevice, st));
    ZK_LAUNCH(gp_lookup_fraction_batch_kernel, dim3((n + blk - 1) / blk, (uint32_t)count), blk, 0, st, (const void* const*)(base + o_cp), n, rd(beta), rd(gamma),
              (void*)(base + o_num), (void*)(base + o_den));
    ZK_CHECK_LAUNCH();
    const uint32_t chunk = 32;
    const int dblk = std::min<int>(ctx->tune.vec_block, (int)GP_T);   // the division kernel scans over its workgroup in LDS arrays of GP_T entries
    ZK_LAUNCH(gp_batch_divide_kernel, (uint32_t)(((N + chunk - 1) / chunk + dblk - 1) / dblk), dblk, 0, st, (void*)(base + o_num), (const void*)(base + o_den), (uint32_t)N, chunk,
              (void*)(base + o_scr));
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(gp_scan_local_kernel, (uint32_t)(count * nblocks), GP_T, 0, st, (void*)(base + o_num), (uint32_t)N, (void*)(base + o_tot));
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(gp_scan_totals_kernel, (uint32_t)count, GP_T, 0, st, (void*)(base + o_tot), nblocks);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(gp_assemble_batch_kernel, dim3((n + blk - 1) / blk, (uint32_t)count), blk, 0, st, (const void*)(base + o_num), (const void*)(base + o_tot), n, n - bf,
              (const void*)(base + o_in), (const void*)(base + o_bl), bf, (void* const*)(base + o_zp), nblocks);
    ZK_CHECK_LAUNCH();
    ZK_HIP(hipStreamSynchronize(st));
    return ZK_OK;
}


// All column sets of the permutation argument in one launch sequence (permutation::Argument::commit's loop over chunks).  The sets
// chain through z_s[0] = z_(s-1)[n - bf - 1]; here every set is first scanned with init = 1, the few chaining products are done on the
// host from two downloaded values per set, and the assemble pass applies them.  d_zs: n_sets outputs; blinding: n_sets x bf x 32 B.
int permutation_product_all(zk_ctx* ctx, const void* const* values, const void* const* sigmas, size_t m, uint32_t chunk_len, uint32_t k, const void* beta,
                            const void* gamma, const void* blinding, uint32_t bf, void* const* d_zs) {
    if (m == 0) return ZK_OK;
    if (!values || !sigmas || !beta || !gamma || !d_zs || (bf && !blinding) || chunk_len == 0) return ctx->fail(ZK_ERR_ARG, "zk_permutation_product_all_dev: null argument");
    if (chunk_len > (uint32_t)GP_MAX_COLS) return ctx->fail(ZK_ERR_LIMIT, "zk_permutation_product_all_dev: %u columns per set (max %d)", chunk_len, GP_MAX_COLS);
    if (k > 27 || k < 1) return ctx->fail(ZK_ERR_ARG, "zk_permutation_product_all_dev: k = %u out of range", k);
    const uint32_t n = 1u << k;
    if (bf + 2 >= n) return ctx->fail(ZK_ERR_ARG, "zk_permutation_product_all_dev: blinding_factors too large");
    const size_t n_sets = (m + chunk_len - 1) / chunk_len;
    auto rd = [](const void* p) { u256 o; memcpy(&o, p, 32); return o; };
    const uint64_t dl[4] = BN254_FR_DELTA_M;
    u256 delta;
    for (int i = 0; i < 8; i++) delta.v[i] = (uint32_t)(dl[i >> 1] >> (32 * (i & 1)));
    const uint32_t span = GP_T * GP_E;
    if (n % span) {   // tiny domains: set by set
        u256 z_init = Fr::one(), dstart = Fr::one();
        for (size_t sidx = 0; sidx < n_sets; sidx++) {
            const size_t lo = sidx * chunk_len, cnt = std::min<size_t>(chunk_len, m - lo);
            u256 last;
            int rc = permutation_product(ctx, values + lo, sigmas + lo, cnt, k, beta, gamma, &dstart, &z_init, (const char*)blinding + sidx * (size_t)bf * 32, bf,
                                         d_zs[sidx], &last);
            if (rc) return rc;
            z_init = last;
            for (size_t j = 0; j < cnt; j++) dstart = Fr::mul(dstart, delta);
        }
        return ZK_OK;
    }
    const uint32_t nblocks = n / span;
    const size_t N = n_sets * (size_t)n;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_num = take(N * 32), o_den = take(N * 32), o_scr = take(N * 32), o_tot = take(n_sets * nblocks * 32), o_bl = take(n_sets * (size_t)(bf + 1) * 32),
                 o_in = take(n_sets * 32), o_zp = take(n_sets * sizeof(void*));
    ZK_HIP(ctx->ws_tmp.ensure(off + 256));
    char* base = (char*)ctx->ws_tmp.p;
    hipStream_t st = ctx->stream;
    const int blk = ctx->tune.vec_block;
    GpPermArgs a;
    memset(&a, 0, sizeof a);
    a.n = n; a.beta = rd(beta); a.gamma = rd(gamma);
    int rc = ntt_pow_tables(ctx, k, domain_omega(k), &a.tw_lo, &a.tw_hi, &a.lo_bits);
    if (rc) return rc;
    ZK_HIP(ctx->ws_tmp.ensure(off + 256));     // (ntt_pow_tables may have touched other workspaces, not this one)
    base = (char*)ctx->ws_tmp.p;
    u256 cur = a.beta;                           // delta^j * beta, continuing across the sets
    for (size_t sidx = 0; sidx < n_sets; sidx++) {
        const size_t lo = sidx * chunk_len, cnt = std::min<size_t>(chunk_len, m - lo);
        a.count = (uint32_t)cnt;
        for (size_t j = 0; j < cnt; j++) {
            if (!values[lo + j] || !sigmas[lo + j]) return ctx->fail(ZK_ERR_ARG, "zk_permutation_product_all_dev: null column %zu", lo + j);
            a.values[j] = values[lo + j]; a.sigmas[j] = sigmas[lo + j];
            a.delta_beta[j] = cur;
            cur = Fr::mul(cur, delta);
        }
        a.num = base + o_num + sidx * (size_t)n * 32;
        a.den = base + o_den + sidx * (size_t)n * 32;
        ZK_LAUNCH(gp_perm_fraction_kernel, (n + blk - 1) / blk, blk, 0, st, a);
        ZK_CHECK_LAUNCH();
    }
    ZK_HIP(hipMemcpyAsync(base + o_zp, d_zs, n_sets * sizeof(void*), hipMemcpyHostToDevice, st));
    if (bf) ZK_HIP(hipMemcpyAsync(base + o_bl, blinding, n_sets * (size_t)bf * 32, hipMemcpyHostToDevice, st));
    const uint32_t chunk = 32;
    const int dblk = std::min<int>(ctx->tune.vec_block, (int)GP_T);   // the division kernel scans over its workgroup in LDS arrays of GP_T entries
    ZK_LAUNCH(gp_batch_divide_kernel, (uint32_t)(((N + chunk - 1) / chunk + dblk - 1) / dblk), dblk, 0, st, (void*)(base + o_num), (const void*)(base + o_den), (uint32_t)N, chunk,
              (void*)(base + o_scr));
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(gp_scan_local_kernel, (uint32_t)(n_sets * nblocks), GP_T, 0, st, (void*)(base + o_num), (uint32_t)N, (void*)(base + o_tot));
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(gp_scan_totals_kernel, (uint32_t)n_sets, GP_T, 0, st, (void*)(base + o_tot), nblocks);
    ZK_CHECK_LAUNCH();
    // chaining: P_s = prod_{j < n - bf - 1} frac_s[j] = local[src] * block_prefix[src / span], src = n - bf - 2
    const uint32_t src = n - bf - 2;
    std::vector<u256> loc(n_sets), pre(n_sets), inits(n_sets);
    for (size_t sidx = 0; sidx < n_sets; sidx++) {
        ZK_HIP(hipMemcpyAsync(&loc[sidx], base + o_num + (sidx * (size_t)n + src) * 32, 32, hipMemcpyDeviceToHost, st));
        ZK_HIP(hipMemcpyAsync(&pre[sidx], base + o_tot + (sidx * (size_t)nblocks + src / span) * 32, 32, hipMemcpyDeviceToHost, st));
    }
    ZK_HIP(hipStreamSynchronize(st));
    u256 init = Fr::one();
    for (size_t sidx = 0; sidx < n_sets; sidx++) { inits[sidx] = init; init = Fr::mul(init, Fr::mul(loc[sidx], pre[sidx])); }
    ZK_HIP(hipMemcpyAsync(base + o_in, inits.data(), n_sets * 32, hipMemcpyHostToDevice, st));
    ZK_LAUNCH(gp_assemble_batch_kernel, dim3((n + blk - 1) / blk, (uint32_t)n_sets), blk, 0, st, (const void*)(base + o_num), (const void*)(base + o_tot), n, n - bf,
              (const void*)(base + o_in), (const void*)(base + o_bl), bf, (void* const*)(base + o_zp), nblocks);
    ZK_CHECK_LAUNCH();
    ZK_HIP(hipStreamSynchronize(st));
    return ZK_OK;
}

}  // namespace zk
