// Lookup argument: permute_expression_pair — SURVEY.md §8(f) "next 4" (second half; the theta-compression half runs on
// the quotient interpreter, see evaluation.expression_program).
//
// Replaces halo2_proofs (zkwebauthn @ c254c75, Cargo.lock:1314-1327) src/plonk/lookup/prover.rs permute_expression_pair:
//   A' = the usable rows of the compressed input, sorted (Ord of Fr = numeric order of the canonical value);
//   S' : at the first occurrence of each distinct value of A' the same value (one copy is taken out of the table
//        multiset; a value missing from the table is a ConstraintSystemFailure); the remaining ("repeated") rows are
//        filled with the leftover table values in ascending order, handed out from the LAST repeated row backwards
//        (the CPU code pops a Vec);  both columns end with blinding_factors + 1 caller-supplied random rows.
// The CPU version is a sort + BTreeMap walk.  Here: LSD radix sort of row indices on 4-bit digits of the canonical
// 256-bit keys (stable, per-thread runs + one scan per digit, digits on which all keys agree are skipped), a
// lower_bound per distinct value to take its table copy, two prefix sums to rank repeated rows and leftover values,
// and one gather to assemble both columns.
#include "ctx.h"
#include <vector>

namespace zk {

constexpr uint32_t LP_RUN = 64;     // consecutive items ranked by one thread (stability)
constexpr uint32_t LP_T = 256;

// canonical keys + identity permutation; ormask[0..8) collects the OR of all keys so that the sort only visits the
// digits some key actually uses (witness values are mostly far below 254 bits)
ZK_KERNEL void lp_canon_kernel(const void* x, uint32_t u, void* canon, uint32_t* idx, uint32_t* ormask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= u) return;
    const u256 c = Fr::from_mont(load_u256(x, i));
    store_u256(canon, i, c);
    idx[i] = i;
#pragma unroll
    for (int w = 0; w < 8; w++) if (c.v[w] & ~ormask[w]) atomicOr(&ormask[w], c.v[w]);
}
ZK_HD uint32_t lp_digit(const void* canon, uint32_t row, uint32_t d) {  // d-th 4-bit digit, d = 0 least significant
    const uint32_t w = reinterpret_cast<const uint32_t*>(canon)[(size_t)row * 8 + (d >> 3)];
    return (w >> (4 * (d & 7))) & 15u;
}
// counts[bin * nthreads + t] = how many items of run t carry digit `bin`
ZK_KERNEL void lp_hist_kernel(const void* canon, const uint32_t* idx, uint32_t u, uint32_t d, uint32_t nruns, uint32_t* counts) {
    __shared__ uint32_t lc[16 * LP_T];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, tid = threadIdx.x;
    for (uint32_t b = 0; b < 16; b++) lc[b * LP_T + tid] = 0;
    if (t < nruns) {
        const uint32_t lo = t * LP_RUN, hi = lo + LP_RUN < u ? lo + LP_RUN : u;
        for (uint32_t i = lo; i < hi; i++) lc[lp_digit(canon, idx[i], d) * LP_T + tid]++;
        for (uint32_t b = 0; b < 16; b++) counts[(size_t)b * nruns + t] = lc[b * LP_T + tid];
    }
}
// single workgroup: exclusive scan of v[0..m) in place; *total_out = sum; flag[0] = 1 if some bin holds every item (digit pass is a no-op)
ZK_KERNEL void lp_scan_kernel(uint32_t* v, uint32_t m, uint32_t nruns, uint32_t u, uint32_t* flag, uint32_t* total_out) {
    __shared__ uint32_t part[1024];
    const uint32_t T = blockDim.x, tid = threadIdx.x;
    const uint32_t per = (m + T - 1) / T;
    const uint32_t lo = tid * per < m ? tid * per : m, hi = lo + per < m ? lo + per : m;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; i++) s += v[i];
    part[tid] = s;
    __syncthreads();
    for (uint32_t d = 1; d < T; d <<= 1) {
        const uint32_t add = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    uint32_t run = part[tid] - s;
    for (uint32_t i = lo; i < hi; i++) { const uint32_t x = v[i]; v[i] = run; run += x; }
    __syncthreads();
    if (tid == 0) {
        if (total_out) *total_out = part[T - 1];
        if (flag && nruns) {
            uint32_t skip = 0;
            for (uint32_t b = 0; b < 16; b++) {
                const uint32_t start = v[(size_t)b * nruns];
                const uint32_t end = b == 15 ? part[T - 1] : v[(size_t)(b + 1) * nruns];
                if (end - start == u) skip = 1;
            }
            flag[0] = skip;
        }
    }
}
ZK_KERNEL void lp_scatter_kernel(const void* canon, const uint32_t* idx_in, uint32_t* idx_out, uint32_t u, uint32_t d, uint32_t nruns,
                                 const uint32_t* offsets, const uint32_t* flag) {
    __shared__ uint32_t lo_[16 * LP_T];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, tid = threadIdx.x;
    if (t >= nruns) return;
    const uint32_t lo = t * LP_RUN, hi = lo + LP_RUN < u ? lo + LP_RUN : u;
    if (flag[0]) { for (uint32_t i = lo; i < hi; i++) idx_out[i] = idx_in[i]; return; }
    for (uint32_t b = 0; b < 16; b++) lo_[b * LP_T + tid] = offsets[(size_t)b * nruns + t];
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t r = idx_in[i];
        idx_out[lo_[lp_digit(canon, r, d) * LP_T + tid]++] = r;
    }
}

ZK_HD int lp_cmp(const u256& a, const u256& b) {  // numeric compare of canonical values
    for (int i = 7; i >= 0; i--) {
        if (a.v[i] < b.v[i]) return -1;
        if (a.v[i] > b.v[i]) return 1;
    }
    return 0;
}
// first[i] = 1 iff sorted input row i starts a new value; such a row takes one copy of its value out of the table
ZK_KERNEL void lp_mark_kernel(const void* in_canon, const uint32_t* in_idx, const void* tab_canon, const uint32_t* tab_idx, uint32_t u,
                              uint32_t* repeated, uint32_t* unconsumed, uint32_t* err) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= u) return;
    const u256 v = load_u256(in_canon, in_idx[i]);
    const bool first = i == 0 || lp_cmp(v, load_u256(in_canon, in_idx[i - 1])) != 0;
    repeated[i] = first ? 0u : 1u;
    if (!first) return;
    uint32_t lo = 0, hi = u;      // lower_bound in the sorted table
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (lp_cmp(load_u256(tab_canon, tab_idx[mid]), v) < 0) lo = mid + 1; else hi = mid;
    }
    if (lo >= u || lp_cmp(load_u256(tab_canon, tab_idx[lo]), v) != 0) { atomicAdd(err, 1u); return; }
    unconsumed[lo] = 0;           // distinct values hit distinct table slots
}
ZK_KERNEL void lp_fill_kernel(uint32_t* a, uint32_t n, uint32_t val) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = val;
}
// leftover[rank] = sorted-table position of the rank-th unconsumed table entry
ZK_KERNEL void lp_compact_kernel(const uint32_t* unconsumed_flag, const uint32_t* rank, uint32_t u, uint32_t* leftover) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < u && unconsumed_flag[t]) leftover[rank[t]] = t;
}
ZK_KERNEL void lp_assemble_kernel(const void* in_mont, const uint32_t* in_idx, const void* tab_mont, const uint32_t* tab_idx, const uint32_t* repeated_flag,
                                  const uint32_t* rep_rank, const uint32_t* leftover, uint32_t n_rep, uint32_t u, uint32_t n, const void* blind_in,
                                  const void* blind_tab, void* out_in, void* out_tab) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i >= u) {
        store_u256(out_in, i, load_u256(blind_in, i - u));
        store_u256(out_tab, i, load_u256(blind_tab, i - u));
        return;
    }
    const u256 v = load_u256(in_mont, in_idx[i]);
    store_u256(out_in, i, v);
    if (!repeated_flag[i]) store_u256(out_tab, i, v);
    else store_u256(out_tab, i, load_u256(tab_mont, tab_idx[leftover[n_rep - 1 - rep_rank[i]]]));   // last repeated row gets the smallest leftover
}

// ---- host ------------------------------------------------------------------------------------------
// sorts `idx` (0..u) by the canonical keys; returns the buffer that holds the result
static uint32_t lp_scan_threads(uint32_t m) {   // single-workgroup scan: enough threads for ~8 items each, power of two
    uint32_t t = 64;
    while (t < 1024 && t * 8 < m) t <<= 1;
    return t;
}
static int lp_sort(zk_ctx* ctx, const void* canon, uint32_t u, uint32_t ndigits, uint32_t* idx_a, uint32_t* idx_b, uint32_t* counts, uint32_t* flag, uint32_t** result) {
    const uint32_t nruns = (u + LP_RUN - 1) / LP_RUN;
    const uint32_t grid = (nruns + LP_T - 1) / LP_T;
    uint32_t* in = idx_a;
    uint32_t* out = idx_b;
    for (uint32_t d = 0; d < ndigits; d++) {
        ZK_LAUNCH(lp_hist_kernel, grid, LP_T, 0, ctx->stream, canon, (const uint32_t*)in, u, d, nruns, counts);
        ZK_CHECK_LAUNCH();
        ZK_LAUNCH(lp_scan_kernel, 1, lp_scan_threads(16 * nruns), 0, ctx->stream, counts, 16 * nruns, nruns, u, flag, (uint32_t*)nullptr);
        ZK_CHECK_LAUNCH();
        ZK_LAUNCH(lp_scatter_kernel, grid, LP_T, 0, ctx->stream, canon, (const uint32_t*)in, out, u, d, nruns, (const uint32_t*)counts, (const uint32_t*)flag);
        ZK_CHECK_LAUNCH();
        std::swap(in, out);
    }
    *result = in;
    return ZK_OK;
}

int lookup_permute(zk_ctx* ctx, const void* d_input, const void* d_table, uint32_t k, uint32_t blinding_factors, const void* h_blind_input,
                   const void* h_blind_table, void* d_out_input, void* d_out_table) {
    if (!d_input || !d_table || !d_out_input || !d_out_table || !h_blind_input || !h_blind_table)
        return ctx->fail(ZK_ERR_ARG, "zk_lookup_permute_dev: null argument");
    if (k < 1 || k > 26) return ctx->fail(ZK_ERR_ARG, "zk_lookup_permute_dev: k = %u out of range", k);
    const uint32_t n = 1u << k, nb = blinding_factors + 1;
    if (nb >= n) return ctx->fail(ZK_ERR_ARG, "zk_lookup_permute_dev: blinding_factors too large");
    const uint32_t u = n - nb;
    const uint32_t nruns = (u + LP_RUN - 1) / LP_RUN;
    // workspace: canon_in | canon_tab | blind (2*nb) | u32 arrays
    const size_t words = (size_t)4 * u /*idx in a/b, tab a/b*/ + (size_t)16 * nruns + 4 * (size_t)u /*repeated, rep_rank, unconsumed, left_rank*/ + u + 64;
    ZK_HIP(ctx->ws_tmp.ensure((size_t)u * 64 + (size_t)nb * 64 + words * 4 + 256));
    char* base = (char*)ctx->ws_tmp.p;
    void* canon_in = base;
    void* canon_tab = base + (size_t)u * 32;
    void* d_blind_in = base + (size_t)u * 64;
    void* d_blind_tab = (char*)d_blind_in + (size_t)nb * 32;
    uint32_t* w = (uint32_t*)((char*)d_blind_tab + (size_t)nb * 32);
    uint32_t* in_a = w; uint32_t* in_b = in_a + u; uint32_t* tab_a = in_b + u; uint32_t* tab_b = tab_a + u;
    uint32_t* counts = tab_b + u;
    uint32_t* repeated = counts + (size_t)16 * nruns; uint32_t* rep_rank = repeated + u;
    uint32_t* unconsumed = rep_rank + u; uint32_t* left_rank = unconsumed + u;
    uint32_t* leftover = left_rank + u;
    uint32_t* scal = leftover + u;      // [0] skip flag, [1] error count, [2] n_rep, [3] n_left
    hipStream_t st = ctx->stream;
    const int blk = ctx->tune.vec_block;
    const uint32_t g = (u + blk - 1) / blk;
    ZK_HIP(hipMemsetAsync(scal, 0, 64, st));
    ZK_HIP(hipMemcpyAsync(d_blind_in, h_blind_input, (size_t)nb * 32, hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(d_blind_tab, h_blind_table, (size_t)nb * 32, hipMemcpyHostToDevice, st));
    ZK_LAUNCH(lp_canon_kernel, g, blk, 0, st, d_input, u, canon_in, in_a, scal + 4);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(lp_canon_kernel, g, blk, 0, st, d_table, u, canon_tab, tab_a, scal + 4);
    ZK_CHECK_LAUNCH();
    uint32_t om[8];
    ZK_HIP(hipMemcpyAsync(om, scal + 4, 32, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    uint32_t ndigits = 1;                                   // number of 4-bit digits any key uses
    for (int wd = 7; wd >= 0; wd--)
        if (om[wd]) { uint32_t top = 31; while (!((om[wd] >> top) & 1)) top--; ndigits = (uint32_t)wd * 8 + top / 4 + 1; break; }
    uint32_t *in_sorted, *tab_sorted;
    int rc = lp_sort(ctx, canon_in, u, ndigits, in_a, in_b, counts, scal, &in_sorted);
    if (rc) return rc;
    rc = lp_sort(ctx, canon_tab, u, ndigits, tab_a, tab_b, counts, scal, &tab_sorted);
    if (rc) return rc;
    ZK_LAUNCH(lp_fill_kernel, g, blk, 0, st, unconsumed, u, 1u);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(lp_mark_kernel, g, blk, 0, st, (const void*)canon_in, (const uint32_t*)in_sorted, (const void*)canon_tab, (const uint32_t*)tab_sorted, u, repeated,
              unconsumed, scal + 1);
    ZK_CHECK_LAUNCH();
    ZK_HIP(hipMemcpyAsync(rep_rank, repeated, (size_t)u * 4, hipMemcpyDeviceToDevice, st));
    ZK_HIP(hipMemcpyAsync(left_rank, unconsumed, (size_t)u * 4, hipMemcpyDeviceToDevice, st));
    ZK_LAUNCH(lp_scan_kernel, 1, lp_scan_threads(u), 0, st, rep_rank, u, 0u, u, (uint32_t*)nullptr, scal + 2);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(lp_scan_kernel, 1, lp_scan_threads(u), 0, st, left_rank, u, 0u, u, (uint32_t*)nullptr, scal + 3);
    ZK_CHECK_LAUNCH();
    uint32_t hs[4];
    ZK_HIP(hipMemcpyAsync(hs, scal, 16, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    if (hs[1]) return ctx->fail(ZK_ERR_ARG, "zk_lookup_permute_dev: %u input value(s) are not in the table (halo2: Error::ConstraintSystemFailure)", hs[1]);
    if (hs[2] != hs[3]) return ctx->fail(ZK_ERR_ARG, "zk_lookup_permute_dev: internal count mismatch (%u repeated rows, %u leftover table values)", hs[2], hs[3]);
    ZK_LAUNCH(lp_compact_kernel, g, blk, 0, st, (const uint32_t*)unconsumed, (const uint32_t*)left_rank, u, leftover);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(lp_assemble_kernel, (n + blk - 1) / blk, blk, 0, st, d_input, (const uint32_t*)in_sorted, d_table, (const uint32_t*)tab_sorted,
              (const uint32_t*)repeated, (const uint32_t*)rep_rank, (const uint32_t*)leftover, hs[2], u, n, (const void*)d_blind_in, (const void*)d_blind_tab,
              d_out_input, d_out_table);
    ZK_CHECK_LAUNCH();
    ZK_HIP(hipStreamSynchronize(st));
    return ZK_OK;
}

}  // namespace zk
