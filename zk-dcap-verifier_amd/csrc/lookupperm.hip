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
    __shared__ uint32_t lor[8];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < 8) lor[threadIdx.x] = 0;
    __syncthreads();
    if (i < u) {
        const u256 c = Fr::from_mont(load_u256(x, i));
        store_u256(canon, i, c);
        idx[i] = i;
#pragma unroll
        for (int w = 0; w < 8; w++) if (c.v[w] & ~lor[w]) atomicOr(&lor[w], c.v[w]);   // workgroup-local first: the global words are hit once per workgroup
    }
    __syncthreads();
    if (threadIdx.x < 8 && lor[threadIdx.x]) atomicOr(&ormask[threadIdx.x], lor[threadIdx.x]);
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


// ---- fast path: stable LSD radix sort of (64-bit key, row index) pairs, 8-bit digits -----------------------------
// The sort key is the 64-bit window of the canonical value that ends at the highest bit any key of the pair of columns
// uses (theta-compressed expressions are full-width, range-check inputs are a few bits wide).  When that window does
// not cover a key completely the result is checked against the full 256-bit order afterwards and the generic 4-bit
// path above is taken if two rows tie on the window but differ below it (for random values: never in practice).
// Per pass: one histogram launch and one scatter launch.  A workgroup owns a tile of FS_T * FS_E consecutive pairs;
// stability inside the tile comes from per-thread runs ranked through an LDS count matrix [thread][digit], stability
// across tiles from the workgroup-major global histogram (each workgroup sums the rows of the workgroups before it).
constexpr uint32_t FS_T = 256, FS_E = 16, FS_TILE = FS_T * FS_E, FS_PAD = 258;

ZK_HD uint32_t fs_digit(uint2 k, uint32_t pass) { return ((pass < 4 ? k.x : k.y) >> (8 * (pass & 3))) & 255u; }

// ---- exclusive scan of u32 flags over many workgroups (ranks of repeated rows / leftover table values) ----------------
constexpr uint32_t XS_T = 256, XS_E = 8, XS_TILE = XS_T * XS_E;
ZK_KERNEL void xs_tile_sum_kernel(const uint32_t* v, uint32_t m, uint32_t* sums) {
    __shared__ uint32_t part[XS_T];
    const uint32_t tid = threadIdx.x, base = blockIdx.x * XS_TILE;
    uint32_t s = 0;
    for (uint32_t e = 0; e < XS_E; e++) { const uint32_t i = base + e * XS_T + tid; if (i < m) s += v[i]; }
    part[tid] = s;
    __syncthreads();
    for (uint32_t d = XS_T >> 1; d > 0; d >>= 1) { if (tid < d) part[tid] += part[tid + d]; __syncthreads(); }
    if (tid == 0) sums[blockIdx.x] = part[0];
}
// in place: v[i] <- exclusive prefix; sums[] must already hold the EXCLUSIVE scan of the tile sums
ZK_KERNEL void xs_apply_kernel(uint32_t* v, uint32_t m, const uint32_t* sums) {
    __shared__ uint32_t part[XS_T];
    const uint32_t tid = threadIdx.x, lo = blockIdx.x * XS_TILE + tid * XS_E;
    uint32_t x[XS_E], s = 0;
#pragma unroll
    for (uint32_t e = 0; e < XS_E; e++) { x[e] = lo + e < m ? v[lo + e] : 0u; s += x[e]; }
    part[tid] = s;
    __syncthreads();
    for (uint32_t d = 1; d < XS_T; d <<= 1) {
        const uint32_t add = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    uint32_t run = sums[blockIdx.x] + part[tid] - s;
#pragma unroll
    for (uint32_t e = 0; e < XS_E; e++) { if (lo + e < m) v[lo + e] = run; run += x[e]; }
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


// v[0..m) <- exclusive scan, *total_dev <- sum (sums: scratch of ceil(m / XS_TILE) + 1 words)
static int xs_scan(zk_ctx* ctx, uint32_t* v, uint32_t m, uint32_t* sums, uint32_t* total_dev) {
    hipStream_t st = ctx->stream;
    const uint32_t nt = (m + XS_TILE - 1) / XS_TILE;
    ZK_LAUNCH(xs_tile_sum_kernel, nt, XS_T, 0, st, (const uint32_t*)v, m, sums);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(lp_scan_kernel, 1, lp_scan_threads(nt), 0, st, sums, nt, 0u, m, (uint32_t*)nullptr, total_dev);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(xs_apply_kernel, nt, XS_T, 0, st, v, m, (const uint32_t*)sums);
    ZK_CHECK_LAUNCH();
    return ZK_OK;
}

// One lookup on the every-digit path (4-bit digits over all bits in use): the fallback of the batched form below for a lookup whose keys tie on the
// 64-bit window but differ below it.
static int lookup_permute_one(zk_ctx* ctx, const void* d_input, const void* d_table, uint32_t k, uint32_t blinding_factors, const void* h_blind_input,
                              const void* h_blind_table, void* d_out_input, void* d_out_table) {
    if (!d_input || !d_table || !d_out_input || !d_out_table || !h_blind_input || !h_blind_table)
        return ctx->fail(ZK_ERR_ARG, "zk_lookup_permute_dev: null argument");
    if (k < 1 || k > 26) return ctx->fail(ZK_ERR_ARG, "zk_lookup_permute_dev: k = %u out of range", k);
    const uint32_t n = 1u << k, nb = blinding_factors + 1;
    if (nb >= n) return ctx->fail(ZK_ERR_ARG, "zk_lookup_permute_dev: blinding_factors too large");
    const uint32_t u = n - nb;
    const uint32_t nruns = (u + LP_RUN - 1) / LP_RUN;
    const uint32_t nxs = (u + XS_TILE - 1) / XS_TILE + 1;
    // workspace: canon_in | canon_tab | blind (2*nb) | u32 arrays
    const size_t words = (size_t)4 * u /*idx in a/b, tab a/b*/ + (size_t)16 * nruns + 4 * (size_t)u /*repeated, rep_rank, unconsumed, left_rank*/ + u + nxs + 64;
    ZK_HIP(ctx->ws_tmp.ensure((size_t)u * 64 + (size_t)nb * 64 + words * 4 + 512));
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
    uint32_t* xsums = leftover + u;
    uint32_t* scal = xsums + nxs;       // [0] skip flag, [1] error count, [2] n_rep, [3] n_left, [4..12) OR of all keys
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
    uint32_t nbits = 1;                                     // number of bits any key of either column uses
    for (int wd = 7; wd >= 0; wd--)
        if (om[wd]) { uint32_t top = 31; while (!((om[wd] >> top) & 1)) top--; nbits = (uint32_t)wd * 32 + top + 1; break; }
    const uint32_t ndigits = (nbits + 3) / 4;
    uint32_t *in_sorted = nullptr, *tab_sorted = nullptr;
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
    rc = xs_scan(ctx, rep_rank, u, xsums, scal + 2);
    if (rc) return rc;
    rc = xs_scan(ctx, left_rank, u, xsums, scal + 3);
    if (rc) return rc;
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


// ================================================================================================================
// batched form: all lookup arguments of a proof in one call.  Every kernel below is the single-lookup kernel with
// blockIdx.y selecting the sort (s = 2 * lookup + {0 input, 1 table}) or the lookup, so a proof pays 2 launches per
// radix pass instead of 2 per pass per column, and the grids are large enough to fill the chip.
// Layout (S = 2 * count sorts, u rows each): canon[s][u] | keys a/b [s][u] | idx a/b [s][u] | flags[s][u] (repeated / unconsumed)
// | ranks[s][u] | leftover[l][u] | ghist[s][nwg][256] | xsums[s][nxs] | scal[l][16]
// ================================================================================================================
// Sorted columns are numbered c < C: c < count is the input of lookup c, c >= count a DISTINCT table column — lookups that share their
// (compressed) table column, as range checks and the base64 lookups of the sgx circuit do, sort it once (tabcol[l] = its number).
struct LpbArgs {
    const void* const* cols;     // S device pointers: input_0, table_0, input_1, table_1, ...
    void* const* outs;           // S device pointers: out_input_0, out_table_0, ...
    const void* const* scols;    // C device pointers: the sorted columns
    const uint32_t* tabcol;      // count words
    uint32_t u, n, nb, S, C, nwg, nxs;
    void* canon; uint2* key_a; uint2* key_b; uint32_t* idx_a; uint32_t* idx_b;       // [C][u]
    uint32_t* flags; uint32_t* ranks; uint32_t* leftover; uint32_t* ghist; uint32_t* xsums; uint32_t* scal;   // flags / ranks [S][u], scal [count][16]
    uint32_t* cscal;             // [C][16]: [0..8) OR of the column's keys, [8] order violations
    const uint32_t* shifts;      // C words
    const void* blind;           // [l][2][nb] x 32 B
};
ZK_KERNEL void lpb_canon_kernel(LpbArgs a) {
    __shared__ uint32_t lor[8];
    const uint32_t s = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < 8) lor[threadIdx.x] = 0;
    __syncthreads();
    if (i < a.u) {
        const u256 c = Fr::from_mont(load_u256(a.scols[s], i));
        store_u256(a.canon, (size_t)s * a.u + i, c);
#pragma unroll
        for (int w = 0; w < 8; w++) if (c.v[w] & ~lor[w]) atomicOr(&lor[w], c.v[w]);
    }
    __syncthreads();
    if (threadIdx.x < 8 && lor[threadIdx.x]) atomicOr(&a.cscal[s * 16 + threadIdx.x], lor[threadIdx.x]);
}
ZK_KERNEL void lpb_keys_kernel(LpbArgs a) {
    const uint32_t s = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.u) return;
    const uint32_t shift = a.shifts[s];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(a.canon) + ((size_t)s * a.u + i) * 8;
    const uint32_t ws = shift >> 5, bs = shift & 31;
    uint32_t x[3];
#pragma unroll
    for (int j = 0; j < 3; j++) x[j] = ws + j < 8 ? w[ws + j] : 0u;
    uint2 k;
    k.x = bs ? (x[0] >> bs) | (x[1] << (32 - bs)) : x[0];
    k.y = bs ? (x[1] >> bs) | (x[2] << (32 - bs)) : x[1];
    a.key_a[(size_t)s * a.u + i] = k;
    a.idx_a[(size_t)s * a.u + i] = i;
}
ZK_KERNEL void lpb_hist_kernel(const uint2* keys, uint32_t u, uint32_t pass, uint32_t* ghist, uint32_t nwg) {
    __shared__ uint32_t lh[256];
    const uint32_t s = blockIdx.y, tid = threadIdx.x, base = blockIdx.x * FS_TILE;
    keys += (size_t)s * u;
    lh[tid] = 0;
    __syncthreads();
    for (uint32_t e = 0; e < FS_E; e++) {
        const uint32_t i = base + e * FS_T + tid;
        bool live = i < u;
        const uint32_t d = live ? fs_digit(keys[i], pass) : 0u;
#ifndef ZK_EMU
        // Most passes of a lookup's sort see ONE hot digit (the upper digits of small values, the zero padding of a column): 64 lanes adding to one LDS word serialise —
        // 96 % of this kernel's LDS cycles were conflicts.  Two rounds of wave aggregation take the lanes that share the first live lane's digit out with one add
        // of their count; what is left goes lane by lane as before (uniform digits pay two ballots).
#pragma unroll
        for (int round = 0; round < 2; round++) {
            if (live) {
                const uint32_t lead = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
                const bool same = d == lead;
                const unsigned long long m = __ballot(same);
                if (same) {
                    if ((__lane_id() & 63u) == (uint32_t)__builtin_ctzll(m)) atomicAdd(&lh[lead], (uint32_t)__builtin_popcountll(m));
                    live = false;
                }
            }
        }
#endif
        if (live) atomicAdd(&lh[d], 1u);
    }
    __syncthreads();
    ghist[((size_t)s * nwg + blockIdx.x) * 256 + tid] = lh[tid];
}
// gbase[s][wg][bin] <- pairs of sort s with a smaller digit anywhere + pairs with this digit in earlier tiles (in place on ghist):
// one workgroup per sort, thread = bin, so the scatter workgroups read their 256 bases instead of each summing nwg histogram rows
ZK_KERNEL void lpb_offsets_kernel(uint32_t* ghist, uint32_t nwg) {
    __shared__ uint32_t scan[256];
    const uint32_t s = blockIdx.x, tid = threadIdx.x;
    uint32_t* h = ghist + (size_t)s * nwg * 256;
    uint32_t total = 0;
    for (uint32_t w = 0; w < nwg; w++) total += h[(size_t)w * 256 + tid];
    scan[tid] = total;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint32_t add = tid >= d ? scan[tid - d] : 0;
        __syncthreads();
        scan[tid] += add;
        __syncthreads();
    }
    uint32_t run = scan[tid] - total;
    for (uint32_t w = 0; w < nwg; w++) {
        const uint32_t c = h[(size_t)w * 256 + tid];
        h[(size_t)w * 256 + tid] = run;
        run += c;
    }
}
ZK_KERNEL void lpb_scatter_kernel(const uint2* keys_in, const uint32_t* idx_in, uint2* keys_out, uint32_t* idx_out, uint32_t u, uint32_t pass,
                                  const uint32_t* gbase_all, uint32_t nwg) {
    __shared__ __attribute__((aligned(16))) uint16_t cnt[FS_T * FS_PAD];
    __shared__ uint32_t gbase[256];
    const uint32_t s = blockIdx.y, tid = threadIdx.x, wg = blockIdx.x, base = wg * FS_TILE;
    keys_in += (size_t)s * u; idx_in += (size_t)s * u; keys_out += (size_t)s * u; idx_out += (size_t)s * u;
    {   // clear the count matrix with 16-byte stores (FS_T * FS_PAD * 2 bytes = a whole number of uint4)
        uint4* c4 = reinterpret_cast<uint4*>(cnt);
        constexpr uint32_t n4 = FS_T * FS_PAD * 2 / 16;
        for (uint32_t q = tid; q < n4; q += FS_T) c4[q] = make_uint4(0, 0, 0, 0);
    }
    gbase[tid] = gbase_all[((size_t)s * nwg + wg) * 256 + tid];
    __syncthreads();
    uint2 k[FS_E];
    uint32_t v[FS_E];
    uint16_t rl[FS_E];
    const uint32_t lo = base + tid * FS_E;
#pragma unroll
    for (uint32_t e = 0; e < FS_E; e++) {
        const uint32_t i = lo + e;
        if (i < u) {
            k[e] = keys_in[i]; v[e] = idx_in[i];
            rl[e] = cnt[tid * FS_PAD + fs_digit(k[e], pass)]++;
        }
    }
    __syncthreads();
    {
        uint32_t run = 0;
        for (uint32_t t = 0; t < FS_T; t++) {
            const uint32_t c = cnt[t * FS_PAD + tid];
            cnt[t * FS_PAD + tid] = (uint16_t)run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t e = 0; e < FS_E; e++) {
        const uint32_t i = lo + e;
        if (i < u) {
            const uint32_t d = fs_digit(k[e], pass);
            const uint32_t pos = gbase[d] + cnt[tid * FS_PAD + d] + rl[e];
            keys_out[pos] = k[e];
            idx_out[pos] = v[e];
        }
    }
}
ZK_KERNEL void lpb_check_kernel(LpbArgs a, const uint32_t* idx_sorted) {
    const uint32_t s = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 || i >= a.u || a.shifts[s] == 0) return;
    const uint32_t* canon = reinterpret_cast<const uint32_t*>(a.canon) + (size_t)s * a.u * 8;
    const uint32_t* idx = idx_sorted + (size_t)s * a.u;
    const uint32_t* x = canon + (size_t)idx[i - 1] * 8;
    const uint32_t* y = canon + (size_t)idx[i] * 8;
    for (int w = 7; w >= 0; w--) {
        if (x[w] < y[w]) return;
        if (x[w] > y[w]) { atomicAdd(&a.cscal[s * 16 + 8], 1u); return; }
    }
}
// Columns whose rows tie on the 64-bit window but differ below it (theta-compressed lookups whose LAST expression differs: v and v + small —
// the base64 lookups of the sgx circuit do, their last expression is a 2-bit chunk): cscal[s][0..8) <- OR over adjacent window-tied rows of x ^ y.
// Any two tied values differ only inside that mask (x ^ z is the XOR of the adjacent XORs between them), so a stable LSD sort over the mask's
// bit range followed by the window passes orders the column exactly — a few extra passes instead of the every-digit sort.
ZK_KERNEL void lpb_tiemask_kernel(LpbArgs a, const uint32_t* idx_sorted) {
    __shared__ uint32_t lor[8];
    const uint32_t s = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (a.cscal[s * 16 + 8] == 0) return;                               // (uniform per column)
    if (threadIdx.x < 8) lor[threadIdx.x] = 0;
    __syncthreads();
    if (i > 0 && i < a.u) {
        const uint32_t shift = a.shifts[s];
        const uint32_t* canon = reinterpret_cast<const uint32_t*>(a.canon) + (size_t)s * a.u * 8;
        const uint32_t* idx = idx_sorted + (size_t)s * a.u;
        const uint32_t* x = canon + (size_t)idx[i - 1] * 8;
        const uint32_t* y = canon + (size_t)idx[i] * 8;
        uint32_t d[8], above = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            d[w] = x[w] ^ y[w];
            const uint32_t lo_bit = (uint32_t)w * 32;                   // bits of word w at or above `shift` belong to the window
            uint32_t in_window = lo_bit + 32 <= shift ? 0u : (lo_bit >= shift ? 0xffffffffu : (0xffffffffu << (shift - lo_bit)));
            above |= d[w] & in_window;
        }
        if (above == 0) {
#pragma unroll
            for (int w = 0; w < 8; w++) if (d[w] & ~lor[w]) atomicOr(&lor[w], d[w]);
        }
    }
    __syncthreads();
    if (threadIdx.x < 8 && lor[threadIdx.x]) atomicOr(&a.cscal[s * 16 + threadIdx.x], lor[threadIdx.x]);
}
// refinement sort on a COMPACT list of columns (f < F, column fcols[f]): keys of the next stage = the 64-bit window at fshift[f] of the canonical
// values, taken in the order of the previous stage (idx_in; null = identity)
ZK_KERNEL void lpb_rekey_kernel(LpbArgs a, const uint32_t* fcols, const uint32_t* fshift, const uint32_t* idx_in, uint2* key_out, uint32_t* idx_out) {
    const uint32_t f = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.u) return;
    const uint32_t c = fcols[f], shift = fshift[f];
    const uint32_t r = idx_in ? idx_in[(size_t)f * a.u + i] : i;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(a.canon) + ((size_t)c * a.u + r) * 8;
    const uint32_t ws = shift >> 5, bs = shift & 31;
    uint32_t x[3];
#pragma unroll
    for (int j = 0; j < 3; j++) x[j] = ws + j < 8 ? w[ws + j] : 0u;
    uint2 k;
    k.x = bs ? (x[0] >> bs) | (x[1] << (32 - bs)) : x[0];
    k.y = bs ? (x[1] >> bs) | (x[2] << (32 - bs)) : x[1];
    key_out[(size_t)f * a.u + i] = k;
    idx_out[(size_t)f * a.u + i] = r;
}
// flags[2l][i] = repeated, flags[2l+1][t] = unconsumed (pre-set to 1 by the caller)
ZK_KERNEL void lpb_mark_kernel(LpbArgs a, const uint32_t* idx_sorted) {
    const uint32_t l = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x, u = a.u;
    if (i >= u) return;
    const uint32_t tc = a.tabcol[l];
    const void* in_canon = (const char*)a.canon + (size_t)l * u * 32;
    const void* tab_canon = (const char*)a.canon + (size_t)tc * u * 32;
    const uint32_t* in_idx = idx_sorted + (size_t)l * u;
    const uint32_t* tab_idx = idx_sorted + (size_t)tc * u;
    uint32_t* repeated = a.flags + (size_t)(2 * l) * u;
    uint32_t* unconsumed = a.flags + (size_t)(2 * l + 1) * u;
    const u256 v = load_u256(in_canon, in_idx[i]);
    const bool first = i == 0 || lp_cmp(v, load_u256(in_canon, in_idx[i - 1])) != 0;
    repeated[i] = first ? 0u : 1u;
    if (!first) return;
    uint32_t lo = 0, hi = u;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (lp_cmp(load_u256(tab_canon, tab_idx[mid]), v) < 0) lo = mid + 1; else hi = mid;
    }
    if (lo >= u || lp_cmp(load_u256(tab_canon, tab_idx[lo]), v) != 0) { atomicAdd(&a.scal[l * 16 + 1], 1u); return; }
    unconsumed[lo] = 0;
}
ZK_KERNEL void lpb_fill_unconsumed_kernel(LpbArgs a) {
    const uint32_t l = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.u) a.flags[(size_t)(2 * l + 1) * a.u + i] = 1u;
}
ZK_KERNEL void lpb_tile_sum_kernel(LpbArgs a) {      // over ranks[s] (a copy of flags[s])
    __shared__ uint32_t part[XS_T];
    const uint32_t s = blockIdx.y, tid = threadIdx.x, base = blockIdx.x * XS_TILE;
    const uint32_t* v = a.ranks + (size_t)s * a.u;
    uint32_t sum = 0;
    for (uint32_t e = 0; e < XS_E; e++) { const uint32_t i = base + e * XS_T + tid; if (i < a.u) sum += v[i]; }
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = XS_T >> 1; d > 0; d >>= 1) { if (tid < d) part[tid] += part[tid + d]; __syncthreads(); }
    if (tid == 0) a.xsums[(size_t)s * a.nxs + blockIdx.x] = part[0];
}
ZK_KERNEL void lpb_sum_scan_kernel(LpbArgs a) {      // one workgroup per sort: exclusive scan of its tile sums, total -> scal[l][2 + j]
    __shared__ uint32_t part[1024];
    const uint32_t s = blockIdx.x, T = blockDim.x, tid = threadIdx.x, m = a.nxs - 1;
    uint32_t* v = a.xsums + (size_t)s * a.nxs;
    const uint32_t per = (m + T - 1) / T;
    const uint32_t lo = tid * per < m ? tid * per : m, hi = lo + per < m ? lo + per : m;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += v[i];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < T; d <<= 1) {
        const uint32_t add = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    for (uint32_t i = lo; i < hi; i++) { const uint32_t x = v[i]; v[i] = run; run += x; }
    if (tid == T - 1) a.scal[(s >> 1) * 16 + 2 + (s & 1)] = part[T - 1];
}
ZK_KERNEL void lpb_apply_kernel(LpbArgs a) {
    __shared__ uint32_t part[XS_T];
    const uint32_t s = blockIdx.y, tid = threadIdx.x, lo = blockIdx.x * XS_TILE + tid * XS_E, m = a.u;
    uint32_t* v = a.ranks + (size_t)s * a.u;
    uint32_t x[XS_E], sum = 0;
#pragma unroll
    for (uint32_t e = 0; e < XS_E; e++) { x[e] = lo + e < m ? v[lo + e] : 0u; sum += x[e]; }
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < XS_T; d <<= 1) {
        const uint32_t add = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    uint32_t run = a.xsums[(size_t)s * a.nxs + blockIdx.x] + part[tid] - sum;
#pragma unroll
    for (uint32_t e = 0; e < XS_E; e++) { if (lo + e < m) v[lo + e] = run; run += x[e]; }
}
ZK_KERNEL void lpb_compact_kernel(LpbArgs a) {
    const uint32_t l = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < a.u && a.flags[(size_t)(2 * l + 1) * a.u + t]) a.leftover[(size_t)l * a.u + a.ranks[(size_t)(2 * l + 1) * a.u + t]] = t;
}
ZK_KERNEL void lpb_assemble_kernel(LpbArgs a, const uint32_t* idx_sorted) {
    const uint32_t l = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x, u = a.u;
    if (i >= a.n) return;
    void* out_in = a.outs[2 * l];
    void* out_tab = a.outs[2 * l + 1];
    if (i >= u) {
        store_u256(out_in, i, load_u256(a.blind, (size_t)(2 * l) * a.nb + (i - u)));
        store_u256(out_tab, i, load_u256(a.blind, (size_t)(2 * l + 1) * a.nb + (i - u)));
        return;
    }
    const uint32_t* in_idx = idx_sorted + (size_t)l * u;
    const uint32_t* tab_idx = idx_sorted + (size_t)a.tabcol[l] * u;
    const u256 v = load_u256(a.cols[2 * l], in_idx[i]);
    store_u256(out_in, i, v);
    if (!a.flags[(size_t)(2 * l) * u + i]) store_u256(out_tab, i, v);
    else {
        const uint32_t n_rep = a.scal[l * 16 + 2];
        store_u256(out_tab, i, load_u256(a.cols[2 * l + 1], tab_idx[a.leftover[(size_t)l * u + (n_rep - 1 - a.ranks[(size_t)(2 * l) * u + i])]]));
    }
}

int lookup_permute_batch(zk_ctx* ctx, const void* const* d_inputs, const void* const* d_tables, size_t count, uint32_t k, uint32_t blinding_factors,
                         const void* h_blind_inputs, const void* h_blind_tables, void* const* d_out_inputs, void* const* d_out_tables) {
    if (count == 0) return ZK_OK;
    if (!d_inputs || !d_tables || !d_out_inputs || !d_out_tables || !h_blind_inputs || !h_blind_tables)
        return ctx->fail(ZK_ERR_ARG, "zk_lookup_permute_batch_dev: null argument");
    if (k < 1 || k > 26 || count > 4096) return ctx->fail(ZK_ERR_ARG, "zk_lookup_permute_batch_dev: k = %u / count = %zu out of range", k, count);
    const uint32_t n = 1u << k, nb = blinding_factors + 1;
    if (nb >= n) return ctx->fail(ZK_ERR_ARG, "zk_lookup_permute_batch_dev: blinding_factors too large");
    for (size_t l = 0; l < count; l++)
        if (!d_inputs[l] || !d_tables[l] || !d_out_inputs[l] || !d_out_tables[l]) return ctx->fail(ZK_ERR_ARG, "zk_lookup_permute_batch_dev: null column %zu", l);
    const uint32_t u = n - nb, S = 2 * (uint32_t)count;
    // sorted columns: every input, every DISTINCT table pointer
    std::vector<const void*> scols(d_inputs, d_inputs + count);
    std::vector<uint32_t> tabcol(count);
    for (size_t l = 0; l < count; l++) {
        size_t c = count;
        while (c < scols.size() && scols[c] != d_tables[l]) c++;
        if (c == scols.size()) scols.push_back(d_tables[l]);
        tabcol[l] = (uint32_t)c;
    }
    const uint32_t C = (uint32_t)scols.size();
    const uint32_t nwg = (u + FS_TILE - 1) / FS_TILE, nxs = (u + XS_TILE - 1) / XS_TILE + 1;
    // workspace carve-up (32-byte aligned pieces first)
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_canon = take((size_t)C * u * 32), o_blind = take((size_t)S * nb * 32), o_ka = take((size_t)C * u * 8), o_kb = take((size_t)C * u * 8),
                 o_ia = take((size_t)C * u * 4), o_ib = take((size_t)C * u * 4), o_fl = take((size_t)S * u * 4), o_rk = take((size_t)S * u * 4),
                 o_lo = take((size_t)count * u * 4), o_gh = take((size_t)C * nwg * 256 * 4), o_xs = take((size_t)S * nxs * 4), o_sc = take((size_t)count * 64),
                 o_cs = take((size_t)C * 64), o_sh = take((size_t)C * 4), o_tc = take((size_t)count * 4), o_cp = take((size_t)S * sizeof(void*)),
                 o_op = take((size_t)S * sizeof(void*)), o_sp = take((size_t)C * sizeof(void*));
    ZK_HIP(ctx->ws_tmp.ensure(off + 256));
    char* base = (char*)ctx->ws_tmp.p;
    hipStream_t st = ctx->stream;
    std::vector<const void*> cols(S);
    std::vector<void*> outs(S);
    std::vector<unsigned char> blind((size_t)S * nb * 32);
    for (size_t l = 0; l < count; l++) {
        cols[2 * l] = d_inputs[l]; cols[2 * l + 1] = d_tables[l];
        outs[2 * l] = d_out_inputs[l]; outs[2 * l + 1] = d_out_tables[l];
        memcpy(&blind[(2 * l) * (size_t)nb * 32], (const char*)h_blind_inputs + l * (size_t)nb * 32, (size_t)nb * 32);
        memcpy(&blind[(2 * l + 1) * (size_t)nb * 32], (const char*)h_blind_tables + l * (size_t)nb * 32, (size_t)nb * 32);
    }
    ZK_HIP(hipMemcpyAsync(base + o_cp, cols.data(), S * sizeof(void*), hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(base + o_op, outs.data(), S * sizeof(void*), hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(base + o_sp, scols.data(), C * sizeof(void*), hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(base + o_tc, tabcol.data(), count * 4, hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(base + o_blind, blind.data(), blind.size(), hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemsetAsync(base + o_sc, 0, count * 64, st));
    ZK_HIP(hipMemsetAsync(base + o_cs, 0, (size_t)C * 64, st));
    LpbArgs a;
    memset(&a, 0, sizeof a);
    a.cols = (const void* const*)(base + o_cp); a.outs = (void* const*)(base + o_op); a.scols = (const void* const*)(base + o_sp);
    a.tabcol = (const uint32_t*)(base + o_tc);
    a.u = u; a.n = n; a.nb = nb; a.S = S; a.C = C; a.nwg = nwg; a.nxs = nxs;
    a.canon = base + o_canon; a.key_a = (uint2*)(base + o_ka); a.key_b = (uint2*)(base + o_kb); a.idx_a = (uint32_t*)(base + o_ia); a.idx_b = (uint32_t*)(base + o_ib);
    a.flags = (uint32_t*)(base + o_fl); a.ranks = (uint32_t*)(base + o_rk); a.leftover = (uint32_t*)(base + o_lo); a.ghist = (uint32_t*)(base + o_gh);
    a.xsums = (uint32_t*)(base + o_xs); a.scal = (uint32_t*)(base + o_sc); a.cscal = (uint32_t*)(base + o_cs); a.shifts = (const uint32_t*)(base + o_sh);
    a.blind = base + o_blind;
    const int blk = ctx->tune.vec_block;
    const uint32_t g = (u + blk - 1) / blk;
    ZK_LAUNCH(lpb_canon_kernel, dim3(g, C), blk, 0, st, a);
    ZK_CHECK_LAUNCH();
    std::vector<uint32_t> sc((size_t)count * 16), csc((size_t)C * 16);
    ZK_HIP(hipMemcpyAsync(csc.data(), a.cscal, (size_t)C * 64, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    std::vector<uint32_t> shifts(C);
    uint32_t max_bits = 1;
    bool any_shift = false;
    for (uint32_t c = 0; c < C; c++) {            // each sorted column takes the 64-bit window that ends at ITS highest bit in use
        const uint32_t* om = &csc[(size_t)c * 16];
        uint32_t nbits = 1;
        for (int wd = 7; wd >= 0; wd--)
            if (om[wd]) { uint32_t top = 31; while (!((om[wd] >> top) & 1)) top--; nbits = (uint32_t)wd * 32 + top + 1; break; }
        shifts[c] = nbits > 64 ? nbits - 64 : 0;
        any_shift |= shifts[c] != 0;
        max_bits = std::max(max_bits, std::min(nbits, 64u));
    }
    ZK_HIP(hipMemcpyAsync(base + o_sh, shifts.data(), (size_t)C * 4, hipMemcpyHostToDevice, st));
    uint32_t* sorted = a.idx_a;                                         // [C][u]: the sorted order of every column ends up here
    const bool use_order_check = any_shift;

    // LSD radix sort of a COMPACT list of columns over a sequence of 64-bit windows ("stages", low to high): stage_shift[t][f] is the bit position of
    // column fcols[f]'s window in stage t (255 = constant key: nothing to sort there), stage_bits[t] the widest window of the stage.  Results go to
    // sorted[fcols[f]].  Workspace: ws_mid.
    auto compact_sort = [&](const std::vector<uint32_t>& fcols, const std::vector<std::vector<uint32_t>>& stage_shift, const std::vector<uint32_t>& stage_bits) -> int {
        const uint32_t Fn = (uint32_t)fcols.size();
        if (!Fn) return ZK_OK;
        const size_t per = (size_t)Fn * u;
        size_t o2 = 0;
        auto take2 = [&](size_t bytes) { size_t o = o2; o2 += (bytes + 255) & ~(size_t)255; return o; };
        const size_t q_ka = take2(per * 8), q_kb = take2(per * 8), q_ia = take2(per * 4), q_ib = take2(per * 4), q_gh = take2((size_t)Fn * nwg * 256 * 4),
                     q_fc = take2((size_t)Fn * 4), q_fs = take2((size_t)Fn * 4 * stage_shift.size());
        ZK_HIP(ctx->ws_mid.ensure(o2 + 256));
        char* b2 = (char*)ctx->ws_mid.p;
        uint2 *rk_in = (uint2*)(b2 + q_ka), *rk_out = (uint2*)(b2 + q_kb);
        uint32_t *ri_in = (uint32_t*)(b2 + q_ia), *ri_out = (uint32_t*)(b2 + q_ib), *rgh = (uint32_t*)(b2 + q_gh);
        ZK_HIP(hipMemcpyAsync(b2 + q_fc, fcols.data(), (size_t)Fn * 4, hipMemcpyHostToDevice, st));
        for (size_t t = 0; t < stage_shift.size(); t++)
            ZK_HIP(hipMemcpyAsync(b2 + q_fs + t * Fn * 4, stage_shift[t].data(), (size_t)Fn * 4, hipMemcpyHostToDevice, st));
        const uint32_t* order = nullptr;                                   // previous stage's order (compact); identity before the first
        for (size_t t = 0; t < stage_shift.size(); t++) {
            ZK_LAUNCH(lpb_rekey_kernel, dim3(g, Fn), blk, 0, st, a, (const uint32_t*)(b2 + q_fc), (const uint32_t*)(b2 + q_fs + t * Fn * 4), order, rk_in, ri_in);
            ZK_CHECK_LAUNCH();
            const uint32_t ps = (stage_bits[t] + 7) / 8;
            for (uint32_t p = 0; p < ps; p++) {
                ZK_LAUNCH(lpb_hist_kernel, dim3(nwg, Fn), FS_T, 0, st, (const uint2*)rk_in, u, p, rgh, nwg);
                ZK_CHECK_LAUNCH();
                ZK_LAUNCH(lpb_offsets_kernel, Fn, 256, 0, st, rgh, nwg);
                ZK_CHECK_LAUNCH();
                ZK_LAUNCH(lpb_scatter_kernel, dim3(nwg, Fn), FS_T, 0, st, (const uint2*)rk_in, (const uint32_t*)ri_in, rk_out, ri_out, u, p, (const uint32_t*)rgh, nwg);
                ZK_CHECK_LAUNCH();
                std::swap(rk_in, rk_out);
                std::swap(ri_in, ri_out);
            }
            // the next rekey reads the order from the buffers just written and writes keys / order into the OTHER pair
            std::swap(rk_in, rk_out);
            std::swap(ri_in, ri_out);
            order = ri_out;
        }
        for (uint32_t f = 0; f < Fn; f++)
            ZK_HIP(hipMemcpyAsync(sorted + (size_t)fcols[f] * u, order + (size_t)f * u, (size_t)u * 4, hipMemcpyDeviceToDevice, st));
        return ZK_OK;
    };
    // stages of a column whose window-tied rows differ inside bits [lo, hi]: 64-bit windows covering that range, low to high
    auto low_stages = [&](const std::vector<uint32_t>& fcols, const std::vector<uint32_t>& lo_bit, const std::vector<uint32_t>& hi_bit,
                          std::vector<std::vector<uint32_t>>& stage_shift, std::vector<uint32_t>& stage_bits) {
        const uint32_t Fn = (uint32_t)fcols.size();
        uint32_t n_low = 0;
        for (uint32_t f = 0; f < Fn; f++) if (lo_bit[f] < 256) n_low = std::max(n_low, (hi_bit[f] - lo_bit[f]) / 64 + 1);
        for (uint32_t t = 0; t < n_low; t++) {
            std::vector<uint32_t> sh(Fn);
            uint32_t bits = 1;
            for (uint32_t f = 0; f < Fn; f++) {
                const uint32_t st_lo = lo_bit[f] == 256 ? 256 : lo_bit[f] + 64 * t;
                if (st_lo > hi_bit[f] || st_lo >= 256) { sh[f] = 255; continue; }               // bit 255 of a canonical value is always 0: constant key
                sh[f] = st_lo;
                bits = std::max(bits, std::min(64u, hi_bit[f] - st_lo + 1));
            }
            stage_shift.push_back(sh); stage_bits.push_back(bits);
        }
        std::vector<uint32_t> sh(Fn);
        for (uint32_t f = 0; f < Fn; f++) sh[f] = shifts[fcols[f]];
        stage_shift.push_back(sh); stage_bits.push_back(max_bits);
    };
    // tie mask of the flagged columns (cscal[c][8] != 0) in the current order -> (lo, hi) differing bit per column of `cols`
    auto tie_ranges = [&](const std::vector<uint32_t>& cols_, std::vector<uint32_t>& lo_bit, std::vector<uint32_t>& hi_bit) -> int {
        for (uint32_t c : cols_) ZK_HIP(hipMemsetAsync((char*)a.cscal + (size_t)c * 64, 0, 32, st));
        ZK_LAUNCH(lpb_tiemask_kernel, dim3(g, C), blk, 0, st, a, (const uint32_t*)sorted); ZK_CHECK_LAUNCH();
        ZK_HIP(hipMemcpyAsync(csc.data(), a.cscal, (size_t)C * 64, hipMemcpyDeviceToHost, st));
        ZK_HIP(hipStreamSynchronize(st));
        lo_bit.assign(cols_.size(), 256); hi_bit.assign(cols_.size(), 0);
        for (size_t f = 0; f < cols_.size(); f++) {
            const uint32_t* m = &csc[(size_t)cols_[f] * 16];
            for (uint32_t b = 0; b < 256; b++) if ((m[b >> 5] >> (b & 31)) & 1) { if (lo_bit[f] == 256) lo_bit[f] = b; hi_bit[f] = b; }
        }
        return ZK_OK;
    };

    // Columns that tied on their window in the PREVIOUS call of the same shape (same proving key: the tie structure of a lookup comes from its expressions,
    // e.g. theta-compressed tuples that differ in their last expression) go straight to the two-stage sort; the order check below still guards them.
    const uint64_t hint_key = ((uint64_t)count << 40) ^ ((uint64_t)C << 20) ^ k;
    std::vector<uint32_t>& hint = ctx->lookup_tie_hint[hint_key];         // per column: 0 = none, else 1 + highest differing bit seen
    if (hint.size() != C) hint.assign(C, 0);
    std::vector<uint32_t> plain, hinted;
    for (uint32_t c = 0; c < C; c++) (hint[c] && shifts[c] && !ctx->tune.lookup_force_generic_sort ? hinted : plain).push_back(c);
    {
        std::vector<std::vector<uint32_t>> ss(1, std::vector<uint32_t>(plain.size()));
        for (size_t f = 0; f < plain.size(); f++) ss[0][f] = shifts[plain[f]];
        int rc = compact_sort(plain, ss, {max_bits});
        if (rc) return rc;
    }
    if (!hinted.empty()) {
        std::vector<uint32_t> lo_bit(hinted.size(), 0), hi_bit(hinted.size());
        for (size_t f = 0; f < hinted.size(); f++) hi_bit[f] = std::min(shifts[hinted[f]] - 1, hint[hinted[f]] - 1 + 16);   // what was seen, plus room for longer carries
        std::vector<std::vector<uint32_t>> ss; std::vector<uint32_t> sb;
        low_stages(hinted, lo_bit, hi_bit, ss, sb);
        int rc = compact_sort(hinted, ss, sb);
        if (rc) return rc;
        ctx->last_ms["lookup_hinted_sorts"] += (double)hinted.size();
    }
    if (use_order_check) {
        ZK_LAUNCH(lpb_check_kernel, dim3(g, C), blk, 0, st, a, (const uint32_t*)sorted); ZK_CHECK_LAUNCH();
        ZK_HIP(hipMemcpyAsync(csc.data(), a.cscal, (size_t)C * 64, hipMemcpyDeviceToHost, st));
        ZK_HIP(hipStreamSynchronize(st));
        std::vector<uint32_t> fcols;
        for (uint32_t c = 0; c < C; c++) if (csc[(size_t)c * 16 + 8]) fcols.push_back(c);
        if (!fcols.empty() && !ctx->tune.lookup_force_generic_sort) {
            // window ties (first seen, or a hinted column whose ties reached above the hinted range): which bits do tied rows differ in?
            std::vector<uint32_t> lo_bit, hi_bit;
            int rc = tie_ranges(fcols, lo_bit, hi_bit);
            if (rc) return rc;
            std::vector<std::vector<uint32_t>> ss; std::vector<uint32_t> sb;
            low_stages(fcols, lo_bit, hi_bit, ss, sb);
            rc = compact_sort(fcols, ss, sb);
            if (rc) return rc;
            for (size_t f = 0; f < fcols.size(); f++) {
                hint[fcols[f]] = hi_bit[f] + 1;
                ZK_HIP(hipMemsetAsync((char*)a.cscal + (size_t)fcols[f] * 64 + 32, 0, 4, st));     // clear the violation count, then check again
            }
            ZK_LAUNCH(lpb_check_kernel, dim3(g, C), blk, 0, st, a, (const uint32_t*)sorted); ZK_CHECK_LAUNCH();
            ctx->last_ms["lookup_refined_sorts"] += (double)fcols.size();
        }
    }
    ZK_LAUNCH(lpb_fill_unconsumed_kernel, dim3(g, (uint32_t)count), blk, 0, st, a);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(lpb_mark_kernel, dim3(g, (uint32_t)count), blk, 0, st, a, sorted);
    ZK_CHECK_LAUNCH();
    ZK_HIP(hipMemcpyAsync(a.ranks, a.flags, (size_t)S * u * 4, hipMemcpyDeviceToDevice, st));
    ZK_LAUNCH(lpb_tile_sum_kernel, dim3(nxs - 1, S), XS_T, 0, st, a);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(lpb_sum_scan_kernel, S, lp_scan_threads(nxs - 1), 0, st, a);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(lpb_apply_kernel, dim3(nxs - 1, S), XS_T, 0, st, a);
    ZK_CHECK_LAUNCH();
    ZK_HIP(hipMemcpyAsync(sc.data(), a.scal, count * 64, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipMemcpyAsync(csc.data(), a.cscal, (size_t)C * 64, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    std::vector<size_t> redo;
    for (size_t l = 0; l < count; l++) {
        const uint32_t* h = &sc[l * 16];
        if (csc[l * 16 + 8] || csc[(size_t)tabcol[l] * 16 + 8] || ctx->tune.lookup_force_generic_sort) { redo.push_back(l); continue; }   // window ties: every-digit sort for this lookup
        if (h[1]) return ctx->fail(ZK_ERR_ARG, "zk_lookup_permute_dev: %u input value(s) of lookup %zu are not in the table (halo2: Error::ConstraintSystemFailure)", h[1], l);
        if (h[2] != h[3]) return ctx->fail(ZK_ERR_ARG, "zk_lookup_permute_dev: internal count mismatch in lookup %zu (%u repeated rows, %u leftover table values)", l, h[2], h[3]);
    }
    ZK_LAUNCH(lpb_compact_kernel, dim3(g, (uint32_t)count), blk, 0, st, a);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(lpb_assemble_kernel, dim3((n + blk - 1) / blk, (uint32_t)count), blk, 0, st, a, sorted);
    ZK_CHECK_LAUNCH();
    ZK_HIP(hipStreamSynchronize(st));
    ctx->last_ms["lookup_sorts"] += (double)count;                     // statistics (zk_timing_get): lookups sorted, and how many took the every-digit path
    ctx->last_ms["lookup_generic_sorts"] += (double)redo.size();
    for (size_t l : redo) {
        int rc = lookup_permute_one(ctx, d_inputs[l], d_tables[l], k, blinding_factors, (const char*)h_blind_inputs + l * (size_t)nb * 32,
                                    (const char*)h_blind_tables + l * (size_t)nb * 32, d_out_inputs[l], d_out_tables[l]);
        if (rc) return rc;
    }
    return ZK_OK;
}
int lookup_permute(zk_ctx* ctx, const void* d_input, const void* d_table, uint32_t k, uint32_t blinding_factors, const void* h_blind_input,
                   const void* h_blind_table, void* d_out_input, void* d_out_table) {
    if (!d_input || !d_table || !d_out_input || !d_out_table || !h_blind_input || !h_blind_table)
        return ctx->fail(ZK_ERR_ARG, "zk_lookup_permute_dev: null argument");
    const void* ins[1] = {d_input}; const void* tabs[1] = {d_table};
    void* oi[1] = {d_out_input}; void* ot[1] = {d_out_table};
    return lookup_permute_batch(ctx, ins, tabs, 1, k, blinding_factors, h_blind_input, h_blind_table, oi, ot);
}

}  // namespace zk
