// BN254 G1 multi-scalar multiplication for MI355X (gfx950).
//
// Replaces halo2_proofs (zkwebauthn @ c254c75, Cargo.lock:1314-1327) src/arithmetic.rs
// best_multiexp — the CPU algorithm there is a per-thread chunked Pippenger with unsigned
// c = ceil(ln n)-bit windows.  This is NOT that algorithm moved to a GPU; the
// result (a G1 point) is canonical, so the design is free and is built for this chip:
//
//  * KZG bases are fixed per SRS, and the GPU has 288 GB of HBM, so registration expands the table
//    to all window multiples T[j][i] = 2^(c*j) * P_i.  Every (scalar, window) pair then lands in ONE
//    shared set of 2^(c-1) buckets: no per-window bucket reduction, no doublings at the end.
//  * signed c-bit digits (c <= 16) halve the bucket count: the whole bucket histogram (<= 128 KiB)
//    fits the 160 KiB LDS of one CU, so the counting sort that groups pairs by bucket runs on LDS
//    atomics and touches HBM only for the scalars and the sorted 4-byte point references.
//  * buckets are cut into sub-buckets of <= L pairs, one thread each (XYZZ mixed adds, 8M + 2S),
//    so degenerate scalar columns (all ones, bytes, ...) cannot serialise on one thread; sub-buckets
//    are merged by fan-in rounds.
//  * bucket reduction sum_w w * B_w is done by weight bits: C_t = sum_{w: bit t} B_w are plain tree
//    sums (depth log instead of the serial running sum), and the host folds sum_t 2^t C_t.
//
// Integer-ALU bound (v_mad_u64_u32), not HBM bound: see DESIGN.md for the roofline accounting.
#include "ctx.h"
#include <algorithm>
#include <math.h>

namespace zk {

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
// Signed-digit recoding of a canonical 254-bit scalar: digits d_j in [-(2^(c-1)-1), 2^(c-1)],
// sum_j d_j 2^(c j) = s.  f(j, magnitude >= 1, negative).
template <class F>
ZK_HD void for_each_digit(u256 s, int c, int W, F&& f) {
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    uint32_t carry = 0;
    for (int j = 0; j < W; j++) {
        uint32_t d = (s.v[0] & mask) + carry;
#pragma unroll
        for (int i = 0; i < 7; i++) s.v[i] = (s.v[i] >> c) | (s.v[i + 1] << (32 - c));
        s.v[7] >>= c;
        if (d > half) {
            carry = 1;
            uint32_t m = (1u << c) - d;
            if (m) f(j, m, true);
        } else {
            carry = 0;
            if (d) f(j, d, false);
        }
    }
}

// the same walk with the window index as a compile-time constant (callers index register arrays with it); windows >= W are skipped
template <int J> struct WinIdx { static constexpr int value = J; };
template <int WMAX, int J = 0, class F>
ZK_HD void for_each_digit_static(u256& s, int c, int W, F&& f, uint32_t carry = 0) {
    if constexpr (J < WMAX) {
        if (J < W) {
            const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
            uint32_t d = (s.v[0] & mask) + carry;
#pragma unroll
            for (int i = 0; i < 7; i++) s.v[i] = (s.v[i] >> c) | (s.v[i + 1] << (32 - c));
            s.v[7] >>= c;
            uint32_t cy = 0;
            if (d > half) {
                cy = 1;
                uint32_t m = (1u << c) - d;
                if (m) f(WinIdx<J>{}, m, true);
            } else if (d) f(WinIdx<J>{}, d, false);
            for_each_digit_static<WMAX, J + 1>(s, c, W, f, cy);
        }
    }
}

// c = 16, W = 16 (every table of 2^17 points and more): the digits are the half-limbs, read in place — no 256-bit shift per window
template <class F>
ZK_HD void for_each_digit16(const u256& s, F&& f) {
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t d = ((s.v[j >> 1] >> (16 * (j & 1))) & 0xffffu) + carry;
        if (d > 0x8000u) {
            carry = 1;
            const uint32_t m = 0x10000u - d;
            if (m) f(j, m, true);
        } else {
            carry = 0;
            if (d) f(j, d, false);
        }
    }
}
template <int J = 0, class F>
ZK_HD void for_each_digit16_static(const u256& s, F&& f, uint32_t carry = 0) {
    if constexpr (J < 16) {
        const uint32_t d = ((s.v[J >> 1] >> (16 * (J & 1))) & 0xffffu) + carry;
        uint32_t cy = 0;
        if (d > 0x8000u) {
            cy = 1;
            const uint32_t m = 0x10000u - d;
            if (m) f(WinIdx<J>{}, m, true);
        } else if (d) f(WinIdx<J>{}, d, false);
        for_each_digit16_static<J + 1>(s, f, cy);
    }
}

// largest b in [0, B) with arr[b] <= j  (arr is a non-decreasing exclusive scan, arr[0] = 0)
ZK_HD uint32_t find_segment(const uint32_t* arr, uint32_t B, uint32_t j) {
    uint32_t lo = 0, hi = B;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (arr[mid] <= j) lo = mid; else hi = mid;
    }
    return lo;
}

ZK_HD uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------------
// launch plan shared by the kernels of one (batched) MSM call.  blockIdx.y = column of the batch.
// ------------------------------------------------------------------------------------------------
constexpr int MSM_MAX_LEVELS = 12;
struct MsmPlan {
    uint32_t n, n_table, B, L, M, Mlog, R, nb;   // R = merge levels above level 0; M = 2^Mlog
    int c, W;
    const void* const* scalars;            // device array: nb column pointers
    const void* table;
    const void* const* col_tables;         // device array of nb table pointers (run columns read the prefix-sum twin), or null: every column reads `table`
    uint32_t* small;                       // per column: hist[B] off[B+1] cursor[B] suboff[R+1][B+1] info[4]
    uint32_t small_stride, o_off, o_cursor, o_suboff, o_info, o_tiles, o_fulloff, o_remorder, o_remhist, o_remstart, o_remcursor;
    uint32_t* sorted;                      // per column: pairs_max references
    uint64_t sorted_stride;
    uint32_t low_bits, nbin, pch, bch, o_bintot, o_bincur, o_binbase, o_chunkbase;   // two-level sort: bin = bucket >> low_bits; pch = scalars per partition workgroup; bch = entries per second-level workgroup
    uint32_t* mid_ref;                     // per column (stride sorted_stride): references grouped by bin ...
    uint16_t* mid_low;                     // ... and the low bits of their bucket
    void* sub[2];                          // level r lives in sub[r & 1]
    uint64_t sub_stride[2];                // entries per column
    void* cls[2];
    uint32_t lo_bits;                      // bucket reduction: weight = hi * 2^lo_bits + lo
};
ZK_HD uint32_t* plan_small(const MsmPlan& p, uint32_t col) { return p.small + (size_t)col * p.small_stride; }
ZK_HD const uint32_t* plan_suboff(const MsmPlan& p, uint32_t col, uint32_t level) { return plan_small(p, col) + p.o_suboff + (size_t)level * (p.B + 1); }
ZK_HD uint32_t plan_eff_levels(const MsmPlan& p, uint32_t max_s) {  // merge rounds that actually run
    uint32_t pw = 1, r = 0;
    while (pw < max_s && r < p.R) { pw *= p.M; r++; }
    return r;
}

// ------------------------------------------------------------------------------------------------
// grouping the (scalar, window) pairs by bucket: a two-level counting sort.
// (Written to replace the one-level sort below, whose workgroups put ~1 pair into each of their 32768 buckets at n = 2^19 — millions of
// global atomics and isolated 4-byte stores per column.  Measured, it does not: see the note at the one-level kernels.)
//        bin_hist    bucket >> low_bits (<= MSM_MAX_BINS bins): per-column bin totals                       (LDS counters, one global atomic per bin and workgroup)
//        bin_base    one workgroup per column: exclusive scan of the bin totals (every later workgroup reads its bases instead of scanning)
//        partition   a workgroup takes pch scalars, groups their pairs by bin in LDS and writes each bin's run with coalesced stores
//                    (reference + low bucket bits) at a range reserved with ONE global atomic per bin
//        bsort_count / bsort_place   a bin's run is cut into chunks of <= bch entries, one workgroup each, <= MSM_MAX_LOW bucket counters in LDS: count adds the
//                    chunk's bucket sizes to the histogram the scans below start from; place (after the scans) reserves and fills the chunk's slots of `sorted`.
// reference = (negative << 31) | (window * n_table + scalar index)
// It is also the ONLY sort for windows wider than 16 bits (c = 17 .. 22: 2^(c-1) counters do not fit one CU's LDS): 2^msm_wide_bins_log bins (512) of 2^7 .. 2^12 buckets.
// Wide windows pay where n is large against the bucket count — 2^24 scalars at c = 20 are 13 windows instead of 16 (-19 % additions) for 2^19 buckets (pick_c).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t MSM_MAX_BINS = 2048, MSM_MAX_LOW = 8192;
constexpr int MSM_MAX_C = 22;

// inclusive scan of v[0..N) in LDS (N <= MSM_MAX_BINS), any block size; tmp: N words
__device__ __forceinline__ void block_incscan(uint32_t* v, uint32_t* tmp, uint32_t N) {
    for (uint32_t d = 1; d < N; d <<= 1) {
        for (uint32_t t = threadIdx.x; t < N; t += blockDim.x) tmp[t] = v[t] + (t >= d ? v[t - d] : 0u);
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < N; t += blockDim.x) v[t] = tmp[t];
        __syncthreads();
    }
}

ZK_KERNEL void msm_bin_hist_kernel(MsmPlan p) {
    __shared__ uint32_t lh[MSM_MAX_BINS];
    const uint32_t col = blockIdx.y;
    const void* scalars = p.scalars[col];
    for (uint32_t b = threadIdx.x; b < p.nbin; b += blockDim.x) lh[b] = 0;
    __syncthreads();
    const uint32_t chunk = ceil_div(p.n, gridDim.x);
    const uint32_t lo = blockIdx.x * chunk;
    const uint32_t hi = lo + chunk < p.n ? lo + chunk : p.n;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        u256 s = Fr::from_mont(load_u256(scalars, i));
        for_each_digit(s, p.c, p.W, [&](int, uint32_t mag, bool) { atomicAdd(&lh[(mag - 1) >> p.low_bits], 1u); });
    }
    __syncthreads();
    uint32_t* tot = plan_small(p, col) + p.o_bintot;
    for (uint32_t b = threadIdx.x; b < p.nbin; b += blockDim.x) {
        uint32_t v = lh[b];
        if (v) atomicAdd(&tot[b], v);
    }
}

// exclusive scans of a column's bin totals (one workgroup per column): binbase[b] = pairs in bins below b, chunkbase[b] = chunks of <= bch pairs in bins below b
// (a bin is sorted by ceil(size / bch) workgroups — the short top window of a scalar puts 1/W of all pairs into the few lowest bins, so bins are NOT of one size)
ZK_KERNEL void msm_bin_base_kernel(MsmPlan p) {
    __shared__ uint32_t v[MSM_MAX_BINS], w[MSM_MAX_BINS], tmp[MSM_MAX_BINS];
    uint32_t* sm = plan_small(p, blockIdx.x);
    for (uint32_t b = threadIdx.x; b < p.nbin; b += blockDim.x) { const uint32_t t = sm[p.o_bintot + b]; v[b] = t; w[b] = ceil_div(t, p.bch); }
    __syncthreads();
    block_incscan(v, tmp, p.nbin);
    block_incscan(w, tmp, p.nbin);
    for (uint32_t b = threadIdx.x; b < p.nbin; b += blockDim.x) { sm[p.o_binbase + b + 1] = v[b]; sm[p.o_chunkbase + b + 1] = w[b]; }
    if (threadIdx.x == 0) { sm[p.o_binbase] = 0; sm[p.o_chunkbase] = 0; }
}

// Every thread keeps its (<= PART_SPT) canonical scalars in registers between the counting and the placing sweep (the sweeps are latency chains:
// load -> Montgomery reduction -> digits -> LDS atomics).  LDS: the staging area (a reference and a bucket index per pair) and four words per bin.
constexpr uint32_t PART_SPT = 2;
ZK_KERNEL void msm_partition_kernel(MsmPlan p) {
    ZK_DYN_SHARED(uint32_t, st_ref);                       // [pch * W] references | [pch * W] bucket indices | cnt, start, dst, tmp [nbin]
    const uint32_t col = blockIdx.y, nbin = p.nbin, cap = p.pch * (uint32_t)p.W;
    uint32_t* st_key = st_ref + cap;
    uint32_t* cnt = st_key + cap;
    uint32_t* start = cnt + nbin;
    uint32_t* dst = start + nbin;
    uint32_t* tmp = dst + nbin;
    const void* scalars = p.scalars[col];
    uint32_t* sm = plan_small(p, col);
    for (uint32_t b = threadIdx.x; b < nbin; b += blockDim.x) cnt[b] = 0;
    const uint32_t lo = blockIdx.x * p.pch;
    const uint32_t hi = lo + p.pch < p.n ? lo + p.pch : p.n;
    u256 sc[PART_SPT];
#pragma unroll
    for (uint32_t u = 0; u < PART_SPT; u++) {              // pch <= PART_SPT * blockDim (host)
        const uint32_t i = lo + threadIdx.x + u * blockDim.x;
        if (i < hi) sc[u] = load_u256(scalars, i);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < PART_SPT; u++) {
        const uint32_t i = lo + threadIdx.x + u * blockDim.x;
        if (i < hi) {
            sc[u] = Fr::from_mont(sc[u]);
            for_each_digit(sc[u], p.c, p.W, [&](int, uint32_t mag, bool) { atomicAdd(&cnt[(mag - 1) >> p.low_bits], 1u); });
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbin; b += blockDim.x) start[b] = cnt[b];
    __syncthreads();
    block_incscan(start, tmp, nbin);                       // start[b] = end of bin b inside this workgroup's staging area
    for (uint32_t b = threadIdx.x; b < nbin; b += blockDim.x) {
        const uint32_t c0 = cnt[b];
        dst[b] = sm[p.o_binbase + b] + (c0 ? atomicAdd(&sm[p.o_bincur + b], c0) : 0u);
        tmp[b] = start[b] - c0;                            // cursor
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbin; b += blockDim.x) start[b] = tmp[b];
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < PART_SPT; u++) {
        const uint32_t i = lo + threadIdx.x + u * blockDim.x;
        if (i < hi)
            for_each_digit(sc[u], p.c, p.W, [&](int j, uint32_t mag, bool neg) {
                const uint32_t pos = atomicAdd(&tmp[(mag - 1) >> p.low_bits], 1u);
                st_ref[pos] = (neg ? 0x80000000u : 0u) | ((uint32_t)j * p.n_table + i);
                st_key[pos] = mag - 1;
            });
    }
    __syncthreads();
    const uint32_t total = tmp[nbin - 1];                  // the last cursor ended at the number of staged pairs
    uint32_t* mref = p.mid_ref + (size_t)col * p.sorted_stride;
    uint16_t* mlow = p.mid_low + (size_t)col * p.sorted_stride;
    const uint32_t lmask = (1u << p.low_bits) - 1u;
    for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {   // neighbouring lanes -> neighbouring addresses of the same bin's run
        const uint32_t key = st_key[e], b = key >> p.low_bits;
        const uint32_t d = dst[b] + (e - start[b]);
        mref[d] = st_ref[e];
        mlow[d] = (uint16_t)(key & lmask);
    }
}

// The second level: workgroup t of a column takes chunk t - chunkbase[bin] of the bin it falls into (<= bch entries of the bin's run, all of them buckets of that bin),
// counts them per bucket in LDS (2^low_bits counters) and
//   count:  adds its counts to the column's histogram (the scans below turn it into bucket offsets and cursors, as for the one-level sort);
//   place:  reserves its entries' slots of every bucket with one global atomic per non-empty bucket, then walks the chunk again and stores each reference at
//           its bucket's running LDS cursor.
// Consecutive lanes read consecutive entries (4-byte references, 2-byte low bits).
constexpr uint32_t BS_U = 8;
__device__ __forceinline__ bool bsort_chunk(const MsmPlan& p, const uint32_t* sm, uint32_t& bin, uint32_t& lo, uint32_t& hi) {
    const uint32_t t = blockIdx.x;
    if (t >= sm[p.o_chunkbase + p.nbin]) return false;
    bin = find_segment(sm + p.o_chunkbase, p.nbin, t);
    lo = sm[p.o_binbase + bin] + (t - sm[p.o_chunkbase + bin]) * p.bch;
    const uint32_t end = sm[p.o_binbase + bin + 1];
    hi = lo + p.bch < end ? lo + p.bch : end;
    return true;
}
ZK_KERNEL void msm_bsort_count_kernel(MsmPlan p) {
    ZK_DYN_SHARED(uint32_t, cnt);                          // [2^low_bits]
    const uint32_t col = blockIdx.y, nlow = 1u << p.low_bits;
    uint32_t* sm = plan_small(p, col);
    uint32_t bin, lo, hi;
    if (!bsort_chunk(p, sm, bin, lo, hi)) return;          // (uniform: before any barrier)
    for (uint32_t b = threadIdx.x; b < nlow; b += blockDim.x) cnt[b] = 0;
    __syncthreads();
    const uint16_t* __restrict__ mlow = p.mid_low + (size_t)col * p.sorted_stride;
    for (uint32_t e0 = lo + threadIdx.x; e0 < hi; e0 += BS_U * blockDim.x) {            // BS_U loads in flight per lane
        uint32_t k[BS_U];
#pragma unroll
        for (uint32_t u = 0; u < BS_U; u++) { const uint32_t e = e0 + u * blockDim.x; k[u] = e < hi ? mlow[e] : 0xffffffffu; }
#pragma unroll
        for (uint32_t u = 0; u < BS_U; u++) if (k[u] != 0xffffffffu) atomicAdd(&cnt[k[u]], 1u);
    }
    __syncthreads();
    uint32_t* hist = sm + (size_t)bin * nlow;
    for (uint32_t b = threadIdx.x; b < nlow; b += blockDim.x) {
        const uint32_t v = cnt[b];
        if (v) atomicAdd(&hist[b], v);
    }
}
ZK_KERNEL void msm_bsort_place_kernel(MsmPlan p) {
    ZK_DYN_SHARED(uint32_t, cnt);                          // [2^low_bits]: counts, then the chunk's cursor per bucket
    const uint32_t col = blockIdx.y, nlow = 1u << p.low_bits;
    uint32_t* sm = plan_small(p, col);
    uint32_t bin, lo, hi;
    if (!bsort_chunk(p, sm, bin, lo, hi)) return;
    for (uint32_t b = threadIdx.x; b < nlow; b += blockDim.x) cnt[b] = 0;
    __syncthreads();
    const uint32_t* __restrict__ mref = p.mid_ref + (size_t)col * p.sorted_stride;
    const uint16_t* __restrict__ mlow = p.mid_low + (size_t)col * p.sorted_stride;
    for (uint32_t e0 = lo + threadIdx.x; e0 < hi; e0 += BS_U * blockDim.x) {            // BS_U loads in flight per lane
        uint32_t k[BS_U];
#pragma unroll
        for (uint32_t u = 0; u < BS_U; u++) { const uint32_t e = e0 + u * blockDim.x; k[u] = e < hi ? mlow[e] : 0xffffffffu; }
#pragma unroll
        for (uint32_t u = 0; u < BS_U; u++) if (k[u] != 0xffffffffu) atomicAdd(&cnt[k[u]], 1u);
    }
    __syncthreads();
    uint32_t* cursor = sm + p.o_cursor + (size_t)bin * nlow;
    for (uint32_t b0 = threadIdx.x; b0 < nlow; b0 += 4 * blockDim.x) {    // 4 reservations in flight per thread
        uint32_t v[4], r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t b = b0 + u * blockDim.x; v[u] = b < nlow ? cnt[b] : 0u; }
#pragma unroll
        for (int u = 0; u < 4; u++) r[u] = v[u] ? atomicAdd(&cursor[b0 + u * blockDim.x], v[u]) : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t b = b0 + u * blockDim.x; if (b < nlow) cnt[b] = r[u]; }
    }
    __syncthreads();
    uint32_t* __restrict__ sorted = p.sorted + (size_t)col * p.sorted_stride;
    for (uint32_t e0 = lo + threadIdx.x; e0 < hi; e0 += BS_U * blockDim.x) {            // (issuing the next round's loads before this round's stores measured slower)
        uint32_t k[BS_U], r[BS_U], d[BS_U];
#pragma unroll
        for (uint32_t u = 0; u < BS_U; u++) { const uint32_t e = e0 + u * blockDim.x; const bool in = e < hi; k[u] = in ? mlow[e] : 0xffffffffu; r[u] = in ? mref[e] : 0u; }
#pragma unroll
        for (uint32_t u = 0; u < BS_U; u++) d[u] = k[u] != 0xffffffffu ? atomicAdd(&cnt[k[u]], 1u) : 0u;
#pragma unroll
        for (uint32_t u = 0; u < BS_U; u++) if (k[u] != 0xffffffffu) sorted[d[u]] = r[u];
    }
}

// ---- one-level sort (the default) ---------------------------------------------------------------------------------------------------------
// Every workgroup keeps all 2^(c-1) bucket counters in LDS (128 KiB at c = 16): hist counts, the scans run, scatter re-counts its chunk,
// reserves a range per bucket with a global atomic and places the references.  scatter's counting atomicAdd already returns the pair's rank
// inside (workgroup, bucket); it stays in a register (16 bits) until the bucket's base is known, so placing needs no second LDS atomic.
// The rank registers bound a workgroup's chunk (<= SC_SPT scalars per thread, chunk * W <= 65536) and the window count (W <= SC_WMAX).
// Measured against the two-level sort above on MI355X (profiles/r01/run41_sort_crossover.txt, run41 prover probes): single columns 2^19 .. 2^24
// within 5 % of each other; batched columns (the prover's case) 4 % per proof in favour of this one — both end in 4-byte stores to 64
// different cache lines per wave, and the two-level sort pays for its coalesced intermediate pass on top.  So the two-level sort runs only
// where this one cannot (W > SC_WMAX: tiny inputs with narrow windows) or when forced (msm_two_level_sort = 1).
ZK_KERNEL void msm_hist_kernel(MsmPlan p) {
    ZK_DYN_SHARED(uint32_t, lh);
    const uint32_t B = p.B, col = blockIdx.y;
    const void* scalars = p.scalars[col];
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) lh[b] = 0;
    __syncthreads();
    const uint32_t chunk = ceil_div(p.n, gridDim.x);
    const uint32_t lo = blockIdx.x * chunk;
    const uint32_t hi = lo + chunk < p.n ? lo + chunk : p.n;
    const bool wide16 = p.c == 16 && p.W == 16;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        u256 s = Fr::from_mont(load_u256(scalars, i));
        if (wide16) for_each_digit16(s, [&](int, uint32_t mag, bool) { atomicAdd(&lh[mag - 1], 1u); });
        else for_each_digit(s, p.c, p.W, [&](int, uint32_t mag, bool) { atomicAdd(&lh[mag - 1], 1u); });
    }
    __syncthreads();
    uint32_t* ghist = plan_small(p, col);
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) {
        uint32_t v = lh[b];
        if (v) atomicAdd(&ghist[b], v);
    }
}


constexpr uint32_t SC_SPT = 4, SC_WMAX = 32;
ZK_KERNEL void msm_scatter_kernel(MsmPlan p) {
    ZK_DYN_SHARED(uint32_t, lh);
    const uint32_t B = p.B, col = blockIdx.y;
    const void* scalars = p.scalars[col];
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) lh[b] = 0;
    const bool wide16 = p.c == 16 && p.W == 16;
    const uint32_t chunk = ceil_div(p.n, gridDim.x);            // <= SC_SPT * blockDim and chunk * W <= 65536 (host): ranks fit 16 bits
    const uint32_t lo = blockIdx.x * chunk;
    const uint32_t hi = lo + chunk < p.n ? lo + chunk : p.n;
    u256 sc[SC_SPT];
    uint32_t rk[SC_SPT][SC_WMAX / 2] = {};                      // two 16-bit ranks per register
#pragma unroll
    for (uint32_t u = 0; u < SC_SPT; u++) {
        const uint32_t i = lo + threadIdx.x + u * blockDim.x;
        if (i < hi) sc[u] = load_u256(scalars, i);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < SC_SPT; u++) {
        const uint32_t i = lo + threadIdx.x + u * blockDim.x;
        if (i < hi) {
            sc[u] = Fr::from_mont(sc[u]);
            auto count = [&](auto jc, uint32_t mag, bool) {
                constexpr int j = decltype(jc)::value;
                const uint32_t r = atomicAdd(&lh[mag - 1], 1u);
                if (j & 1) rk[u][j / 2] |= r << 16; else rk[u][j / 2] = r;
            };
            if (wide16) for_each_digit16_static(sc[u], count);
            else { u256 t = sc[u]; for_each_digit_static<SC_WMAX>(t, p.c, p.W, count); }
        }
    }
    __syncthreads();
    uint32_t* cursor = plan_small(p, col) + p.o_cursor;
    for (uint32_t b0 = threadIdx.x; b0 < B; b0 += 8 * blockDim.x) {   // 8 reservations in flight per thread
        uint32_t v[8], r[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint32_t b = b0 + u * blockDim.x; v[u] = b < B ? lh[b] : 0u; }
#pragma unroll
        for (int u = 0; u < 8; u++) r[u] = v[u] ? atomicAdd(&cursor[b0 + u * blockDim.x], v[u]) : 0u;
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint32_t b = b0 + u * blockDim.x; if (b < B) lh[b] = r[u]; }
    }
    __syncthreads();
    uint32_t* sorted = p.sorted + (size_t)col * p.sorted_stride;
#pragma unroll
    for (uint32_t u = 0; u < SC_SPT; u++) {
        const uint32_t i = lo + threadIdx.x + u * blockDim.x;
        if (i < hi) {
            auto place = [&](auto jc, uint32_t mag, bool neg) {
                constexpr int j = decltype(jc)::value;
                const uint32_t r = (j & 1) ? rk[u][j / 2] >> 16 : rk[u][j / 2] & 0xffffu;
                sorted[lh[mag - 1] + r] = (neg ? 0x80000000u : 0u) | ((uint32_t)j * p.n_table + i);
            };
            if (wide16) for_each_digit16_static(sc[u], place);
            else for_each_digit_static<SC_WMAX>(sc[u], p.c, p.W, place);
        }
    }
}

// exclusive scans of the bucket sizes and of the entry counts of every merge level
// (level 0 = sub-buckets of <= L pairs, level r = ceil(level r-1 / M)), NV = R + 2 running sums in all:
//   off[b] = first sorted slot of bucket b, suboff[r][b] = first level-r entry of bucket b, info[0] = max level-0 entries of any bucket
// Three short launches over tiles of SC_T * SC_E buckets (a single 1024-thread workgroup walking 32 buckets per thread with
// strided 4-byte accesses took 0.23 ms per call — pure latency in front of every MSM):
//   tiles:    per-tile sums of the NV quantities (coalesced 16-byte reads, LDS tree)            -> tsum[tile][v]
//   tilesums: one wave per column turns them into exclusive tile bases and writes the totals
//   apply:    per-tile exclusive scan + base, coalesced 16-byte writes of off / cursor / suboff
// Besides the offsets, the scans prepare the EXECUTION ORDER of the accumulate launch: all complete sub-buckets (exactly L pairs) first — fulloff[b] is
// the exclusive scan of cnt_b / L — then one remainder sub-buckets per bucket with cnt_b % L > 0, grouped by length (longest first) through a counting sort
// on REM_CLASSES length classes.  Neighbouring lanes then run the same number of additions: a wave's time is its longest lane, and with bucket-major
// order every wave mixed full sub-buckets with remainders of random length (7 % idle lanes on uniform columns, 40 % on columns of 16-bit values whose
// buckets hold ~16 pairs).  Where a sub-bucket's partial sum is stored does not change (suboff[0][b] + k), so the merge levels are untouched.
constexpr uint32_t SC_T = 256, SC_E = 4, SC_TILE = SC_T * SC_E, SC_NV = MSM_MAX_LEVELS + 3, REM_CLASSES = 256;
struct ScanVals { uint32_t v[SC_NV]; };
ZK_HD uint32_t rem_class(uint32_t r) { return r < REM_CLASSES ? r : REM_CLASSES - 1; }
ZK_HD void scan_bucket_values(const MsmPlan& p, uint32_t cnt, ScanVals& o, uint32_t& e0) {
    o.v[0] = cnt;
    uint32_t e = ceil_div(cnt, p.L);
    e0 = e;
#pragma unroll
    for (uint32_t r = 0; r <= MSM_MAX_LEVELS + 1; r++) {            // v[1 + r]: level-r entries for r <= R, then (r == R + 1) the complete sub-buckets
        o.v[1 + r] = r <= p.R ? e : (r == p.R + 1 ? cnt / p.L : 0u);
        e = (e + (1u << p.Mlog) - 1) >> p.Mlog;
    }
}
ZK_KERNEL void msm_scan_tiles_kernel(MsmPlan p) {
    __shared__ uint32_t red[SC_T];
    __shared__ uint32_t rh[REM_CLASSES];
    const uint32_t col = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, B = p.B, NV = p.R + 3;
    uint32_t* sm = plan_small(p, col);
    const uint32_t b0 = tile * SC_TILE + tid * SC_E;
    ScanVals sum;
#pragma unroll
    for (uint32_t v = 0; v < SC_NV; v++) sum.v[v] = 0;
    uint32_t mx = 0;
    for (uint32_t c = tid; c < REM_CLASSES; c += SC_T) rh[c] = 0;
    __syncthreads();
    for (uint32_t e = 0; e < SC_E; e++) {
        const uint32_t b = b0 + e;
        if (b < B) {
            ScanVals x; uint32_t e0;
            const uint32_t cnt = sm[b];
            scan_bucket_values(p, cnt, x, e0);
            mx = e0 > mx ? e0 : mx;
#pragma unroll
            for (uint32_t v = 0; v < SC_NV; v++) sum.v[v] += x.v[v];
            if (cnt % p.L) atomicAdd(&rh[rem_class(cnt % p.L)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t c = tid; c < REM_CLASSES; c += SC_T) if (rh[c]) atomicAdd(&sm[p.o_remhist + c], rh[c]);
    uint32_t* tsum = sm + p.o_tiles + (size_t)tile * SC_NV;
    for (uint32_t v = 0; v <= NV; v++) {               // v == NV: the maximum
        red[tid] = v < NV ? sum.v[v < SC_NV ? v : 0] : mx;
        __syncthreads();
        for (uint32_t d = SC_T >> 1; d > 0; d >>= 1) {
            if (tid < d) red[tid] = v < NV ? red[tid] + red[tid + d] : (red[tid] > red[tid + d] ? red[tid] : red[tid + d]);
            __syncthreads();
        }
        if (tid == 0) { if (v < NV) tsum[v] = red[0]; else atomicMax(&sm[p.o_info], red[0]); }
        __syncthreads();
    }
}
ZK_KERNEL void msm_scan_tilesums_kernel(MsmPlan p, uint32_t ntiles) {
    const uint32_t col = blockIdx.x, v = threadIdx.x, B = p.B, NV = p.R + 3;
    uint32_t* sm = plan_small(p, col);
    if (v == 63) {   // the last lane: start of every remainder class in the execution order, longest class first; info[1] = number of remainders
        uint32_t run = 0;
        for (uint32_t c = REM_CLASSES; c-- > 1;) { sm[p.o_remstart + c] = run; run += sm[p.o_remhist + c]; }
        sm[p.o_info + 1] = run;
    }
    if (v >= NV) return;
    uint32_t* tsum = sm + p.o_tiles;
    uint32_t run = 0;
    for (uint32_t t = 0; t < ntiles; t++) { const uint32_t x = tsum[(size_t)t * SC_NV + v]; tsum[(size_t)t * SC_NV + v] = run; run += x; }
    if (v == 0) sm[p.o_off + B] = run;
    else if (v <= p.R + 1) sm[p.o_suboff + (size_t)(v - 1) * (B + 1) + B] = run;
    else sm[p.o_fulloff + B] = run;
}
ZK_KERNEL void msm_scan_apply_kernel(MsmPlan p) {
    __shared__ uint32_t sc[SC_T];
    const uint32_t col = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, B = p.B, NV = p.R + 3;
    uint32_t* sm = plan_small(p, col);
    const uint32_t* tsum = sm + p.o_tiles + (size_t)tile * SC_NV;
    const uint32_t b0 = tile * SC_TILE + tid * SC_E;
    {   // remainder sub-buckets of this tile take their slots in the execution order: one reservation per (tile, class), ranks inside the tile from LDS
        __shared__ uint32_t rcnt[REM_CLASSES], rbase[REM_CLASSES];
        for (uint32_t c = tid; c < REM_CLASSES; c += SC_T) rcnt[c] = 0;
        __syncthreads();
        uint32_t rk[SC_E], rc[SC_E];
        for (uint32_t e = 0; e < SC_E; e++) {
            const uint32_t b = b0 + e, r = b < B ? sm[b] % p.L : 0u;
            rc[e] = r ? rem_class(r) : 0u;
            rk[e] = r ? atomicAdd(&rcnt[rc[e]], 1u) : 0u;
        }
        __syncthreads();
        for (uint32_t c = tid; c < REM_CLASSES; c += SC_T) rbase[c] = rcnt[c] ? sm[p.o_remstart + c] + atomicAdd(&sm[p.o_remcursor + c], rcnt[c]) : 0u;
        __syncthreads();
        for (uint32_t e = 0; e < SC_E; e++) if (rc[e]) sm[p.o_remorder + rbase[rc[e]] + rk[e]] = b0 + e;
        __syncthreads();
    }
    ScanVals x[SC_E], sum;
#pragma unroll
    for (uint32_t v = 0; v < SC_NV; v++) sum.v[v] = 0;
    for (uint32_t e = 0; e < SC_E; e++) {
        uint32_t e0;
        scan_bucket_values(p, b0 + e < B ? sm[b0 + e] : 0u, x[e], e0);
#pragma unroll
        for (uint32_t v = 0; v < SC_NV; v++) sum.v[v] += x[e].v[v];
    }
    uint32_t* off = sm + p.o_off;
    uint32_t* cursor = sm + p.o_cursor;
    uint32_t* suboff = sm + p.o_suboff;
#pragma unroll
    for (uint32_t v = 0; v < SC_NV; v++) {
        if (v >= NV) continue;                            // (uniform: levels beyond R do not exist)
        sc[tid] = sum.v[v];
        __syncthreads();
        for (uint32_t d = 1; d < SC_T; d <<= 1) {
            const uint32_t add = tid >= d ? sc[tid - d] : 0;
            __syncthreads();
            sc[tid] += add;
            __syncthreads();
        }
        uint32_t run = tsum[v] + sc[tid] - sum.v[v];
        __syncthreads();
        for (uint32_t e = 0; e < SC_E; e++) {
            const uint32_t b = b0 + e;
            if (b < B) {
                if (v == 0) { off[b] = run; cursor[b] = run; }
                else if (v <= p.R + 1) suboff[(size_t)(v - 1) * (B + 1) + b] = run;
                else sm[p.o_fulloff + b] = run;
            }
            run += x[e].v[v];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bucket accumulation: one thread per sub-bucket (<= L references), XYZZ mixed additions
// ------------------------------------------------------------------------------------------------
// the chain runs on carry-free 29-bit limbs (ec.cuh xyzz29_madd_fast, field29.cuh); the 32-bit loop behind it finishes a chain that met a rare case
ZK_KERNEL void ZK_LAUNCH_BOUNDS(256) ZK_WAVES_PER_EU(4) msm_accumulate_kernel(MsmPlan p) {
    const uint32_t col = blockIdx.y, B = p.B;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t* sm = plan_small(p, col);
    const uint32_t* fulloff = sm + p.o_fulloff;
    const uint32_t n_full = fulloff[B];
    uint32_t b, k;
    if (j < n_full) {                                  // a complete sub-bucket: L pairs
        b = find_segment(fulloff, B, j);
        k = j - fulloff[b];
    } else {                                           // the remainder of a bucket; neighbours have (almost) the same length
        if (j - n_full >= sm[p.o_info + 1]) return;
        b = sm[p.o_remorder + (j - n_full)];
        k = fulloff[b + 1] - fulloff[b];
    }
    const uint32_t* off = sm + p.o_off;
    const uint32_t* sorted = p.sorted + (size_t)col * p.sorted_stride;
    uint32_t q = off[b] + k * p.L;
    const uint32_t bend = off[b + 1];
    const uint32_t end = q + p.L < bend ? q + p.L : bend;
    const uint32_t out = plan_suboff(p, col, 0)[b] + k;   // bucket-major slot of this partial sum, as the merge levels expect it
    uint32_t ref = sorted[q];
    const void* table = p.col_tables ? p.col_tables[col] : p.table;
    Affine pt = load_affine(table, ref & 0x7fffffffu);
    XYZZ acc = xyzz_identity();
    {
        // the chain on carry-free limbs: the first point enters the 2^261 form, every further point is one xyzz29_madd_fast; the (rare) step that form does not cover —
        // an identity base, a doubling, a cancellation — ends the fast loop and the 32-bit loop below finishes the chain from there
        XYZZ29 a29 = xyzz29_identity();
        bool fast = !affine_is_identity(pt);
        if (fast) {
            a29.x = Fq29::enter(pt.x);
            a29.y = Fq29::enter((ref >> 31) ? Fq::neg(pt.y) : pt.y);
            a29.zz = Fq29::one(); a29.zzz = Fq29::one();
            a29.ident = false;
            ++q;
            if (q < end) { ref = sorted[q]; pt = load_affine(table, ref & 0x7fffffffu); }
        }
        while (fast && q < end) {
            const uint32_t cur_ref = ref;
            const Affine cur = pt;
            const bool more = q + 1 < end;
            if (more) ref = sorted[q + 1];
            bool fetched = false;
            fast = !affine_is_identity(cur) && xyzz29_madd_fast(a29, cur.x, (cur_ref >> 31) ? Fq::neg(cur.y) : cur.y, [&]() {
                // the next point's 64-byte gather goes out here — late in the step, where few values are live (the kernel fits 128 VGPRs without spills), and
                // still three products (about a third of the step) ahead of its first use
                if (more) pt = load_affine(table, ref & 0x7fffffffu);
                fetched = true;
            });
            if (fast) ++q;
            else { ref = cur_ref; if (fetched) pt = cur; }            // this point again, in the complete form
        }
        acc = xyzz29_leave(a29);                                       // canonical coordinates of the library's form: the merge levels are unchanged
        if (q >= end) {
            store_xyzz(p.sub[0], (size_t)col * p.sub_stride[0] + out, acc);
            return;
        }
    }
    while (true) {
        const uint32_t cur_ref = ref;
        const Affine cur = pt;
        ++q;
        if (q < end) {  // fetch the next point while the current addition runs
            ref = sorted[q];
            pt = load_affine(table, ref & 0x7fffffffu);
        }
        xyzz_madd_signed_lazy(acc, cur, (cur_ref >> 31) != 0);   // coordinates in [0, 2q) along the chain: eight of the ten products skip their final subtraction
        if (q >= end) break;
    }
    xyzz_normalize(acc);
    store_xyzz(p.sub[0], (size_t)col * p.sub_stride[0] + out, acc);
}

// merge level r-1 -> level r: one thread per OUTPUT entry (dense), fan-in M.
ZK_KERNEL void msm_merge_kernel(MsmPlan p, uint32_t r, uint32_t Mpow_prev) {
    const uint32_t col = blockIdx.y, B = p.B;
    const uint32_t* sm = plan_small(p, col);
    if (Mpow_prev >= sm[p.o_info]) return;  // every bucket already has a single entry
    const uint32_t* so_in = plan_suboff(p, col, r - 1);
    const uint32_t* so_out = plan_suboff(p, col, r);
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= so_out[B]) return;
    const uint32_t b = find_segment(so_out, B, t);
    const uint32_t k = t - so_out[b];
    const uint32_t cnt_in = so_in[b + 1] - so_in[b];
    const void* in = p.sub[(r - 1) & 1];
    const size_t ibase = (size_t)col * p.sub_stride[(r - 1) & 1] + so_in[b] + (size_t)k * p.M;
    XYZZ acc = load_xyzz(in, ibase);
    for (uint32_t m = 1; m < p.M && k * p.M + m < cnt_in; m++) xyzz_add_lazy(acc, load_xyzz(in, ibase + m));   // coordinates in [0, 2q) along the chain (ec.cuh)
    xyzz_normalize(acc);
    store_xyzz(p.sub[r & 1], (size_t)col * p.sub_stride[r & 1] + t, acc);
}

// ------------------------------------------------------------------------------------------------
// bucket reduction  sum_w w * B_w,  bucket b has weight w = b + 1 in [1, 2^(c-1)].
// Split w = hi * 2^L + lo (L = ceil((c-1)/2)):  sum_w w B_w = 2^L * sum_hi hi * R_hi + sum_lo lo * C_lo  with the row sums
// R_hi = sum_lo B_(hi,lo) and column sums C_lo = sum_hi B_(hi,lo) — about 2 * 2^(c-1) plain additions instead of the
// (c-1) * 2^(c-2) of summing every weight-bit class over the buckets directly.  The two small weighted sums (<= 2^L + 1
// and 2^L entries) are then done by weight bits:  H_t = sum_{hi: bit t} R_hi,  L_t = sum_{lo: bit t} C_lo, c class sums
// per column in all, and the host folds 2^L * sum_t 2^t H_t + sum_t 2^t L_t  (about 3c group operations).
// One wave per group: every lane adds its members serially, then an LDS tree over the 64 lanes.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t RC_T = 64;
template <uint32_t SEG>                     // sum over segments of SEG consecutive lanes; valid in each segment's first lane
__device__ __forceinline__ XYZZ rc_wave_sum(XYZZ acc) {
    __shared__ uint4 pl[8 * RC_T];
    const uint32_t tid = threadIdx.x;
    auto put = [&](const XYZZ& v) {
        const u256* f = &v.x;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            pl[(2 * q) * RC_T + tid] = make_uint4(f[q].v[0], f[q].v[1], f[q].v[2], f[q].v[3]);
            pl[(2 * q + 1) * RC_T + tid] = make_uint4(f[q].v[4], f[q].v[5], f[q].v[6], f[q].v[7]);
        }
    };
    auto get = [&](uint32_t t) {
        XYZZ v;
        u256* f = &v.x;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 l = pl[(2 * q) * RC_T + t], h = pl[(2 * q + 1) * RC_T + t];
            f[q].v[0] = l.x; f[q].v[1] = l.y; f[q].v[2] = l.z; f[q].v[3] = l.w; f[q].v[4] = h.x; f[q].v[5] = h.y; f[q].v[6] = h.z; f[q].v[7] = h.w;
        }
        return v;
    };
    put(acc);
    __syncthreads();
    for (uint32_t d = SEG >> 1; d > 0; d >>= 1) {
        const bool on = (tid & (SEG - 1)) < d;
        if (on) { xyzz_add_lazy(acc, get(tid + d)); }              // callers normalise what they store
        __syncthreads();
        if (on) put(acc);
        __syncthreads();
    }
    return acc;
}
// grid (ceil((n_hi + n_lo) / (RC_T / RC_G)), nb): group g < n_hi is row hi = g, else column lo = g - n_hi.  out[col * (n_hi + n_lo) + g]
// RC_G lanes per group, each adding its members serially, then a log2(RC_G)-level tree inside the group.  RC_G = 64 (one wave per group) is the
// shortest dependency chain — single columns, where the launch is latency-bound (0.125 vs 0.186 ms at 2^19); RC_G = 16 spends fewer addition-times
// per group (a 64-lane group spends 6 of its 10 in the tree, most lanes idle) — batches, where the launch is throughput-bound (0.61 -> 0.40 ms for 25 columns).
template <uint32_t RC_G>
ZK_KERNEL void msm_rowcol_kernel(MsmPlan p) {
    const uint32_t col = blockIdx.y, tid = threadIdx.x, B = p.B, lane = tid & (RC_G - 1);
    const uint32_t lo_bits = p.lo_bits, n_lo = 1u << lo_bits, n_hi = (B >> lo_bits) + 1;
    const uint32_t g = blockIdx.x * (RC_T / RC_G) + tid / RC_G;
    const uint32_t r = plan_eff_levels(p, plan_small(p, col)[p.o_info]);
    const uint32_t* so = plan_suboff(p, col, r);
    const void* buf = p.sub[r & 1];
    const size_t cbase = (size_t)col * p.sub_stride[r & 1];
    XYZZ acc = xyzz_identity();
    const bool live = g < n_hi + n_lo, row = g < n_hi;
    const uint32_t fixed = row ? g : g - n_hi, count = live ? (row ? n_lo : n_hi) : 0u;
    for (uint32_t m = lane; m < count; m += RC_G) {
        const uint32_t w = row ? ((fixed << lo_bits) | m) : ((m << lo_bits) | fixed);
        if (w >= 1 && w <= B) {
            const uint32_t b = w - 1;
            if (so[b + 1] > so[b]) xyzz_add_lazy(acc, load_xyzz(buf, cbase + so[b]));
        }
    }
    acc = rc_wave_sum<RC_G>(acc);
    xyzz_normalize(acc);
    if (live && lane == 0) store_xyzz(p.cls[0], (size_t)col * (n_hi + n_lo) + g, acc);
}
// grid (c, nb): class t < hi_bits sums the rows whose index has bit t, class hi_bits + t' the columns whose index has bit t'
ZK_KERNEL void msm_rc_class_kernel(MsmPlan p) {
    const uint32_t col = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    const uint32_t lo_bits = p.lo_bits, n_lo = 1u << lo_bits, n_hi = (p.B >> lo_bits) + 1, hi_bits = (uint32_t)p.c - lo_bits;
    const bool rows = t < hi_bits;
    const uint32_t bit = rows ? t : t - hi_bits, count = rows ? n_hi : n_lo;
    const size_t base = (size_t)col * (n_hi + n_lo) + (rows ? 0 : n_hi);
    XYZZ acc = xyzz_identity();
    for (uint32_t m = tid; m < count; m += RC_T)
        if ((m >> bit) & 1u) xyzz_add_lazy(acc, load_xyzz(p.cls[0], base + m));
    acc = rc_wave_sum<RC_T>(acc);
    xyzz_normalize(acc);
    if (tid == 0) store_xyzz(p.cls[1], (size_t)col * p.c + t, acc);
}

// ------------------------------------------------------------------------------------------------
// base-table expansion (registration time)
// ------------------------------------------------------------------------------------------------
ZK_KERNEL void g1_affine_to_xyzz_kernel(const void* aff, uint32_t n, void* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_xyzz(out, i, xyzz_from_affine(load_affine(aff, i)));
}
ZK_KERNEL void g1_dbl_times_kernel(void* pts, uint32_t n, int times) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZ p = load_xyzz(pts, i);
    for (int k = 0; k < times; k++) p = xyzz_dbl(p);
    store_xyzz(pts, i, p);
}
// XYZZ -> affine with one inversion per `chunk` points (Montgomery's trick); out.x doubles as the
// prefix-product scratch between the two sweeps.
ZK_KERNEL void g1_batch_to_affine_kernel(const void* in, uint32_t n, uint32_t chunk, void* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo64 = (uint64_t)t * chunk;
    if (lo64 >= n) return;
    const uint32_t lo = (uint32_t)lo64;
    const uint32_t hi = lo + chunk < n ? lo + chunk : n;
    u256 acc = Fq::one();
    for (uint32_t k = lo; k < hi; k++) {
        u256 zz = load_u256(in, 4 * (size_t)k + 2), zzz = load_u256(in, 4 * (size_t)k + 3);
        store_u256(out, 2 * (size_t)k, acc);
        if (!Fq::is_zero(zz)) acc = Fq::mul(acc, Fq::mul(zz, zzz));
    }
    u256 inv = Fq::inv(acc);
    for (uint32_t k = hi; k-- > lo;) {
        XYZZ p = load_xyzz(in, k);
        Affine o;
        if (Fq::is_zero(p.zz)) {
            o.x = Fq::zero(); o.y = Fq::zero();
        } else {
            u256 pre = load_u256(out, 2 * (size_t)k);
            u256 I = Fq::mul(inv, pre);  // 1 / (zz * zzz)
            inv = Fq::mul(inv, Fq::mul(p.zz, p.zzz));
            o.x = Fq::mul(p.x, Fq::mul(I, p.zzz));  // X / ZZ
            o.y = Fq::mul(p.y, Fq::mul(I, p.zz));   // Y / ZZZ
        }
        store_affine(out, k, o);
    }
}

// out[i] = [s_i] G via an 8-bit fixed-window table of the generator: gtab[w][d-1] = d * 256^w * G
ZK_KERNEL void g1_fixed_base_kernel(const void* scalars, uint32_t n, const void* gtab, void* out_xyzz) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u256 s = Fr::from_mont(load_u256(scalars, i));
    XYZZ acc = xyzz_identity();
    for (int w = 0; w < 32; w++) {
        uint32_t d = (s.v[w >> 2] >> (8 * (w & 3))) & 0xffu;
        if (d) {
            Affine p = load_affine(gtab, (size_t)w * 255 + (d - 1));
            xyzz_madd(acc, p.x, p.y);
        }
    }
    store_xyzz(out_xyzz, i, acc);
}
// gtab build: row w from row w-1: entry(d) of row w = 256 * entry(d) of row w-1; row 0 = d*G
ZK_KERNEL void g1_gtab_row0_kernel(void* row_xyzz) {  // 255 threads: d*G by double-and-add
    const uint32_t d = threadIdx.x + 1;
    if (d > 255) return;
    Affine g;
    g.x = Fq::one();
    g.y = Fq::dbl(Fq::one());  // generator (1, 2)
    XYZZ acc = xyzz_identity();
    for (int b = 7; b >= 0; b--) {
        acc = xyzz_dbl(acc);
        if ((d >> b) & 1) xyzz_madd(acc, g.x, g.y);
    }
    store_xyzz(row_xyzz, threadIdx.x, acc);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int windows_for(int c) { return 254 / c + 1; }

static int pick_c(size_t n, const Tune& t) {
    if (t.msm_c >= 3 && t.msm_c <= MSM_MAX_C) return t.msm_c;
    // From 2^22 points a 20-bit window: 13 windows per scalar instead of 16 (-19 % additions) for 2^19 buckets and the two-level sort — measured on single MSMs
    // (profiles/r05/run353, run354): 2^22 627 -> 699, 2^23 683 -> 775, 2^24 704 -> 827 Mscalar/s; 2^21 +5 %, 2^20 -3 %.  Not below: the prover's BATCHES at
    // k = 20 / 21 lose what the shorter chains save to the per-column bucket reduction (x 4 buckets) and the two-level sort (k = 21: 226 -> 229 ms per proof).
    // Of the wider windows only c = 17 and c = 20 have a well-filled top window (254 = 14 x 17 + 16 = 12 x 20 + 14 bits): at c = 18, 19, 21, 22 the top digit of every scalar
    // falls into the lowest 2^2 .. 2^12 buckets (1/W of all pairs: deep merge levels there, and c = 22 measured behind c = 20 at 2^24).
    if (n >= ((size_t)1 << 22)) return 20;
    int best = 3;
    double best_cost = 1e300;
    for (int c = 3; c <= 16; c++) {
        double cost = (double)n * windows_for(c) + 24.0 * (double)(1u << (c - 1));
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}

// Expand `n` affine points (device) into the window-multiple table.
static int build_table(zk_ctx* ctx, const void* d_pts, size_t n, BaseTable& bt) {
    const int c = bt.c, W = bt.W;
    const size_t tbytes = (size_t)W * n * 64;
    bt.mem = std::make_shared<TableMem>();                            // (the owner first: whatever fails below, the table goes with its last holder)
    bt.mem->device = ctx->device;
    ZK_HIP(hipMalloc(&bt.mem->p, tbytes));
    bt.d_table = bt.mem->p;
    ZK_HIP(hipMemcpyAsync(bt.d_table, d_pts, n * 64, hipMemcpyDeviceToDevice, ctx->stream));
    if (W > 1) {
        ZK_HIP(ctx->ws_pts.ensure(n * 128));
        const int blk = ctx->tune.msm_block;
        const uint32_t grid = (uint32_t)((n + blk - 1) / blk);
        ZK_LAUNCH(g1_affine_to_xyzz_kernel, grid, blk, 0, ctx->stream, d_pts, (uint32_t)n, ctx->ws_pts.p);
        ZK_CHECK_LAUNCH();
        const uint32_t chunk = 32;
        const uint32_t bgrid = (uint32_t)(((n + chunk - 1) / chunk + blk - 1) / blk);
        for (int j = 1; j < W; j++) {
            ZK_LAUNCH(g1_dbl_times_kernel, grid, blk, 0, ctx->stream, ctx->ws_pts.p, (uint32_t)n, c);
            ZK_CHECK_LAUNCH();
            ZK_LAUNCH(g1_batch_to_affine_kernel, bgrid, blk, 0, ctx->stream, (const void*)ctx->ws_pts.p, (uint32_t)n, chunk,
                      (void*)((char*)bt.d_table + (size_t)j * n * 64));
            ZK_CHECK_LAUNCH();
        }
    }
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

int msm_register(zk_ctx* ctx, const void* pts, size_t n, bool on_device, uint64_t* handle) {
    if (!pts || !handle || n == 0) return ctx->fail(ZK_ERR_ARG, "zk_bases_register: null/empty");
    BaseTable bt;
    bt.n = n;
    bt.c = pick_c(n, ctx->tune);
    bt.W = windows_for(bt.c);
    if ((uint64_t)bt.W * n >= (1ull << 31)) return ctx->fail(ZK_ERR_LIMIT, "zk_bases_register: W*n = %llu exceeds 2^31", (unsigned long long)bt.W * n);
    const void* d_pts = pts;
    if (!on_device) {
        ZK_HIP(ctx->ws_tmp.ensure(n * 64));
        ZK_HIP(hipMemcpyAsync(ctx->ws_tmp.p, pts, n * 64, hipMemcpyHostToDevice, ctx->stream));
        d_pts = ctx->ws_tmp.p;
    }
    int rc = build_table(ctx, d_pts, n, bt);
    if (rc) return rc;                                            // (bt.mem frees a partly built table)
    *handle = ctx->next_handle++;
    ctx->bases[*handle] = bt;
    return ZK_OK;
}

// lend a registered table to another context on the same device: no second copy in HBM (one expanded SRS table per process, not per context)
int msm_share(zk_ctx* dst, const BaseTable& bt, uint64_t* handle) {
    *handle = dst->next_handle++;
    dst->bases[*handle] = bt;
    return ZK_OK;
}

// ---- run-length path: prefix sums of the bases, adjacent differences of the scalars -------------------------------------------------------------
// total of chunk t of the affine points (XYZZ), then — given the exclusive prefix of the chunk totals — the inclusive prefix sums inside the chunk
ZK_KERNEL void g1_chunk_total_kernel(const void* aff, uint32_t n, uint32_t chunk, void* totals) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo = (uint64_t)t * chunk;
    if (lo >= n) return;
    const uint32_t hi = (uint32_t)std::min<uint64_t>(lo + chunk, n);
    XYZZ acc = xyzz_identity();
    for (uint32_t k = (uint32_t)lo; k < hi; k++) xyzz_madd_signed(acc, load_affine(aff, k), false);
    store_xyzz(totals, t, acc);
}
ZK_KERNEL void g1_chunk_prefix_kernel(const void* aff, uint32_t n, uint32_t chunk, const void* excl, void* out_xyzz) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo = (uint64_t)t * chunk;
    if (lo >= n) return;
    const uint32_t hi = (uint32_t)std::min<uint64_t>(lo + chunk, n);
    XYZZ acc = load_xyzz(excl, t);
    for (uint32_t k = (uint32_t)lo; k < hi; k++) { xyzz_madd_signed(acc, load_affine(aff, k), false); store_xyzz(out_xyzz, k, acc); }
}
// counts[col] = {non-zero scalars, non-zero adjacent differences (a[i] != a[i+1], a[n] := 0)}
// Each thread tests PROBE_EPT elements a workgroup-stride apart (neighbouring lanes read neighbouring 32-byte elements; the successor a[i + 1] is the next lane's
// element, so its load hits the same cache lines), keeps both counts packed in one register (<= 16 bits each per workgroup) and the workgroup folds them with
// an LDS tree — no two lanes ever add to the same counter.
constexpr uint32_t PROBE_EPT = 8;
ZK_KERNEL void msm_runs_probe_kernel(const void* const* cols, uint32_t n, uint32_t* counts) {
    __shared__ uint32_t sh[256];
    const uint32_t col = blockIdx.y;
    const uint64_t base = (uint64_t)blockIdx.x * blockDim.x * PROBE_EPT + threadIdx.x;
    uint32_t c = 0;                                                  // low half: non-zero scalars, high half: non-zero adjacent differences
#pragma unroll
    for (uint32_t e = 0; e < PROBE_EPT; e++) {
        const uint64_t i = base + (uint64_t)e * blockDim.x;
        if (i < n) {
            const u256 a = load_u256(cols[col], i);
            const u256 b = i + 1 < n ? load_u256(cols[col], i + 1) : Fr::zero();
            c += (Fr::is_zero(a) ? 0u : 1u) + (Fr::eq(a, b) ? 0u : 1u << 16);
        }
    }
    sh[threadIdx.x] = c;
    __syncthreads();
    for (uint32_t d = blockDim.x >> 1; d; d >>= 1) {
        if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0 && sh[0]) {
        if (sh[0] & 0xffffu) atomicAdd(&counts[2 * col], sh[0] & 0xffffu);
        if (sh[0] >> 16) atomicAdd(&counts[2 * col + 1], sh[0] >> 16);
    }
}
ZK_KERNEL void fr_adjacent_diff_kernel(const void* const* cols, const uint32_t* which, uint32_t n, void* out) {
    const uint32_t f = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const void* col = cols[which[f]];
    const u256 a = load_u256(col, i);
    store_u256(out, (size_t)f * n + i, i + 1 < n ? Fr::sub(a, load_u256(col, i + 1)) : a);
}

int msm_enable_runs(zk_ctx* ctx, uint64_t handle) {
    auto it = ctx->bases.find(handle);
    if (it == ctx->bases.end()) return ctx->fail(ZK_ERR_ARG, "zk_bases_enable_runs: unknown handle %llu", (unsigned long long)handle);
    BaseTable& bt = it->second;
    if (bt.d_runs_table) return ZK_OK;
    const size_t n = bt.n;
    const uint32_t chunk = 64, m = (uint32_t)((n + chunk - 1) / chunk);
    const int blk = ctx->tune.msm_block;
    hipStream_t st = ctx->stream;
    DevTmp t_tot, t_pre, t_aff;                                       // freed on every way out
    hipError_t e = hipMalloc(&t_tot.p, (size_t)m * 128);
    if (e == hipSuccess) e = hipMalloc(&t_pre.p, n * 128);
    if (e == hipSuccess) e = hipMalloc(&t_aff.p, n * 64);
    if (e != hipSuccess) return ctx->fail(ZK_ERR_HIP, "zk_bases_enable_runs: device allocation failed");
    void* const d_tot = t_tot.p; void* const d_pre = t_pre.p; void* const d_aff = t_aff.p;
    ZK_LAUNCH(g1_chunk_total_kernel, (m + blk - 1) / blk, blk, 0, st, (const void*)bt.d_table, (uint32_t)n, chunk, d_tot);
    std::vector<XYZZ> tot(m), excl(m);
    if (hipMemcpyAsync(tot.data(), d_tot, (size_t)m * 128, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return ctx->fail(ZK_ERR_HIP, "zk_bases_enable_runs: chunk totals");
    XYZZ run = xyzz_identity();                                      // exclusive scan of the chunk totals on the host (one-time, n / 64 additions)
    for (uint32_t t = 0; t < m; t++) { excl[t] = run; xyzz_add(run, tot[t]); }
    if (hipMemcpyAsync(d_tot, excl.data(), (size_t)m * 128, hipMemcpyHostToDevice, st) != hipSuccess) return ctx->fail(ZK_ERR_HIP, "zk_bases_enable_runs: upload");
    ZK_LAUNCH(g1_chunk_prefix_kernel, (m + blk - 1) / blk, blk, 0, st, (const void*)bt.d_table, (uint32_t)n, chunk, (const void*)d_tot, d_pre);
    const uint32_t bchunk = 32;
    ZK_LAUNCH(g1_batch_to_affine_kernel, (uint32_t)(((n + bchunk - 1) / bchunk + blk - 1) / blk), blk, 0, st, (const void*)d_pre, (uint32_t)n, bchunk, d_aff);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ctx->fail(ZK_ERR_HIP, "zk_bases_enable_runs: prefix kernels");
    t_pre.reset();
    BaseTable ps;
    ps.n = n; ps.c = bt.c; ps.W = bt.W;
    int rc = build_table(ctx, d_aff, n, ps);
    if (rc) return rc;
    bt.d_runs_table = ps.d_table;
    bt.runs_mem = ps.mem;
    return ZK_OK;
}

int msm_release(zk_ctx* ctx, uint64_t handle) {
    auto it = ctx->bases.find(handle);
    if (it == ctx->bases.end()) return ctx->fail(ZK_ERR_ARG, "zk_bases_release: unknown handle %llu", (unsigned long long)handle);
    ctx->bases.erase(it);                                          // the memory goes when its last holder lets go
    return ZK_OK;
}

static void xyzz_to_jacobian_host(const XYZZ& p, void* out96) {
    u256* o = reinterpret_cast<u256*>(out96);
    if (xyzz_is_identity(p)) { o[0] = Fq::zero(); o[1] = Fq::zero(); o[2] = Fq::zero(); return; }
    u256 I = Fq::inv(Fq::mul(p.zz, p.zzz));
    o[0] = Fq::mul(p.x, Fq::mul(I, p.zzz));
    o[1] = Fq::mul(p.y, Fq::mul(I, p.zz));
    o[2] = Fq::one();
}
// normalise a batch with ONE field inversion (Montgomery's trick)
static void xyzz_batch_to_jacobian_host(const std::vector<XYZZ>& pts, void* out) {
    const size_t nb = pts.size();
    std::vector<u256> pre(nb);
    u256 acc = Fq::one();
    for (size_t i = 0; i < nb; i++) { pre[i] = acc; if (!xyzz_is_identity(pts[i])) acc = Fq::mul(acc, Fq::mul(pts[i].zz, pts[i].zzz)); }
    u256 inv = Fq::inv(acc);
    for (size_t i = nb; i-- > 0;) {
        u256* o = reinterpret_cast<u256*>((char*)out + i * 96);
        const XYZZ& p = pts[i];
        if (xyzz_is_identity(p)) { o[0] = Fq::zero(); o[1] = Fq::zero(); o[2] = Fq::zero(); continue; }
        u256 I = Fq::mul(inv, pre[i]);
        inv = Fq::mul(inv, Fq::mul(p.zz, p.zzz));
        o[0] = Fq::mul(p.x, Fq::mul(I, p.zzz));
        o[1] = Fq::mul(p.y, Fq::mul(I, p.zz));
        o[2] = Fq::one();
    }
}

// core: nb columns of n scalars each against one table; out_xyzz[nb] receive the unnormalised sums.
// h_col_tables (optional): per-column table pointer (bt.d_table or bt.d_runs_table).
static int msm_core(zk_ctx* ctx, const BaseTable& bt, const void* const* h_scal_ptrs, uint32_t nb, size_t n, XYZZ* out_xyzz, const void* const* h_col_tables = nullptr) {
    const int c = bt.c, W = bt.W;
    const uint32_t B = 1u << (c - 1);
    const Tune& tn = ctx->tune;
    const uint64_t pairs_max = (uint64_t)n * W;
    uint32_t Mlog = 1;
    while ((2u << Mlog) <= (uint32_t)std::max(2, tn.msm_merge_fanin) && Mlog < 8) Mlog++;
    const uint32_t M = 1u << Mlog;   // merge fan-in rounded down to a power of two
    // (wide windows = single large MSMs: hundreds of pairs per bucket, so longer sub-buckets halve the merge level's work — 2^24: reduction 1.33 -> 0.82 ms at 128, profiles/r05/run368;
    //  the prover's batches measure flat from 48 to 96: run369)
    uint32_t L = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((uint64_t)nb * pairs_max / (uint64_t)tn.msm_target_threads, (uint64_t)tn.msm_min_chunk),
                                              (uint64_t)(c > 16 ? tn.msm_max_chunk_wide : tn.msm_max_chunk));
    MsmPlan p;
    memset(&p, 0, sizeof p);
    p.n = (uint32_t)n; p.n_table = (uint32_t)bt.n; p.B = B; p.L = L; p.M = M; p.Mlog = Mlog; p.nb = nb; p.c = c; p.W = W;
    // merge levels for the worst case (one bucket holds every pair)
    uint64_t cap[MSM_MAX_LEVELS + 1];
    cap[0] = pairs_max / L + B + 1;
    uint32_t R = 0;
    {
        uint64_t worst = pairs_max / L + 1;
        while (worst > 1 && R < MSM_MAX_LEVELS) { worst = (worst + M - 1) / M; R++; cap[R] = cap[R - 1] / M + B + 1; }
        if (worst > 1) return ctx->fail(ZK_ERR_LIMIT, "zk_msm: merge depth exceeds %d levels (raise msm_merge_fanin)", MSM_MAX_LEVELS);
    }
    p.R = R;
    p.o_off = B; p.o_cursor = p.o_off + B + 1; p.o_suboff = p.o_cursor + B; p.o_info = p.o_suboff + (R + 1) * (B + 1);
    const uint32_t scan_tiles = ceil_div(B, SC_TILE);
    p.o_tiles = p.o_info + 4;
    p.o_fulloff = p.o_tiles + scan_tiles * SC_NV;
    p.o_remorder = p.o_fulloff + B + 1;
    p.o_remhist = p.o_remorder + B;
    p.o_remstart = p.o_remhist + REM_CLASSES;
    p.o_remcursor = p.o_remstart + REM_CLASSES;
    p.o_bintot = p.o_remcursor + REM_CLASSES;
    p.low_bits = c <= 16 ? (uint32_t)std::min(7, c - 1) : (uint32_t)std::max(c - 1 - std::min(std::max(tn.msm_wide_bins_log, 8), 11), 5);      // c <= 16: at most 256 bins of 128 buckets; wider: 2^msm_wide_bins_log bins
    p.nbin = B >> p.low_bits;
    p.bch = (uint32_t)std::max(64, tn.msm_bsort_chunk);
    p.o_bincur = p.o_bintot + p.nbin;
    p.o_binbase = p.o_bincur + p.nbin;
    p.o_chunkbase = p.o_binbase + p.nbin + 1;
    p.small_stride = p.o_chunkbase + p.nbin + 1;
    const int part_threads = std::min(std::max(tn.msm_part_threads, 64), std::min(1024, tn.msm_sort_threads));
    p.pch = std::min<uint32_t>(std::max<uint32_t>(1, (uint32_t)std::min(std::max(tn.msm_part_pairs, 64), 16384) / (uint32_t)W), PART_SPT * (uint32_t)part_threads);
    if (p.nbin > MSM_MAX_BINS || (1u << p.low_bits) > MSM_MAX_LOW) return ctx->fail(ZK_ERR_LIMIT, "zk_msm: window width c = %d exceeds %d", c, MSM_MAX_C);
    ZK_HIP(ctx->ws_small.ensure((size_t)nb * p.small_stride * 4 + 2 * (size_t)nb * sizeof(void*) + 64));
    p.small = (uint32_t*)ctx->ws_small.p;
    const void** d_ptrs = (const void**)((char*)ctx->ws_small.p + (((size_t)nb * p.small_stride * 4 + 15) & ~(size_t)15));
    p.scalars = d_ptrs;
    p.table = bt.d_table;
    p.col_tables = h_col_tables ? d_ptrs + nb : nullptr;
    p.sorted_stride = (pairs_max + 4 + 15) & ~(uint64_t)15;      // columns of the 1-, 4-byte arrays start 16-byte aligned
    ZK_HIP(ctx->ws_sorted.ensure((size_t)nb * p.sorted_stride * 4));
    p.sorted = (uint32_t*)ctx->ws_sorted.p;
    ZK_HIP(ctx->ws_mid.ensure((size_t)nb * p.sorted_stride * 6 + 64));          // (vector loads may touch the 3 entries after a column's last)
    p.mid_ref = (uint32_t*)ctx->ws_mid.p;
    p.mid_low = (uint16_t*)((char*)ctx->ws_mid.p + (size_t)nb * p.sorted_stride * 4);
    p.sub_stride[0] = cap[0];
    p.sub_stride[1] = R >= 1 ? cap[1] : 1;
    ZK_HIP(ctx->ws_sub0.ensure((size_t)nb * p.sub_stride[0] * 128));
    ZK_HIP(ctx->ws_sub1.ensure((size_t)nb * p.sub_stride[1] * 128));
    p.sub[0] = ctx->ws_sub0.p; p.sub[1] = ctx->ws_sub1.p;
    p.lo_bits = (uint32_t)(c - 1 + 1) / 2;                               // ceil((c-1)/2)
    const uint32_t n_groups = (B >> p.lo_bits) + 1 + (1u << p.lo_bits);  // rows + columns of the weight matrix
    ZK_HIP(ctx->ws_cls0.ensure((size_t)nb * n_groups * 128));
    ZK_HIP(ctx->ws_cls1.ensure((size_t)nb * c * 128 + 128));
    p.cls[0] = ctx->ws_cls0.p; p.cls[1] = ctx->ws_cls1.p;

    hipStream_t st = ctx->stream;
    ZK_HIP(hipMemsetAsync(p.small, 0, (size_t)nb * p.small_stride * 4, st));
    ZK_HIP(hipMemcpyAsync((void*)d_ptrs, h_scal_ptrs, (size_t)nb * sizeof(void*), hipMemcpyHostToDevice, st));
    if (h_col_tables) ZK_HIP(hipMemcpyAsync((void*)(d_ptrs + nb), h_col_tables, (size_t)nb * sizeof(void*), hipMemcpyHostToDevice, st));
    int wgs = tn.msm_sort_wgs;
    {   // do not spread small inputs over many workgroups (each one flushes its counters)
        uint64_t per = (uint64_t)tn.msm_sort_threads * 4;
        uint64_t want = (n + per - 1) / per;
        if (want < (uint64_t)wgs) wgs = (int)std::max<uint64_t>(want, 1);
        if (nb > 1) wgs = std::max(1, std::min(wgs, (int)((tn.msm_sort_batch_wgs) / nb) + 1));
    }
    // one-level sort: the rank registers of msm_scatter_kernel hold SC_SPT scalars per thread and 16-bit ranks
    const uint64_t chunk_max = std::min<uint64_t>((uint64_t)SC_SPT * tn.msm_sort_threads, 65536 / (uint64_t)W);
    const uint64_t need = (n + chunk_max - 1) / chunk_max;
    const bool two_level = W > (int)SC_WMAX || c > 16 || tn.msm_two_level_sort != 0;   // (tiny inputs pick narrow windows: more of them than the rank registers cover; wide windows: more counters than LDS)
    const uint32_t bsort_grid = (uint32_t)(pairs_max / p.bch) + p.nbin;       // >= sum over bins of ceil(size / bch): workgroups past the last chunk exit
    EvTimer t_sort(ctx, "msm_sort");
    if (two_level) {
        ZK_LAUNCH(msm_bin_hist_kernel, dim3(wgs, nb), tn.msm_sort_threads, 0, st, p);
        ZK_CHECK_LAUNCH();
        ZK_LAUNCH(msm_bin_base_kernel, nb, 256, 0, st, p);
        ZK_CHECK_LAUNCH();
        ZK_LAUNCH(msm_partition_kernel, dim3(ceil_div(p.n, p.pch), nb), part_threads, ((size_t)p.pch * W * 2 + 4 * (size_t)p.nbin) * 4, st, p);
        ZK_CHECK_LAUNCH();
        ZK_LAUNCH(msm_bsort_count_kernel, dim3(bsort_grid, nb), tn.msm_bsort_threads, (size_t)4 << p.low_bits, st, p);
        ZK_CHECK_LAUNCH();
    } else {
        ZK_LAUNCH(msm_hist_kernel, dim3(wgs, nb), tn.msm_sort_threads, (size_t)B * 4, st, p);
        ZK_CHECK_LAUNCH();
    }
    ZK_LAUNCH(msm_scan_tiles_kernel, dim3(scan_tiles, nb), SC_T, 0, st, p);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(msm_scan_tilesums_kernel, nb, 64, 0, st, p, scan_tiles);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(msm_scan_apply_kernel, dim3(scan_tiles, nb), SC_T, 0, st, p);
    ZK_CHECK_LAUNCH();
    if (two_level) {
        ZK_LAUNCH(msm_bsort_place_kernel, dim3(bsort_grid, nb), tn.msm_bsort_threads, (size_t)4 << p.low_bits, st, p);
        ZK_CHECK_LAUNCH();
    } else {
        ZK_LAUNCH(msm_scatter_kernel, dim3((uint32_t)std::max<uint64_t>(need, (uint64_t)wgs), nb), tn.msm_sort_threads, (size_t)B * 4, st, p);
        ZK_CHECK_LAUNCH();
    }
    t_sort.stop();

    const int blk = tn.msm_block;
    EvTimer t_acc(ctx, "msm_accumulate");
    if (blk > 256) return ctx->fail(ZK_ERR_ARG, "msm_block: at most 256 threads per workgroup");
    {
        const dim3 grid((uint32_t)((cap[0] + blk - 1) / blk), nb);
        ZK_LAUNCH(msm_accumulate_kernel, grid, blk, 0, st, p);
    }
    ZK_CHECK_LAUNCH();
    t_acc.stop();

    EvTimer t_red(ctx, "msm_reduce");
    {
        uint64_t pw = 1;
        for (uint32_t r = 1; r <= R; r++) {
            ZK_LAUNCH(msm_merge_kernel, dim3((uint32_t)((cap[r] + blk - 1) / blk), nb), blk, 0, st, p, r, (uint32_t)std::min<uint64_t>(pw, 0xffffffffull));
            ZK_CHECK_LAUNCH();
            pw *= M;
        }
    }
    if (nb >= 4) { ZK_LAUNCH(msm_rowcol_kernel<16>, dim3((n_groups + 3) / 4, nb), RC_T, 0, st, p); }
    else { ZK_LAUNCH(msm_rowcol_kernel<64>, dim3(n_groups, nb), RC_T, 0, st, p); }
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(msm_rc_class_kernel, dim3((uint32_t)c, nb), RC_T, 0, st, p);
    ZK_CHECK_LAUNCH();
    t_red.stop();
    std::vector<XYZZ> cls((size_t)nb * c);
    ZK_HIP(hipMemcpyAsync(cls.data(), p.cls[1], (size_t)nb * c * 128, hipMemcpyDeviceToHost, st));
    std::vector<uint32_t> npairs(ctx->timing ? nb : 0);
    for (uint32_t col = 0; col < (uint32_t)npairs.size(); col++)   // off[B] = pairs of the column (zero digits are skipped)
        ZK_HIP(hipMemcpyAsync(&npairs[col], plan_small(p, col) + p.o_off + B, 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    for (uint32_t v : npairs) ctx->last_ms["msm_pairs"] += (double)v;
    if (ctx->timing) ctx->last_ms["msm_columns"] += (double)nb;
    t_sort.resolve(); t_acc.resolve(); t_red.resolve();
    const int hi_bits = c - (int)p.lo_bits;
    for (uint32_t col = 0; col < nb; col++) {   // 2^L * sum_t 2^t H_t + sum_t 2^t L_t, Horner from the top class of each half
        const XYZZ* cc = &cls[(size_t)col * c];
        XYZZ h = cc[hi_bits - 1];
        for (int t = hi_bits - 2; t >= 0; t--) { h = xyzz_dbl(h); xyzz_add(h, cc[t]); }
        for (uint32_t t = 0; t < p.lo_bits; t++) h = xyzz_dbl(h);
        XYZZ l = cc[c - 1];
        for (int t = (int)p.lo_bits - 2; t >= 0; t--) { l = xyzz_dbl(l); xyzz_add(l, cc[hi_bits + t]); }
        xyzz_add(h, l);
        out_xyzz[col] = h;
    }
    return ZK_OK;
}

// scalars: nb pointers (host array) to host or device columns of n scalars
int msm_run_batch(zk_ctx* ctx, uint64_t handle, const void* const* scalars, size_t nb, size_t n, bool on_device, void* out, int partial) {
    auto it = ctx->bases.find(handle);
    if (it == ctx->bases.end()) return ctx->fail(ZK_ERR_ARG, "zk_msm: unknown bases handle %llu", (unsigned long long)handle);
    const BaseTable& bt = it->second;
    if (nb == 0) return ZK_OK;
    if (!out || !scalars) return ctx->fail(ZK_ERR_ARG, "zk_msm: null pointer");
    if (n > bt.n) return ctx->fail(ZK_ERR_ARG, "zk_msm: n = %zu exceeds registered table size %zu", n, bt.n);
    if (nb > 4096) return ctx->fail(ZK_ERR_LIMIT, "zk_msm_batch: more than 4096 columns");
    std::vector<XYZZ> res(nb, xyzz_identity());
    if (n > 0) {
        std::vector<const void*> ptrs(nb);
        for (size_t i = 0; i < nb; i++) {
            if (!scalars[i]) return ctx->fail(ZK_ERR_ARG, "zk_msm: null scalar column %zu", i);
            ptrs[i] = scalars[i];
        }
        if (!on_device) {
            ZK_HIP(ctx->ws_scalars.ensure(nb * n * 32));
            for (size_t i = 0; i < nb; i++) {
                void* d = (char*)ctx->ws_scalars.p + i * n * 32;
                ZK_HIP(hipMemcpyAsync(d, scalars[i], n * 32, hipMemcpyHostToDevice, ctx->stream));
                ptrs[i] = d;
            }
        }
        // per column: directly, or — when the table has its prefix-sum twin and the column is mostly runs of equal values — through its adjacent differences
        std::vector<uint32_t> direct, runs;
        if (bt.d_runs_table && ctx->tune.msm_runs && n >= 1024) {
            hipStream_t st = ctx->stream;
            ZK_HIP(ctx->ws_runs.ensure(nb * (sizeof(void*) + 8 + 4) + 64));
            const void** d_cols = (const void**)ctx->ws_runs.p;
            uint32_t* d_counts = (uint32_t*)((char*)ctx->ws_runs.p + nb * sizeof(void*));
            ZK_HIP(hipMemcpyAsync((void*)d_cols, ptrs.data(), nb * sizeof(void*), hipMemcpyHostToDevice, st));
            ZK_HIP(hipMemsetAsync(d_counts, 0, nb * 8, st));
            ZK_LAUNCH(msm_runs_probe_kernel, dim3((uint32_t)((n + 256 * PROBE_EPT - 1) / (256 * PROBE_EPT)), (uint32_t)nb), 256, 0, st, (const void* const*)d_cols, (uint32_t)n, d_counts);
            ZK_CHECK_LAUNCH();
            std::vector<uint32_t> counts(2 * nb);
            ZK_HIP(hipMemcpyAsync(counts.data(), d_counts, nb * 8, hipMemcpyDeviceToHost, st));
            ZK_HIP(hipStreamSynchronize(st));
            for (uint32_t i = 0; i < nb; i++)       // worth it when the differences are clearly sparser than the column
                ((uint64_t)counts[2 * i + 1] * 5 < (uint64_t)counts[2 * i] * 4 && (counts[2 * i] >= 4096 || ctx->tune.msm_runs >= 2) ? runs : direct).push_back(i);
            uint64_t saved = 0;                        // point additions the run columns avoid, against one more pass over them (the difference kernel)
            for (uint32_t i : runs) saved += (uint64_t)(counts[2 * i] - counts[2 * i + 1]) * bt.W;
            // (msm_runs = 2: take the run path whenever it is sparser, whatever the size — tests)
            if (saved < (1u << 18) && ctx->tune.msm_runs < 2) { for (uint32_t i : runs) direct.push_back(i); runs.clear(); std::sort(direct.begin(), direct.end()); }
        } else for (uint32_t i = 0; i < nb; i++) direct.push_back(i);
        std::vector<const void*> col_tables;
        if (!runs.empty()) {
            hipStream_t st = ctx->stream;
            const size_t Fn = runs.size();
            const size_t head = (nb * (sizeof(void*) + 8) + Fn * 4 + 255) & ~(size_t)255;
            ZK_HIP(ctx->ws_runs.ensure(head + Fn * n * 32 + 64));
            const void** d_cols = (const void**)ctx->ws_runs.p;                       // (ensure may have moved the buffer: upload the pointers again)
            uint32_t* d_which = (uint32_t*)((char*)ctx->ws_runs.p + nb * (sizeof(void*) + 8));
            char* d_diff = (char*)ctx->ws_runs.p + head;
            ZK_HIP(hipMemcpyAsync((void*)d_cols, ptrs.data(), nb * sizeof(void*), hipMemcpyHostToDevice, st));
            ZK_HIP(hipMemcpyAsync(d_which, runs.data(), Fn * 4, hipMemcpyHostToDevice, st));
            ZK_LAUNCH(fr_adjacent_diff_kernel, dim3((uint32_t)((n + 255) / 256), (uint32_t)Fn), 256, 0, st, (const void* const*)d_cols, (const uint32_t*)d_which, (uint32_t)n, (void*)d_diff);
            ZK_CHECK_LAUNCH();
            // ONE launch sequence for the whole batch: run columns read their differences and the prefix-sum twin, the others their scalars and the table
            col_tables.assign(nb, bt.d_table);
            for (size_t f = 0; f < Fn; f++) { ptrs[runs[f]] = d_diff + f * n * 32; col_tables[runs[f]] = bt.d_runs_table; }
            ctx->last_ms["msm_run_columns"] += (double)Fn;
        }
        int rc = msm_core(ctx, bt, ptrs.data(), (uint32_t)nb, n, res.data(), col_tables.empty() ? nullptr : col_tables.data());
        if (rc) return rc;
    }
    if (partial) memcpy(out, res.data(), nb * 128);
    else if (nb == 1) xyzz_to_jacobian_host(res[0], out);
    else xyzz_batch_to_jacobian_host(res, out);
    return ZK_OK;
}
int msm_run(zk_ctx* ctx, uint64_t handle, const void* scalars, size_t n, bool on_device, void* out, int partial) {
    if (!scalars && n) return ctx->fail(ZK_ERR_ARG, "zk_msm: null pointer");
    const void* one[1] = {scalars ? scalars : (const void*)ctx};
    return msm_run_batch(ctx, handle, one, 1, n, on_device, out, partial);
}

int g1_sum_xyzz_host(const void* xyzz, size_t count, void* out_jac) {
    const XYZZ* p = reinterpret_cast<const XYZZ*>(xyzz);
    XYZZ acc = xyzz_identity();
    for (size_t i = 0; i < count; i++) xyzz_add(acc, p[i]);
    xyzz_to_jacobian_host(acc, out_jac);
    return ZK_OK;
}

// fixed-base batch multiplication ------------------------------------------------------------------
struct GTab { void* d = nullptr; };
static std::map<zk_ctx*, GTab> g_gtabs;
static std::mutex g_gtab_mu;

static int ensure_gtab(zk_ctx* ctx, void** out) {
    std::lock_guard<std::mutex> lk(g_gtab_mu);
    GTab& g = g_gtabs[ctx];
    if (!g.d) {
        DevTmp t_row, t_tab;
        ZK_HIP(hipMalloc(&t_row.p, 255 * 128));
        ZK_HIP(hipMalloc(&t_tab.p, (size_t)32 * 255 * 64));
        void* const row = t_row.p; void* const tab = t_tab.p;
        ZK_LAUNCH(g1_gtab_row0_kernel, 1, 256, 0, ctx->stream, row);
        ZK_CHECK_LAUNCH();
        for (int w = 0; w < 32; w++) {
            if (w) { ZK_LAUNCH(g1_dbl_times_kernel, 1, 256, 0, ctx->stream, row, 255u, 8); ZK_CHECK_LAUNCH(); }
            ZK_LAUNCH(g1_batch_to_affine_kernel, 1, 64, 0, ctx->stream, (const void*)row, 255u, 4u, (void*)((char*)tab + (size_t)w * 255 * 64));
            ZK_CHECK_LAUNCH();
        }
        ZK_HIP(hipStreamSynchronize(ctx->stream));
        g.d = t_tab.release();
    }
    *out = g.d;
    return ZK_OK;
}
void release_gtab(zk_ctx* ctx) {
    std::lock_guard<std::mutex> lk(g_gtab_mu);
    auto it = g_gtabs.find(ctx);
    if (it != g_gtabs.end()) { if (it->second.d) (void)hipFree(it->second.d); g_gtabs.erase(it); }
}

int g1_fixed_base_mul(zk_ctx* ctx, const void* d_scalars, size_t n, void* d_out_affine) {
    if (!d_scalars || !d_out_affine) return ctx->fail(ZK_ERR_ARG, "zk_g1_fixed_base_mul_dev: null pointer");
    if (n == 0) return ZK_OK;
    if (n >= (1ull << 31)) return ctx->fail(ZK_ERR_LIMIT, "zk_g1_fixed_base_mul_dev: n too large");
    void* gtab = nullptr;
    int rc = ensure_gtab(ctx, &gtab);
    if (rc) return rc;
    ZK_HIP(ctx->ws_pts.ensure(n * 128));
    const int blk = ctx->tune.msm_block;
    ZK_LAUNCH(g1_fixed_base_kernel, (uint32_t)((n + blk - 1) / blk), blk, 0, ctx->stream, d_scalars, (uint32_t)n, (const void*)gtab, ctx->ws_pts.p);
    ZK_CHECK_LAUNCH();
    const uint32_t chunk = 32;
    ZK_LAUNCH(g1_batch_to_affine_kernel, (uint32_t)(((n + chunk - 1) / chunk + blk - 1) / blk), blk, 0, ctx->stream, (const void*)ctx->ws_pts.p,
              (uint32_t)n, chunk, d_out_affine);
    ZK_CHECK_LAUNCH();
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

int msm_set_lds_attr() {
#ifndef ZK_EMU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msm_partition_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msm_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msm_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
    return 0;
}

}  // namespace zk
