// Evaluation phase of create_proof — SURVEY.md §8(f) "next 2".
//
// Replaces halo2_proofs (zkwebauthn @ c254c75, Cargo.lock:1314-1327) src/arithmetic.rs
//   eval_polynomial(poly, point)        Horner over n coefficients (one call per (polynomial, rotation) query)
//   kate_division(a, b)                 quotient of a(X) by (X - b)   (SHPLONK / multiopen)
// Both are serial recurrences on the CPU.  Here eval is a strided Horner per thread (coalesced loads, every
// thread walks the polynomial in x^T) + an LDS tree sum, and kate_division is a 3-phase scan of the linear
// recurrence q[i-1] = a[i] + b*q[i] carried as (value, multiplier) pairs.
#include "ctx.h"
#include <vector>

namespace zk {

constexpr uint32_t PE_T = 256;      // threads per workgroup
constexpr uint32_t PE_E = 126;      // coefficients per thread per workgroup (the per-workgroup x^T and x^tid powers cost ~22 products per thread: 32 made them 40 % of the kernel)

ZK_HD u256 fr_pow_u32(u256 base, uint32_t e) {
    u256 acc = Fr::one();
    for (int b = 31; b >= 0; b--) {
        acc = Fr::sqr(acc);
        if ((e >> b) & 1) acc = Fr::mul(acc, base);
    }
    return acc;
}

__device__ __forceinline__ u256 pe_block_sum(u256 v) {
    __shared__ uint4 slo[PE_T], shi[PE_T];
    const uint32_t tid = threadIdx.x;
    slo[tid] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
    shi[tid] = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
    __syncthreads();
    for (uint32_t d = blockDim.x >> 1; d > 0; d >>= 1) {
        if (tid < d) {
            uint4 l = slo[tid + d], h = shi[tid + d];
            u256 o;
            o.v[0] = l.x; o.v[1] = l.y; o.v[2] = l.z; o.v[3] = l.w; o.v[4] = h.x; o.v[5] = h.y; o.v[6] = h.z; o.v[7] = h.w;
            v = Fr::add(v, o);
            slo[tid] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
            shi[tid] = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
        }
        __syncthreads();
    }
    return v;   // valid in thread 0
}

// partial[q][blk] = sum over the workgroup's span of c[i] * x_q^(i - span_start)
ZK_KERNEL void pe_eval_partial_kernel(const void* const* polys, const void* points, uint32_t n, void* partial) {
    const uint32_t q = blockIdx.y, T = blockDim.x, tid = threadIdx.x;
    const void* poly = polys[q];
    const u256 x = load_u256(points, q);
    const uint32_t span = T * PE_E, start = blockIdx.x * span;
    u256 xT = x;                                   // x^T, T a power of two
    for (uint32_t t = T; t > 1; t >>= 1) xT = Fr::sqr(xT);
    // Horner in x^T over c[start + tid + e*T], SIX coefficients per step on carry-free limbs (field29.cuh, sums of products): acc * x^6T + c5 x^5T + .. + c1 x^T with ONE
    // reduction for the six products, c0 added as it is; the powers are the same in every thread (scalar registers).  The product with x^tid below is a full one and
    // returns the canonical value.
    static_assert(PE_E % 6 == 0, "six coefficients per Horner step");
    __shared__ uint32_t X[6][12];                  // x^T .. x^6T as x 2^261 operands (exact limbs): the same in every thread, read back from LDS (a broadcast) at each use
    if (tid < 6) {
        u256 c32 = Fr::zero();
        c32.v[0] = 32;
        u256 pw = Fr::mul(xT, Fr::to_mont(c32));   // x^T * 32
        for (uint32_t k = 0; k < tid; k++) pw = Fr::mul(pw, xT);
        const u261 t = Fr29::from32<0>(pw);
        for (int i = 0; i < 9; i++) X[tid][i] = t.l[i];
    }
    __syncthreads();
    auto xp = [&](int k) -> u261 { u261 o; for (int i = 0; i < 9; i++) o.l[i] = X[k][i]; return o; };
    u261 acc = Fr29::zero();
    auto coef = [&](uint32_t e) -> u261 { const uint32_t idx = start + tid + e * T; return idx < n ? Fr29::from32<0>(load_u256(poly, idx)) : Fr29::zero(); };
    for (int e = (int)PE_E - 6; e >= 0; e -= 6) {
        Fr29::Dot29 d;
        Fr29::dot_clear(d);
        Fr29::dot_add(d, acc, xp(5));
#pragma unroll 1
        for (int k = 5; k >= 1; k--) Fr29::dot_add(d, coef((uint32_t)e + k), xp(k - 1));
        acc = Fr29::carry(Fr29::add(Fr29::dot_reduce(d), coef((uint32_t)e)));       // below 2.05 p, N-form
    }
    u256 acc32 = Fr::mul(Fr29::to32(acc), fr_pow_u32(x, tid));
    acc32 = pe_block_sum(acc32);
    if (tid == 0) store_u256(partial, (size_t)q * gridDim.x + blockIdx.x, acc32);
}
// out[q] = sum_b partial[q][b] * (x^span)^b
ZK_KERNEL void pe_eval_final_kernel(const void* partial, const void* points, uint32_t nblk, uint32_t span, void* out) {
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const u256 xs = fr_pow_u32(load_u256(points, q), span);
    u256 acc = Fr::zero();
    for (uint32_t b = tid; b < nblk; b += blockDim.x) acc = Fr::add(acc, Fr::mul(load_u256(partial, (size_t)q * nblk + b), fr_pow_u32(xs, b)));
    acc = pe_block_sum(acc);
    if (tid == 0) store_u256(out, q, acc);
}

// ---- kate_division ---------------------------------------------------------------------------------
// With r[t] = a[n-1-t]:  Q[t] = r[t] + b*Q[t-1],  q[n-2-t] = Q[t]  (t <= n-2).
constexpr uint32_t KD_E = 8;
__device__ __forceinline__ void kd_lds_put(uint4* lo, uint4* hi, uint32_t i, const u256& v) {
    lo[i] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]); hi[i] = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
}
__device__ __forceinline__ u256 kd_lds_get(const uint4* lo, const uint4* hi, uint32_t i) {
    uint4 l = lo[i], h = hi[i];
    u256 o;
    o.v[0] = l.x; o.v[1] = l.y; o.v[2] = l.z; o.v[3] = l.w; o.v[4] = h.x; o.v[5] = h.y; o.v[6] = h.z; o.v[7] = h.w;
    return o;
}
// mode 0: write the workgroup's (total, b^span) pair;  mode 1: carries[blk] holds the carry-in, write q.
ZK_KERNEL void kd_scan_kernel(const void* a, uint32_t n, u256 b, int mode, void* carries, void* q) {
    __shared__ uint4 vlo[PE_T], vhi[PE_T], mlo[PE_T], mhi[PE_T];
    const uint32_t tid = threadIdx.x, T = blockDim.x;
    const uint32_t span = T * KD_E, t0 = blockIdx.x * span + tid * KD_E;
    u256 L[KD_E];
    u256 run = Fr::zero();
#pragma unroll
    for (uint32_t e = 0; e < KD_E; e++) {
        const uint32_t t = t0 + e;
        run = Fr::mul(run, b);
        if (t < n) run = Fr::add(run, load_u256(a, n - 1 - t));
        L[e] = run;
    }
    u256 bE = b;                                            // b^KD_E
    for (uint32_t s = KD_E; s > 1; s >>= 1) bE = Fr::sqr(bE);
    // inclusive scan of (value, multiplier) pairs across the threads:  (v1,m1) then (v2,m2)  ->  (v2 + m2*v1, m1*m2)
    u256 val = run, mul = bE;
    kd_lds_put(vlo, vhi, tid, val); kd_lds_put(mlo, mhi, tid, mul);
    __syncthreads();
    for (uint32_t d = 1; d < T; d <<= 1) {
        u256 pv = Fr::zero(), pm = Fr::one();
        const bool act = tid >= d;
        if (act) { pv = kd_lds_get(vlo, vhi, tid - d); pm = kd_lds_get(mlo, mhi, tid - d); }
        __syncthreads();
        if (act) {
            val = Fr::add(val, Fr::mul(mul, pv));
            mul = Fr::mul(mul, pm);
            kd_lds_put(vlo, vhi, tid, val); kd_lds_put(mlo, mhi, tid, mul);
        }
        __syncthreads();
    }
    if (mode == 0) {
        if (tid == T - 1) { store_u256(carries, 2 * (size_t)blockIdx.x, val); store_u256(carries, 2 * (size_t)blockIdx.x + 1, mul); }
        return;
    }
    // carry into this thread = (inclusive value of the previous thread) + (its multiplier) * (workgroup carry-in)
    const u256 cin = load_u256(carries, blockIdx.x);
    u256 carry = cin;
    if (tid > 0) carry = Fr::add(kd_lds_get(vlo, vhi, tid - 1), Fr::mul(kd_lds_get(mlo, mhi, tid - 1), cin));
    u256 bp = b;
#pragma unroll
    for (uint32_t e = 0; e < KD_E; e++) {
        const uint32_t t = t0 + e;
        if (t + 1 < n) store_u256(q, n - 2 - t, Fr::add(L[e], Fr::mul(bp, carry)));
        bp = Fr::mul(bp, b);
    }
}
// single workgroup: carries[2*blk] = (total, mult) pairs -> out[blk] = carry-in of workgroup blk: the same (value, multiplier) scan as
// inside kd_scan_kernel, one pair per thread per round of PE_T workgroups (a serial walk over a few hundred dependent products cost 0.2 ms)
ZK_KERNEL void kd_carry_kernel(void* carries, uint32_t nblk, void* out) {
    __shared__ uint4 vlo[PE_T], vhi[PE_T], mlo[PE_T], mhi[PE_T];
    const uint32_t tid = threadIdx.x, T = blockDim.x;
    u256 cin = Fr::zero();                                  // carry into the current round of T workgroups
    for (uint32_t base = 0; base < nblk; base += T) {
        const uint32_t b = base + tid;
        u256 val = Fr::zero(), mul = Fr::one();
        if (b < nblk) { val = load_u256(carries, 2 * (size_t)b); mul = load_u256(carries, 2 * (size_t)b + 1); }
        kd_lds_put(vlo, vhi, tid, val); kd_lds_put(mlo, mhi, tid, mul);
        __syncthreads();
        for (uint32_t d = 1; d < T; d <<= 1) {
            u256 pv = Fr::zero(), pm = Fr::one();
            const bool act = tid >= d;
            if (act) { pv = kd_lds_get(vlo, vhi, tid - d); pm = kd_lds_get(mlo, mhi, tid - d); }
            __syncthreads();
            if (act) {
                val = Fr::add(val, Fr::mul(mul, pv));
                mul = Fr::mul(mul, pm);
                kd_lds_put(vlo, vhi, tid, val); kd_lds_put(mlo, mhi, tid, mul);
            }
            __syncthreads();
        }
        // exclusive: carry-in of workgroup b = inclusive(b - 1) applied to cin
        u256 c = cin;
        if (tid > 0) c = Fr::add(kd_lds_get(vlo, vhi, tid - 1), Fr::mul(kd_lds_get(mlo, mhi, tid - 1), cin));
        if (b < nblk) store_u256(out, b, c);
        const u256 last_v = kd_lds_get(vlo, vhi, T - 1), last_m = kd_lds_get(mlo, mhi, T - 1);
        cin = Fr::add(last_v, Fr::mul(last_m, cin));
        __syncthreads();
    }
}

// ---- linear combinations (SHPLONK / multiopen polynomial combos) -----------------------------------
// out[i] = sum_j s_j * p_j[i]: the `poly * power_of_y` ... `reduce(|acc, poly| acc + &poly)` chains of
// halo2_proofs src/poly/kzg/multiopen/shplonk/prover.rs, one pass over all inputs instead of one pass per term.
// args: [count pointers | pad to 32 B | count scalars]
ZK_KERNEL void pe_lincomb_kernel(const void* const* polys, const void* scalars32, uint32_t count, size_t n, void* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // terms are taken SIX at a time on carry-free limbs with ONE Montgomery reduction for the six products (field29.cuh, sums of products: 96 multiplications per term against
    // 171); scalars32: the scalars times 32 (x 2^261 operands: value * scalar * 2^-261 stays in the library's form), the same in every thread.  Chunk sums are added with a
    // carry round; the element is reduced and normalised when it is stored.
    for (; i < n; i += stride) {
        u261 total = Fr29::zero();
        uint32_t chunks = 0;
        for (uint32_t j0 = 0; j0 < count; j0 += 6) {
            Fr29::Dot29 d;
            Fr29::dot_clear(d);
            const uint32_t j1 = j0 + 6 < count ? j0 + 6 : count;
            for (uint32_t j = j0; j < j1; j++) Fr29::dot_add(d, Fr29::from32<0>(load_u256(polys[j], i)), Fr29::from32<0>(load_u256(scalars32, j)));
            total = Fr29::carry(Fr29::add(total, Fr29::dot_reduce(d)));              // every chunk adds less than 1.3 p
            if ((++chunks & 15u) == 0) total = Fr29::reduce_small(total);            // (below 32 p at all times)
        }
        store_u256(out, i, Fr::normalize(Fr29::to32(Fr29::reduce_small(total))));
    }
}

// ---- host ------------------------------------------------------------------------------------------
int eval_polynomial_batch(zk_ctx* ctx, const void* const* polys, size_t count, size_t n, const void* points, void* out) {
    if (!polys || !points || !out) return ctx->fail(ZK_ERR_ARG, "zk_eval_polynomial_batch_dev: null argument");
    if (count == 0) return ZK_OK;
    if (n == 0 || n > (1u << 27) || count > 65535) return ctx->fail(ZK_ERR_LIMIT, "zk_eval_polynomial_batch_dev: n or count out of range");
    for (size_t i = 0; i < count; i++) if (!polys[i]) return ctx->fail(ZK_ERR_ARG, "zk_eval_polynomial_batch_dev: null polynomial %zu", i);
    const uint32_t span = PE_T * PE_E, nblk = (uint32_t)((n + span - 1) / span);
    const size_t off_pts = (count * sizeof(void*) + 31) & ~(size_t)31, off_part = off_pts + count * 32, off_out = off_part + count * (size_t)nblk * 32;
    ZK_HIP(ctx->ws_tmp.ensure(off_out + count * 32));
    char* base = (char*)ctx->ws_tmp.p;
    hipStream_t st = ctx->stream;
    ZK_HIP(hipMemcpyAsync(base, polys, count * sizeof(void*), hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(base + off_pts, points, count * 32, hipMemcpyHostToDevice, st));
    ZK_LAUNCH(pe_eval_partial_kernel, dim3(nblk, (uint32_t)count), PE_T, 0, st, (const void* const*)base, (const void*)(base + off_pts), (uint32_t)n, (void*)(base + off_part));
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(pe_eval_final_kernel, (uint32_t)count, PE_T, 0, st, (const void*)(base + off_part), (const void*)(base + off_pts), nblk, span, (void*)(base + off_out));
    ZK_CHECK_LAUNCH();
    ZK_HIP(hipMemcpyAsync(out, base + off_out, count * 32, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    return ZK_OK;
}

int kate_division(zk_ctx* ctx, const void* d_a, size_t n, const void* b_host, void* d_q) {
    if (!d_a || !b_host || !d_q) return ctx->fail(ZK_ERR_ARG, "zk_kate_division_dev: null argument");
    if (n < 2 || n > (1u << 27)) return ctx->fail(ZK_ERR_ARG, "zk_kate_division_dev: n = %zu out of range", n);
    u256 b;
    memcpy(&b, b_host, 32);
    const uint32_t span = PE_T * KD_E, nblk = (uint32_t)((n + span - 1) / span);
    ZK_HIP(ctx->ws_tmp.ensure((size_t)nblk * 96 + 64));
    void* pairs = ctx->ws_tmp.p;
    void* cin = (char*)ctx->ws_tmp.p + (size_t)nblk * 64;
    hipStream_t st = ctx->stream;
    ZK_LAUNCH(kd_scan_kernel, nblk, PE_T, 0, st, d_a, (uint32_t)n, b, 0, pairs, d_q);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(kd_carry_kernel, 1, PE_T, 0, st, pairs, nblk, cin);
    ZK_CHECK_LAUNCH();
    ZK_LAUNCH(kd_scan_kernel, nblk, PE_T, 0, st, d_a, (uint32_t)n, b, 1, cin, d_q);
    ZK_CHECK_LAUNCH();
    ZK_HIP(hipStreamSynchronize(st));
    return ZK_OK;
}

int fr_lincomb(zk_ctx* ctx, const void* const* polys, const void* scalars, size_t count, size_t n, void* d_out) {
    if (!polys || !scalars || !d_out) return ctx->fail(ZK_ERR_ARG, "zk_fr_lincomb_dev: null argument");
    if (count == 0 || count > 65535) return ctx->fail(ZK_ERR_ARG, "zk_fr_lincomb_dev: count out of range");
    if (n == 0) return ZK_OK;
    for (size_t i = 0; i < count; i++) if (!polys[i]) return ctx->fail(ZK_ERR_ARG, "zk_fr_lincomb_dev: null polynomial %zu", i);
    const size_t off_sc = (count * sizeof(void*) + 31) & ~(size_t)31;
    ZK_HIP(ctx->ws_tmp.ensure(off_sc + count * 32));
    char* base = (char*)ctx->ws_tmp.p;
    hipStream_t st = ctx->stream;
    ZK_HIP(hipMemcpyAsync(base, polys, count * sizeof(void*), hipMemcpyHostToDevice, st));
    std::vector<u256> sc32(count);                                   // the kernel multiplies on 29-bit limbs (Montgomery radix 2^261): scalars as x 2^261 operands
    {
        u256 c32 = Fr::zero();
        c32.v[0] = 32;
        c32 = Fr::to_mont(c32);
        for (size_t j = 0; j < count; j++) { u256 v; memcpy(&v, (const char*)scalars + 32 * j, 32); sc32[j] = Fr::mul(v, c32); }
    }
    ZK_HIP(hipMemcpyAsync(base + off_sc, sc32.data(), count * 32, hipMemcpyHostToDevice, st));
    const int blk = ctx->tune.vec_block;
    size_t grid = (n + blk - 1) / blk; if (grid > 8192) grid = 8192;
    ZK_LAUNCH(pe_lincomb_kernel, (uint32_t)grid, blk, 0, st, (const void* const*)base, (const void*)(base + off_sc), (uint32_t)count, n, d_out);
    ZK_CHECK_LAUNCH();
    ZK_HIP(hipStreamSynchronize(st));
    return ZK_OK;
}

}  // namespace zk
