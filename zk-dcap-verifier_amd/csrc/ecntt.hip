// NTT over G1 ("EC-FFT") — the group-valued instance of halo2_proofs (zkwebauthn @ c254c75) src/arithmetic.rs best_fft
// that ParamsKZG::setup uses for g_to_lagrange (src/poly/kzg/commitment.rs; reached from gen_srs,
// circuits/src/sgx_dcap_verifier.rs:799).  SURVEY.md §8 row a3 / §8(f) "next 3": one-time setup work.
//
// out[j] = [scale] * sum_i [omega^(i j)] P_i, natural order in and out.  Butterflies multiply a point by a 254-bit
// twiddle (double-and-add, ~4000 field products), so the transform is entirely integer-ALU bound and the simplest
// mapping is the right one: bit-reverse once, then log n radix-2 sweeps over XYZZ points in HBM, one butterfly per thread.
#include "ctx.h"

namespace zk {

int ntt_pow_tables(zk_ctx* ctx, uint32_t log_n, const u256& omega, const void** lo, const void** hi, uint32_t* lo_bits);

ZK_HD XYZZ xyzz_neg(const XYZZ& p) {
    XYZZ o = p;
    o.y = Fq::neg(p.y);
    return o;
}
// [k] P for a canonical (non-Montgomery) scalar k
ZK_HD XYZZ xyzz_scalar_mul(const XYZZ& p, const u256& k) {
    XYZZ acc = xyzz_identity();
    int top = 255;
    while (top >= 0 && !((k.v[top >> 5] >> (top & 31)) & 1)) top--;
    for (int b = top; b >= 0; b--) {
        acc = xyzz_dbl(acc);
        if ((k.v[b >> 5] >> (b & 31)) & 1) xyzz_add(acc, p);
    }
    return acc;
}
ZK_HD uint32_t ec_bitrev(uint32_t x, uint32_t bits) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
ZK_KERNEL void ecntt_load_kernel(const void* affine_in, uint32_t log_n, void* pts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << log_n)) return;
    store_xyzz(pts, ec_bitrev(i, log_n), xyzz_from_affine(load_affine(affine_in, i)));
}
// DIT stage s (half = 2^s): (a, b) -> (a + [w] b, a - [w] b), w = omega^(pos << (log_n - 1 - s))
ZK_KERNEL void ecntt_stage_kernel(void* pts, uint32_t log_n, uint32_t s, const void* tw_lo, const void* tw_hi, uint32_t lo_bits) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (1u << (log_n - 1))) return;
    const uint32_t half = 1u << s, grp = t >> s, pos = t & (half - 1);
    const uint32_t i0 = (grp << (s + 1)) + pos, i1 = i0 + half;
    XYZZ a = load_xyzz(pts, i0), b = load_xyzz(pts, i1);
    if (pos) {
        const uint32_t ex = pos << (log_n - 1 - s);
        u256 w = load_u256(tw_lo, ex & ((1u << lo_bits) - 1u));
        const uint32_t h = ex >> lo_bits;
        if (h) w = Fr::mul(w, load_u256(tw_hi, h));
        b = xyzz_scalar_mul(b, Fr::from_mont(w));
    }
    XYZZ sum = a, diff = a;
    xyzz_add(sum, b);
    xyzz_add(diff, xyzz_neg(b));
    store_xyzz(pts, i0, sum);
    store_xyzz(pts, i1, diff);
}
ZK_KERNEL void ecntt_scale_kernel(void* pts, uint32_t n, u256 scale_canonical) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_xyzz(pts, i, xyzz_scalar_mul(load_xyzz(pts, i), scale_canonical));
}
ZK_KERNEL void g1_batch_to_affine_kernel(const void* in, uint32_t n, uint32_t chunk, void* out);   // msm.hip

int g1_ntt(zk_ctx* ctx, const void* d_affine_in, uint32_t log_n, const void* omega_host, const void* scale_host, void* d_affine_out) {
    if (!d_affine_in || !d_affine_out || !omega_host) return ctx->fail(ZK_ERR_ARG, "zk_g1_ntt_dev: null argument");
    if (log_n > 24) return ctx->fail(ZK_ERR_LIMIT, "zk_g1_ntt_dev: log_n = %u > 24", log_n);
    const uint32_t n = 1u << log_n;
    u256 omega;
    memcpy(&omega, omega_host, 32);
    ZK_HIP(ctx->ws_pts.ensure((size_t)n * 128));
    const int blk = ctx->tune.msm_block;
    hipStream_t st = ctx->stream;
    ZK_LAUNCH(ecntt_load_kernel, (n + blk - 1) / blk, blk, 0, st, d_affine_in, log_n, ctx->ws_pts.p);
    ZK_CHECK_LAUNCH();
    if (log_n) {
        const void *lo, *hi;
        uint32_t lo_bits;
        int rc = ntt_pow_tables(ctx, log_n, omega, &lo, &hi, &lo_bits);
        if (rc) return rc;
        const uint32_t nb = n / 2;
        for (uint32_t s = 0; s < log_n; s++) {
            ZK_LAUNCH(ecntt_stage_kernel, (nb + blk - 1) / blk, blk, 0, st, ctx->ws_pts.p, log_n, s, lo, hi, lo_bits);
            ZK_CHECK_LAUNCH();
        }
    }
    if (scale_host) {
        u256 sc;
        memcpy(&sc, scale_host, 32);
        ZK_LAUNCH(ecntt_scale_kernel, (n + blk - 1) / blk, blk, 0, st, ctx->ws_pts.p, n, Fr::from_mont(sc));
        ZK_CHECK_LAUNCH();
    }
    const uint32_t chunk = n >= 32 ? 32 : 1;
    ZK_LAUNCH(g1_batch_to_affine_kernel, (uint32_t)(((n + chunk - 1) / chunk + blk - 1) / blk), blk, 0, st, (const void*)ctx->ws_pts.p, n, chunk, d_affine_out);
    ZK_CHECK_LAUNCH();
    ZK_HIP(hipStreamSynchronize(st));
    return ZK_OK;
}

}  // namespace zk
