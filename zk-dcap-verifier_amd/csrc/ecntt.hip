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

// ---- G1 point encoding (ParamsKZG::{read, write}: the on-disk SRS params/kzg_bn254_{k}.srs) ---------------
// halo2curves G1Affine::{to_bytes, from_bytes}: 32-byte little-endian canonical x with the parity of y in a flag bit —
//   sign_bit = 255: halo2curves 0.3.1 (stack A, pasta-style; identity = all zero bytes)            [3P-MEM]
//   sign_bit = 254: halo2curves-axiom 0.5.2 (stack B; bit 255 flags the identity) — the convention bin/assets/proof.bin shows (App. B)
// from_bytes needs a square root per point: y = (x^3 + 3)^((p+1)/4) (p = 3 mod 4), 2^k of them per table when an SRS file is loaded.
namespace zk {

ZK_KERNEL void g1_decompress_kernel(const void* bytes, uint32_t n, uint32_t sign_bit, void* out_affine, uint32_t* n_bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u256 xc = load_u256(bytes, i);                       // canonical little-endian words
    const uint32_t top = xc.v[7];
    const bool sign = (top >> (sign_bit - 224)) & 1u;
    const bool id_flag = sign_bit == 254 ? (top >> 31) & 1u : false;
    xc.v[7] &= sign_bit == 254 ? 0x3fffffffu : 0x7fffffffu;
    Affine o;
    o.x = Fq::zero(); o.y = Fq::zero();
    bool zero = true;
#pragma unroll
    for (int w = 0; w < 8; w++) zero = zero && xc.v[w] == 0;
    if (id_flag || (zero && !sign && sign_bit == 255)) { store_affine(out_affine, i, o); return; }   // identity
    bool lt = false;                                     // canonical: x < p
    for (int w = 7; w >= 0; w--) { if (xc.v[w] != Fq::p(w)) { lt = xc.v[w] < Fq::p(w); break; } }
    if (!lt) { atomicAdd(n_bad, 1u); store_affine(out_affine, i, o); return; }
    const u256 x = Fq::to_mont(xc);
    const uint64_t three[4] = BN254_FQ_THREE_M;
    u256 b3;
#pragma unroll
    for (int w = 0; w < 8; w++) b3.v[w] = (uint32_t)(three[w >> 1] >> (32 * (w & 1)));
    const u256 rhs = Fq::add(Fq::mul(Fq::sqr(x), x), b3);
    u256 e;                                              // (p + 1) / 4
    {
        uint32_t carry = 1;
        u256 t;
        for (int w = 0; w < 8; w++) { const uint64_t sum = (uint64_t)Fq::p(w) + carry; t.v[w] = (uint32_t)sum; carry = (uint32_t)(sum >> 32); }
        for (int w = 0; w < 8; w++) e.v[w] = (t.v[w] >> 2) | (w < 7 ? t.v[w + 1] << 30 : 0u);
    }
    u256 y = Fq::pow(rhs, e);
    if (!Fq::eq(Fq::sqr(y), rhs)) { atomicAdd(n_bad, 1u); store_affine(out_affine, i, o); return; }   // x is not on the curve
    const u256 yc = Fq::from_mont(y);
    if ((bool)(yc.v[0] & 1u) != sign) y = Fq::neg(y);
    o.x = x; o.y = y;
    store_affine(out_affine, i, o);
}
ZK_KERNEL void g1_compress_kernel(const void* affine, uint32_t n, uint32_t sign_bit, void* bytes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Affine p = load_affine(affine, i);
    u256 o = Fq::zero();
    if (affine_is_identity(p)) {
        if (sign_bit == 254) o.v[7] = 0x80000000u;
    } else {
        o = Fq::from_mont(p.x);
        const u256 yc = Fq::from_mont(p.y);
        o.v[7] |= (yc.v[0] & 1u) << (sign_bit - 224);
    }
    store_u256(bytes, i, o);
}

int g1_decompress(zk_ctx* ctx, const void* d_bytes, size_t n, uint32_t sign_bit, void* d_out_affine, uint32_t* n_bad_host) {
    if (!d_bytes || !d_out_affine) return ctx->fail(ZK_ERR_ARG, "zk_g1_decompress_dev: null pointer");
    if (sign_bit != 255 && sign_bit != 254) return ctx->fail(ZK_ERR_ARG, "zk_g1_decompress_dev: sign_bit must be 255 (halo2curves 0.3) or 254 (halo2curves-axiom 0.5)");
    if (n == 0) { if (n_bad_host) *n_bad_host = 0; return ZK_OK; }
    if (n >= (1ull << 31)) return ctx->fail(ZK_ERR_LIMIT, "zk_g1_decompress_dev: n too large");
    ZK_HIP(ctx->ws_tmp.ensure(64));
    uint32_t* d_bad = (uint32_t*)ctx->ws_tmp.p;
    ZK_HIP(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    const int blk = ctx->tune.msm_block;
    ZK_LAUNCH(g1_decompress_kernel, (uint32_t)((n + blk - 1) / blk), blk, 0, ctx->stream, d_bytes, (uint32_t)n, sign_bit, d_out_affine, d_bad);
    ZK_CHECK_LAUNCH();
    uint32_t bad = 0;
    ZK_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    if (n_bad_host) *n_bad_host = bad;
    if (bad) return ctx->fail(ZK_ERR_ARG, "zk_g1_decompress_dev: %u encodings are not points of y^2 = x^3 + 3 (non-canonical x or no square root)", bad);
    return ZK_OK;
}
int g1_compress(zk_ctx* ctx, const void* d_affine, size_t n, uint32_t sign_bit, void* d_bytes) {
    if (!d_bytes || !d_affine) return ctx->fail(ZK_ERR_ARG, "zk_g1_compress_dev: null pointer");
    if (sign_bit != 255 && sign_bit != 254) return ctx->fail(ZK_ERR_ARG, "zk_g1_compress_dev: sign_bit must be 255 or 254");
    if (n == 0) return ZK_OK;
    if (n >= (1ull << 31)) return ctx->fail(ZK_ERR_LIMIT, "zk_g1_compress_dev: n too large");
    const int blk = ctx->tune.msm_block;
    ZK_LAUNCH(g1_compress_kernel, (uint32_t)((n + blk - 1) / blk), blk, 0, ctx->stream, d_affine, (uint32_t)n, sign_bit, d_bytes);
    ZK_CHECK_LAUNCH();
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

}  // namespace zk
