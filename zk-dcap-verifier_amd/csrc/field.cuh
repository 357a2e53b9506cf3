// BN254 Fq / Fr arithmetic for gfx950 (CDNA4): 8 x 32-bit limbs, Montgomery form, R = 2^256.
//
// Memory layout is exactly halo2curves::bn256::{Fq,Fr} ([u64;4] little endian ==
// [u32;8] little endian on this target), so buffers cross the FFI with zero conversion
// (halo2curves 0.3.1 @ bdb2e66, Cargo.lock:1329-1344).
//
// CDNA4 has no 64x64 multiplier in the VALU; the widest integer multiply is v_mad_u64_u32
// (32x32+64 -> 64, measured at ~1/3 of the full VALU rate on MI355X, profiles/r01).  Field::mul is a
// product-scanning Montgomery multiplication built directly on that instruction and its carry-out.
#pragma once
#include "rt.h"
#include <stdint.h>
#include "bn254_consts.h"

#define ZK_HD __host__ __device__ __forceinline__
// host passes of the product build only (not device code, not the CPU emulator of the kernels): see Field::mul_host64
#if !defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_EMU) && defined(__SIZEOF_INT128__)
#define ZK_HOST64 1
#else
#define ZK_HOST64 0
#endif

namespace zk {

struct alignas(16) u256 {
    uint32_t v[8];
};

struct FqParams {
    static constexpr uint64_t P[4] = BN254_FQ_MODULUS;
    static constexpr uint64_t R[4] = BN254_FQ_R;
    static constexpr uint64_t R2[4] = BN254_FQ_R2;
    static constexpr uint32_t INV = BN254_FQ_INV32;
};
struct FrParams {
    static constexpr uint64_t P[4] = BN254_FR_MODULUS;
    static constexpr uint64_t R[4] = BN254_FR_R;
    static constexpr uint64_t R2[4] = BN254_FR_R2;
    static constexpr uint32_t INV = BN254_FR_INV32;
};

#include "field_mac.inc"

template <class FP>
struct Field {
    // compile-time limb accessors (fold to literals once loops are unrolled)
    static ZK_HD constexpr uint32_t p(int i) { return (uint32_t)(FP::P[i >> 1] >> (32 * (i & 1))); }
    static ZK_HD constexpr uint32_t r(int i) { return (uint32_t)(FP::R[i >> 1] >> (32 * (i & 1))); }
    static ZK_HD constexpr uint32_t r2(int i) { return (uint32_t)(FP::R2[i >> 1] >> (32 * (i & 1))); }

    static ZK_HD u256 zero() {
        u256 o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.v[i] = 0;
        return o;
    }
    static ZK_HD u256 one() {  // Montgomery 1
        u256 o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.v[i] = r(i);
        return o;
    }
    static ZK_HD u256 R2() {
        u256 o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.v[i] = r2(i);
        return o;
    }
    static ZK_HD bool is_zero(const u256& a) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= a.v[i];
        return o == 0;
    }
    static ZK_HD bool eq(const u256& a, const u256& b) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
        return o == 0;
    }

    // r = a - p if a >= p else a     (a < 2p)
    static ZK_HD u256 reduce_once(const u256& a) {
        u256 d;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(a.v[i], p(i), br, &br);
        u256 o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.v[i] = br ? a.v[i] : d.v[i];
        return o;
    }
    static ZK_HD u256 add(const u256& a, const u256& b) {
        u256 s;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
        return reduce_once(s);  // p < 2^254 => a + b < 2^255, no carry out
    }
    static ZK_HD u256 sub(const u256& a, const u256& b) {
        u256 d;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(a.v[i], b.v[i], br, &br);
        uint32_t mask = 0u - br;  // add p back if we borrowed
        uint32_t c = 0;
        u256 o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.v[i] = __builtin_addc(d.v[i], p(i) & mask, c, &c);
        return o;
    }
    static ZK_HD u256 neg(const u256& a) {
        u256 d;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(p(i), a.v[i], br, &br);
        bool z = is_zero(a);
        u256 o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.v[i] = z ? 0u : d.v[i];
        return o;
    }
    static ZK_HD u256 dbl(const u256& a) { return add(a, a); }

    // ---- redundant ranges (the NTT butterflies, Harvey's scheme).  p < 2^254, so values may live in [0, 4p) inside 256 bits: a butterfly then needs ONE
    // conditional correction instead of three (product, sum, difference).  Every function states the ranges it takes and gives. ----
    static ZK_HD constexpr uint32_t p2(int i) { return (p(i) << 1) | (i ? p(i - 1) >> 31 : 0u); }      // limbs of 2p
    // a in [0, 4p) -> a mod 2p in [0, 2p)
    static ZK_HD u256 red2p(const u256& a) {
        u256 d;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(a.v[i], p2(i), br, &br);
        u256 o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.v[i] = br ? a.v[i] : d.v[i];
        return o;
    }
    // a, b in [0, 2p) -> a + b in [0, 4p)   (no correction)
    static ZK_HD u256 add_lazy(const u256& a, const u256& b) {
        u256 s;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
        return s;
    }
    // a, b in [0, 2p) -> a - b + 2p in (0, 4p)   (no correction)
    static ZK_HD u256 sub_lazy(const u256& a, const u256& b) {
        u256 s;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s.v[i] = __builtin_addc(a.v[i], p2(i), c, &c);
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s.v[i] = __builtin_subc(s.v[i], b.v[i], br, &br);
        return s;
    }
    // a in [0, 4p), b in [0, p] -> a*b*R^-1 in [0, 2p): the Montgomery product without its final subtraction ((4p*p + 2^256 p) / 2^256 < 1.76 p)
    static ZK_HD u256 mul_lazy(const u256& a, const u256& b) {
        uint64_t acc = 0;
        uint32_t cnt = 0;
        uint32_t m[8];
        u256 r;
#include "field_mul_body.inc"
        r.v[7] = (uint32_t)acc;
        return r;
    }
    // a in [0, 4p) -> the canonical representative in [0, p)
    static ZK_HD u256 normalize(const u256& a) { return reduce_once(red2p(a)); }
    // ---- the same idea one notch tighter, for chains of products (the bucket accumulation): values in [0, 2p).  A Montgomery product of two such values is
    // < (4p^2 + 2^256 p) / 2^256 < 1.76 p, i.e. again in [0, 2p) WITHOUT the final subtraction; differences are corrected by 2p instead of p (same cost). ----
    static ZK_HD u256 sqr_lazy(const u256& a) {                       // a in [0, 2p) -> [0, 2p)
        uint64_t acc = 0;
        uint32_t cnt = 0;
        uint32_t m[8];
        u256 r;
#include "field_sqr_body.inc"
        r.v[7] = (uint32_t)acc;
        return r;
    }
    static ZK_HD u256 sub2(const u256& a, const u256& b) {            // a, b in [0, 2p) -> a - b (+ 2p if it borrowed) in [0, 2p)
        u256 d;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(a.v[i], b.v[i], br, &br);
        uint32_t mask = 0u - br;
        uint32_t c = 0;
        u256 o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.v[i] = __builtin_addc(d.v[i], p2(i) & mask, c, &c);
        return o;
    }
    static ZK_HD u256 dbl2(const u256& a) { return red2p(add_lazy(a, a)); }     // a in [0, 2p) -> 2a in [0, 2p)
    static ZK_HD u256 neg2(const u256& a) {                           // a in [0, 2p) -> 2p - a in (0, 2p]
        u256 d;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(p2(i), a.v[i], br, &br);
        return d;
    }
    // a, b, c, d in [0, 2p] -> (a*b + c*d) R^-1 in [0, 2p): (8p^2 + 2^256 p) / 2^256 < 2.52 p, one correction by 2p
    static ZK_HD u256 mul2_add_2p(const u256& a, const u256& b, const u256& c, const u256& d) {
        uint64_t acc = 0;
        uint32_t cnt = 0;
        uint32_t m[8];
        u256 r;
#include "field_mul2_body.inc"
        r.v[7] = (uint32_t)acc;
        return red2p(r);
    }
    static ZK_HD bool is_zero_mod(const u256& a) {                    // a in [0, 2p): a == 0 (mod p)
        uint32_t o = 0, q = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { o |= a.v[i]; q |= a.v[i] ^ p(i); }
        return o == 0 || q == 0;
    }

    // Montgomery product a*b*R^-1 mod p: finely integrated product scanning (Comba columns).  Column k
    // gathers every a_i*b_j and m_i*p_j with i + j = k in a 96-bit accumulator (64-bit register pair +
    // overflow counter): each of the 128 partial products costs one v_mad_u64_u32 (carry-out to VCC)
    // and one v_addc_co_u32 into the counter — see field_mac.inc — and the modulus limbs sit in SGPRs.
    // The quotient digit m_k is formed when column k is complete.  Both moduli are < 2^254, so the
    // 16-limb total divided by 2^256 is < 2p and fits 8 limbs; one conditional subtraction finishes.
    static ZK_HD u256 mul(const u256& a, const u256& b) {
#if ZK_HOST64
        return mul_host64(a, b);
#else
        uint64_t acc = 0;
        uint32_t cnt = 0;
        uint32_t m[8];
        u256 r;
#include "field_mul_body.inc"
        r.v[7] = (uint32_t)acc;
        return reduce_once(r);
#endif
    }
#if ZK_HOST64
    // The product library's HOST code (the folds of the MSM's class sums, point normalisation, domain constants) multiplies on four 64-bit limbs with 128-bit
    // products (CIOS): a third of the time of the 32-bit column scan above, which is written for v_mad_u64_u32.  Same canonical result.  Not in the emulator build,
    // whose point is to run the device code's own arithmetic on the CPU.
    static constexpr uint64_t inv64() {                                // -p^-1 mod 2^64 (Newton from the 32-bit constant)
        uint64_t x = (uint64_t)0 - (uint64_t)FP::INV;                  // p^-1 mod 2^32
        x *= 2 - FP::P[0] * x;                                         // mod 2^64
        return (uint64_t)0 - x;
    }
    static inline u256 mul_host64(const u256& a, const u256& b) {
        typedef unsigned __int128 u128;
        uint64_t A[4], B[4], t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) { A[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32); B[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32); }
        constexpr uint64_t ninv = inv64();
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) { c += (u128)t[j] + (u128)A[j] * B[i]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * ninv;
            c = ((u128)t[0] + (u128)m * FP::P[0]) >> 64;
            for (int j = 1; j < 4; j++) { c += (u128)t[j] + (u128)m * FP::P[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
        }
        u256 r;                                                        // below 2 p (p < 2^254): t[4] = 0
        for (int i = 0; i < 4; i++) { r.v[2 * i] = (uint32_t)t[i]; r.v[2 * i + 1] = (uint32_t)(t[i] >> 32); }
        return reduce_once(r);
    }
#endif
    // (a*b + c*d) * R^-1 mod p with ONE reduction: both products feed the same columns.  a*b + c*d < 2p^2 and p < 2^254,
    // so (a*b + c*d + m*p) / 2^256 < p/2 + p < 2p: still 8 limbs and one conditional subtraction.
    static ZK_HD u256 mul2_add(const u256& a, const u256& b, const u256& c, const u256& d) {
#if ZK_HOST64
        return add(mul_host64(a, b), mul_host64(c, d));
#endif
        uint64_t acc = 0;
        uint32_t cnt = 0;
        uint32_t m[8];
        u256 r;
#include "field_mul2_body.inc"
        r.v[7] = (uint32_t)acc;
        return reduce_once(r);
    }
    // a*b - c*d (as a*b + c*(p - d))
    static ZK_HD u256 mul2_sub(const u256& a, const u256& b, const u256& c, const u256& d) { return mul2_add(a, b, c, neg(d)); }

    // a*a*R^-1 with the 28 cross products taken once (36 operand products instead of 64; the 64 reduction products stay): field_sqr_body.inc
    static ZK_HD u256 sqr(const u256& a) {
#if ZK_HOST64
        return mul_host64(a, a);
#endif
        uint64_t acc = 0;
        uint32_t cnt = 0;
        uint32_t m[8];
        u256 r;
#include "field_sqr_body.inc"
        r.v[7] = (uint32_t)acc;
        return reduce_once(r);
    }

    // a * w mod p for a CONSTANT w (an NTT twiddle): w as its canonical integer, wq = floor(w 2^256 / p) precomputed (Shoup; Harvey's lazy butterflies).  q = floor(a wq / 2^256)
    // is the true quotient of a w by p or up to two below it for any a below 2^256 (one from wq's truncation, one from the columns 0 .. 5 of a * wq that are not computed),
    // so r = a w - q p, read off the low 256 bits of a w + q (2^256 - p), lies in [0, 3p) and one conditional subtraction of 2p returns [0, 2p) — mul_lazy's range, from 115
    // partial products instead of 128 + 8, with NO Montgomery factor: a value in the library's form stays in it.  field_shoup{q,r}_body.inc (tools/gen_mac.py).
    static ZK_HD constexpr uint32_t np(int i) {                        // limb i of 2^256 - p
        uint64_t borrow = 0, limb = 0;
        for (int j = 0; j <= i; j++) {                                 // 0 - p, limb by limb (the minuend 2^256 shows only as the final borrow)
            const uint64_t pj = p(j);
            limb = (0x100000000ull - pj - borrow) & 0xffffffffull;
            borrow = (pj + borrow) ? 1 : 0;
        }
        return (uint32_t)limb;
    }
    // wq = floor(w 2^256 / p) from w's library form w_lib = w 2^256 mod p: w 2^256 = wq p + w_lib, so wq = -w_lib p^-1 mod 2^256 (a low-half product; table builds only)
    struct Limbs8 { uint32_t l[8]; };
    static constexpr Limbs8 pinv256() {                                // p^-1 mod 2^256 by Hensel lifting on 32-bit limbs: x <- x (2 - p x), the precision doubling each step
        uint32_t x[8] = {1, 0, 0, 0, 0, 0, 0, 0};                      // p is odd: correct to 1 bit
        for (int it = 0; it < 9; it++) {
            uint32_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, u[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < 8; i++) {                              // t = p x mod 2^256
                uint64_t c = 0;
                for (int j = 0; i + j < 8; j++) { c += (uint64_t)p(i) * x[j] + t[i + j]; t[i + j] = (uint32_t)c; c >>= 32; }
            }
            uint64_t br = 0;                                           // t = 2 - t
            for (int i = 0; i < 8; i++) { const uint64_t m = (i == 0 ? 2ull : 0ull), s_ = (uint64_t)t[i] + br; t[i] = (uint32_t)(m - s_); br = m < s_ ? 1 : 0; }
            for (int i = 0; i < 8; i++) {                              // u = x t mod 2^256
                uint64_t c = 0;
                for (int j = 0; i + j < 8; j++) { c += (uint64_t)x[i] * t[j] + u[i + j]; u[i + j] = (uint32_t)c; c >>= 32; }
            }
            for (int i = 0; i < 8; i++) x[i] = u[i];
        }
        Limbs8 o{};
        for (int i = 0; i < 8; i++) o.l[i] = x[i];
        return o;
    }
    static ZK_HD u256 shoup_quotient(const u256& w_lib) {
        constexpr Limbs8 pi = pinv256();
        uint32_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 8; i++) {                                  // t = w_lib p^-1 mod 2^256
            uint64_t c = 0;
            for (int j = 0; i + j < 8; j++) { c += (uint64_t)w_lib.v[i] * pi.l[j] + t[i + j]; t[i + j] = (uint32_t)c; c >>= 32; }
        }
        u256 o;
        uint64_t br = 0;                                               // o = -t mod 2^256
        for (int i = 0; i < 8; i++) { const uint64_t s_ = (uint64_t)t[i] + br; o.v[i] = (uint32_t)(0ull - s_); br = s_ ? 1 : 0; }
        return o;
    }
    static ZK_HD u256 mul_shoup_lazy(const u256& a, const u256& w, const u256& wq) {
        uint64_t acc = 0;
        uint32_t cnt = 0;
        uint32_t q[8];
        u256 r;
#include "field_shoupq_body.inc"
#include "field_shoupr_body.inc"
        return red2p(r);
    }

    // a * R^-1: the Montgomery reduction alone (64 m_i*p_j products; a product with the constant 1 would spend 128) — field_redc_body.inc
    static ZK_HD u256 from_mont(const u256& a) {
        uint64_t acc = 0;
        uint32_t cnt = 0;
        uint32_t m[8];
        u256 r;
#include "field_redc_body.inc"
        r.v[7] = (uint32_t)acc;
        return reduce_once(r);
    }
    static ZK_HD u256 to_mont(const u256& a) { return mul(a, R2()); }

    // a^e, e given as canonical 8x32 limbs (variable-time, top-down square & multiply)
    static ZK_HD u256 pow(const u256& a, const u256& e) {
        u256 acc = one();
        for (int i = 255; i >= 0; i--) {
            acc = sqr(acc);
            if ((e.v[i >> 5] >> (i & 31)) & 1) acc = mul(acc, a);
        }
        return acc;
    }
    static ZK_HD u256 inv(const u256& a) {  // Fermat; 0 -> 0
        u256 e;
#pragma unroll
        for (int i = 0; i < 8; i++) e.v[i] = p(i);
        e.v[0] -= 2;
        return pow(a, e);
    }
};

using Fq = Field<FqParams>;
using Fr = Field<FrParams>;

ZK_HD u256 load_u256(const void* base, size_t idx) {
    const uint4* q = reinterpret_cast<const uint4*>(base) + 2 * idx;
    uint4 lo = q[0], hi = q[1];
    u256 o;
    o.v[0] = lo.x; o.v[1] = lo.y; o.v[2] = lo.z; o.v[3] = lo.w;
    o.v[4] = hi.x; o.v[5] = hi.y; o.v[6] = hi.z; o.v[7] = hi.w;
    return o;
}
ZK_HD void store_u256(void* base, size_t idx, const u256& a) {
    uint4* q = reinterpret_cast<uint4*>(base) + 2 * idx;
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

}  // namespace zk
