// Carry-free-limb field arithmetic for the long product chains (round 3): a value is 9 limbs of 29 bits (261 >= 254 + 7 bits of headroom) and the Montgomery
// radix is 2^261.  A partial product of two limbs is < 2^58, so a whole column of a product — up to 9 a_i*b_j plus 9 m_i*p_j — accumulates in ONE 64-bit
// register with no carry-out: `v_mad_u64_u32` alone, where the 8 x 32-bit form (field.cuh) pairs every mad with a `v_addc_co_u32` on VCC.  162 mads per product
// against 128 mads + 128 addc: 179 G products/s against 142 G on the same box (profiles/r03/run85_microbench_29bit_variants.txt), and the slack of the radix
// (2^261 / p = 151) makes most modular corrections unnecessary: a product of operands a, b (as integers) comes out below a*b / 2^261 + p.
//
// Representation rules (the callers keep them; tests/test_host_logic.py drives every function at the edges):
//   * N-form: limbs 0..7 below 2^29 + 8, limb 8 holds the rest of the integer.  Products return limbs 0..7 strictly below 2^29.
//   * mul / mul2 accept operands whose limb magnitudes A, B satisfy 9*A*B + 9*2^58 + 2^36 < 2^64 per operand pair of a column (e.g. both below 2^30, or one
//     below 2^31 against an N-form one); mul2 sums two such pairs, so at least three of its four operands must be N-form.  sqr: limbs below 2^30.
//   * sub_bias<K, LOG>(a, b) = a - b + K*p with K*p written so that every limb 0..7 is 2^LOG + (a 29-bit digit): limb-wise non-negative whenever b's limbs are
//     at most 2^LOG and b < (K - 1) p; the result has limbs below 2^LOG + 2^29 + max limb of a.
//   * carry(): one PARALLEL round — every limb keeps its low 29 bits and takes its neighbour's overflow — brings limbs below 2^32 to N-form without a
//     dependency chain.
//   * values are congruent to x * 2^261 mod p ("R' form"); from the library's 2^256 Montgomery form that is a multiplication by 32: a shift inside the limb
//     conversion (from_mont32_shl5; the result is below 32 p, fine as ONE operand of a product) or a product with 2^266 mod p (enter, result below 2 p).
#pragma once
#include "field.cuh"

namespace zk {

struct u261 {
    uint32_t l[9];
};

#include "field29_mac.inc"

template <class FP>
struct Field29 {
    using F32 = Field<FP>;
    static constexpr uint32_t M29 = (1u << 29) - 1;
    static constexpr uint32_t INV29 = FP::INV & M29;                  // -p^-1 mod 2^29: the low 29 bits of -p^-1 mod 2^32
    // limb i of p in base 2^29
    static ZK_HD constexpr uint32_t p29(int i) {
        const int bit = i * 29, w = bit / 64, sh = bit % 64;
        uint64_t v = FP::P[w] >> sh;
        if (sh > 35 && w + 1 < 4) v |= FP::P[w + 1] << (64 - sh);
        return (uint32_t)(v & M29);
    }
    // limb i of 2^261 - p (the Shoup product adds q * (2^261 - p) where a Montgomery product adds m * p)
    static ZK_HD constexpr uint32_t np29(int i) { return i == 0 ? (1u << 29) - p29(0) : M29 - p29(i); }
    struct Limbs9 { uint32_t l[9]; };
    // x (four 64-bit words, below 2^256) in base 2^29
    static constexpr Limbs9 digits_of(const uint64_t (&x)[4]) {
        Limbs9 o{};
        for (int i = 0; i < 9; i++) {
            const int bit = i * 29, w = bit / 64, sh = bit % 64;
            uint64_t v = x[w] >> sh;
            if (sh > 35 && w + 1 < 4) v |= x[w + 1] << (64 - sh);
            o.l[i] = (uint32_t)(v & M29);
        }
        return o;
    }
    // K * p with limbs 0..7 written as 2^LOG + digit (see sub_bias): digits of K*p - sum_{i<8} 2^(LOG + 29 i)
    static constexpr Limbs9 bias(uint32_t K, int LOG) {
        Limbs9 o{};
        int64_t t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        uint64_t carry = 0;
        for (int i = 0; i < 9; i++) { const uint64_t v = (uint64_t)p29(i) * K + carry; t[i] = (int64_t)(i < 8 ? (v & M29) : v); carry = i < 8 ? v >> 29 : 0; }
        const int64_t up = (int64_t)1 << (LOG - 29);                  // 2^LOG at limb i is `up` units of limb i + 1
        for (int i = 1; i < 9; i++) t[i] -= up;
        for (int i = 1; i < 8; i++) while (t[i] < 0) { t[i] += (int64_t)1 << 29; t[i + 1] -= 1; }
        for (int i = 0; i < 8; i++) o.l[i] = (uint32_t)(((int64_t)1 << LOG) + t[i]);
        o.l[8] = (uint32_t)t[8];                                      // (positive for every K >= 1: K p >> 2^(LOG + 204))
        return o;
    }

    static ZK_HD u261 zero() { u261 o; for (int i = 0; i < 9; i++) o.l[i] = 0; return o; }
    static ZK_HD u261 constant(const Limbs9& c) { u261 o; for (int i = 0; i < 9; i++) o.l[i] = c.l[i]; return o; }

    // ---- conversions --------------------------------------------------------------------------------------------------------------------------------------------
    // the integer x (8 x 32 bits) as 9 limbs; SHL = 5 multiplies by 32 on the way (261 = 256 + 5: it still fits exactly)
    template <int SHL>
    static ZK_HD u261 from32(const u256& x) {
        u261 o;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int bit = i * 29 - SHL;                             // bit of x where limb i starts (negative: the limb's low bits are zero)
            if (bit < 0) { o.l[i] = (x.v[0] << (-bit)) & M29; continue; }
            const int w = bit / 32, sh = bit % 32;
            uint64_t v = x.v[w];
            if (w + 1 < 8) v |= (uint64_t)x.v[w + 1] << 32;
            o.l[i] = (uint32_t)(v >> sh) & M29;
        }
        return o;
    }
    // N-form or looser (limbs below 2^32) -> the exact integer as 8 x 32 bits; the value must be below 2^256
    static ZK_HD u256 to32(const u261& a) {
        uint32_t n[9];
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { const uint32_t v = a.l[i] + c; n[i] = v & M29; c = v >> 29; }    // (limbs below 2^32 - 8: no wrap)
        n[8] = a.l[8] + c;
        u256 o;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const int bit = w * 32, i = bit / 29, sh = bit % 29;      // word w = bits [32 w, 32 w + 32): limb i from bit sh, then limb i + 1 (and i + 2 when sh > 26)
            uint64_t v = (uint64_t)n[i] >> sh;
            if (i + 1 < 9) v |= (uint64_t)n[i + 1] << (29 - sh);
            if (i + 2 < 9 && 58 - sh < 32) v |= (uint64_t)n[i + 2] << (58 - sh);
            o.v[w] = (uint32_t)v;
        }
        return o;
    }

    // the same for limbs that are already exact (0..7 below 2^29: what a product returns): no carry chain, word w is two or three limbs shifted together
    static ZK_HD u256 to32_exact(const u261& a) {
        u256 o;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const int bit = w * 32, i = bit / 29, sh = bit % 29;
            uint64_t v = (uint64_t)a.l[i] >> sh;
            if (i + 1 < 9) v |= (uint64_t)a.l[i + 1] << (29 - sh);
            if (i + 2 < 9 && 58 - sh < 32) v |= (uint64_t)a.l[i + 2] << (58 - sh);
            o.v[w] = (uint32_t)v;
        }
        return o;
    }

    // ---- products -------------------------------------------------------------------------------------------------------------------------------------------------
    static ZK_HD u261 mul(const u261& a, const u261& b) {
        uint64_t acc = 0;
        uint32_t m[9];
        u261 r;
#include "field29_mul_body.inc"
        return r;
    }
    static ZK_HD u261 sqr(const u261& a) {
        uint64_t acc = 0;
        uint32_t m[9], d[9];
        u261 r;
#pragma unroll
        for (int i = 0; i < 9; i++) d[i] = a.l[i] << 1;
#include "field29_sqr_body.inc"
        return r;
    }
    // (a*b + c*d) * 2^-261 with one reduction; below (a*b + c*d) / 2^261 + p
    static ZK_HD u261 mul2(const u261& a, const u261& b, const u261& c, const u261& d) {
        uint64_t acc = 0;
        uint32_t m[9];
        u261 r;
#include "field29_mul2_body.inc"
        return r;
    }

    // a * w mod p for a CONSTANT w (a twiddle factor): w canonical, wq = shoup_quotient(w) = floor(w 2^261 / p).  q = floor(a wq / 2^261) is taken from the columns 7 .. 16
    // of a * wq only — the columns below add less than 2^-23 of a unit, so q is the true quotient or up to two below — and r = a w - q p is read off the low 261 bits of
    // a w + q (2^261 - p).  For any a below 2^261 with limbs below 3 * 2^30 (a biased difference of N-form values): r has exact limbs below 2^29, is congruent to a w and below 3 p.
    // 143 mads against the 162 + 9 multiplications of the Montgomery product, and NO Montgomery factor: a, r stay in whatever form a is in.  In two halves, so that a caller
    // which fetches wq, then w keeps one of the two constants live at a time (the NTT butterflies: ntt.hip mul_tw).
    struct Quot9 { uint32_t q[9]; };
    static ZK_HD Quot9 shoup_q(const u261& a, const u261& wq) {
        uint64_t acc = 0;
        Quot9 o;
        uint32_t* q = o.q;
#include "field29_shoupq_body.inc"
        return o;
    }
    static ZK_HD u261 shoup_r(const u261& a, const u261& w, const Quot9& qq) {
        uint64_t acc = 0;
        const uint32_t* q = qq.q;
        u261 r;
#include "field29_shoupr_body.inc"
        return r;
    }
    static ZK_HD u261 mul_shoup(const u261& a, const u261& w, const u261& wq) { return shoup_r(a, w, shoup_q(a, wq)); }
    // p^-1 mod 2^261 (Hensel, bit by bit), as limbs
    static constexpr Limbs9 pinv261() {
        uint64_t x[5] = {0, 0, 0, 0, 0}, px[5] = {0, 0, 0, 0, 0};        // px = p * x mod 2^320
        for (int bit = 0; bit < 261; bit++) {
            const uint64_t have = (px[bit / 64] >> (bit % 64)) & 1, want = bit == 0 ? 1 : 0;
            if (have == want) continue;
            x[bit / 64] |= (uint64_t)1 << (bit % 64);
            uint64_t sh[5] = {0, 0, 0, 0, 0};                             // p << bit
            for (int i = 0; i < 4; i++) {
                const int w = i + bit / 64, s = bit % 64;
                if (w < 5) sh[w] |= FP::P[i] << s;
                if (s && w + 1 < 5) sh[w + 1] |= FP::P[i] >> (64 - s);
            }
            uint64_t c = 0;
            for (int i = 0; i < 5; i++) { const uint64_t t = px[i] + sh[i], t2 = t + c; c = (t < px[i] || t2 < t) ? 1 : 0; px[i] = t2; }
        }
        Limbs9 o{};
        for (int i = 0; i < 9; i++) {
            const int bit = i * 29, w = bit / 64, s = bit % 64;
            uint64_t v = x[w] >> s;
            if (s > 35 && w + 1 < 5) v |= x[w + 1] << (64 - s);
            o.l[i] = (uint32_t)(v & M29);
        }
        return o;
    }
    // floor(w 2^261 / p) for a canonical w: w 2^261 = wq p + rho with rho = w 2^261 mod p (a Montgomery product with 2^522 mod p, made canonical), hence wq = -rho p^-1 mod 2^261
    static ZK_HD u261 shoup_quotient(const u256& w) {
        constexpr Limbs9 c522 = pow2_mod_p(522), pinv = pinv261();
        const u261 rho = from32<0>(F32::reduce_once(to32(mul(from32<0>(w), constant(c522)))));     // (the product is below 2 p)
        uint32_t nr[9];                                                // 2^261 - rho (rho = 0 only for w = 0: every limb wraps to 0, wq = 0)
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) { const uint32_t t = 0u - rho.l[i] - borrow; nr[i] = t & M29; borrow = (rho.l[i] | borrow) ? 1 : 0; }
        u261 o;
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) acc += (uint64_t)nr[i] * pinv.l[k - i];
            o.l[k] = (uint32_t)acc & M29;
            acc >>= 29;
        }
        return o;
    }

    // ---- sums of products with ONE reduction ---------------------------------------------------------------------------------------------------------------------------------
    // sum_j a_j b_j 2^-261 for up to SIX pairs of operands with exact limbs (below 2^29: conversions, product outputs, canonical constants): the 17 columns of every a_j b_j are
    // added into 17 64-bit accumulators (81 multiply-adds per pair, no shifts, no carries: a column holds at most 6 * 9 products below 2^58, and the reduction's 9 more plus a
    // carry still fit 64 bits), then one Montgomery reduction (9 + 81 multiplications) serves them all — 96 multiplications per term at six terms against the 162 + 9 of a
    // product of its own.  Result: exact limbs, below sum a_j b_j / 2^261 + p.  Users: the SHPLONK linear combinations and the Horner evaluation (polyeval.hip).
    struct Dot29 { uint64_t c[17]; };
    static ZK_HD void dot_clear(Dot29& d) {
#pragma unroll
        for (int k = 0; k < 17; k++) d.c[k] = 0;
    }
    template <class B9>                                                // B9: u261 or anything with .l[9] (a constant held in scalar registers)
    static ZK_HD void dot_add(Dot29& d, const u261& a, const B9& b) {
#pragma unroll
        for (int i = 0; i < 9; i++)
#pragma unroll
            for (int j = 0; j < 9; j++) d.c[i + j] += (uint64_t)a.l[i] * b.l[j];
    }
    static ZK_HD u261 dot_reduce(Dot29& d) {
        u261 r;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const uint32_t m = ((uint32_t)d.c[k] * INV29) & M29;
#pragma unroll
            for (int j = 0; j < 9; j++) d.c[k + j] += (uint64_t)m * p29(j);
            d.c[k + 1] += d.c[k] >> 29;                                // (the low 29 bits of column k are zero now)
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
            r.l[k - 9] = (uint32_t)d.c[k] & M29;
            if (k < 16) d.c[k + 1] += d.c[k] >> 29;
            else r.l[8] = (uint32_t)(d.c[k] >> 29);
        }
        return r;
    }

    // ---- sums -----------------------------------------------------------------------------------------------------------------------------------------------------
    static ZK_HD u261 add(const u261& a, const u261& b) { u261 o; for (int i = 0; i < 9; i++) o.l[i] = a.l[i] + b.l[i]; return o; }
    static ZK_HD u261 dbl(const u261& a) { u261 o; for (int i = 0; i < 9; i++) o.l[i] = a.l[i] << 1; return o; }
    template <uint32_t K, int LOG>
    static ZK_HD u261 sub_bias(const u261& a, const u261& b) {
        constexpr Limbs9 kp = bias(K, LOG);
        u261 o;
#pragma unroll
        for (int i = 0; i < 9; i++) o.l[i] = a.l[i] + kp.l[i] - b.l[i];
        return o;
    }
    template <uint32_t K, int LOG>
    static ZK_HD u261 neg_bias(const u261& b) {                        // K*p - b
        constexpr Limbs9 kp = bias(K, LOG);
        u261 o;
#pragma unroll
        for (int i = 0; i < 9; i++) o.l[i] = kp.l[i] - b.l[i];
        return o;
    }
    static ZK_HD u261 carry(const u261& a) {                           // one parallel round: limbs below 2^32 -> N-form
        u261 o;
        o.l[0] = a.l[0] & M29;
#pragma unroll
        for (int i = 1; i < 8; i++) o.l[i] = (a.l[i] & M29) + (a.l[i - 1] >> 29);
        o.l[8] = a.l[8] + (a.l[7] >> 29);
        return o;
    }

    // a (N-form, below 32 p) -> the same residue below 3 p, N-form: one quotient estimate from the top limb (q = floor(top * floor(2^285 / p) / 2^53) is the true
    // quotient or up to 2 below it) and one limb-wise a - q p with signed carries.  A third of a product's cost: what ends a chain that grew by additions only.
    static constexpr uint32_t mu285() {                               // floor(2^285 / p): 2^285 = 2^29 * 2^256, long division of (1 << 285) by p in 64-bit words
        // p = P[3..0]; the quotient is below 2^32 (p > 2^253).  Binary long division over 286 bits.
        uint64_t r[5] = {0, 0, 0, 0, 0};
        uint32_t q = 0;
        for (int bit = 285; bit >= 0; bit--) {
            for (int i = 4; i > 0; i--) r[i] = (r[i] << 1) | (r[i - 1] >> 63);          // r = 2 r + bit of the dividend
            r[0] = (r[0] << 1) | (bit == 285 ? 1u : 0u);
            bool ge = r[4] != 0;
            if (!ge) { ge = true; for (int i = 3; i >= 0; i--) if (r[i] != FP::P[i]) { ge = r[i] > FP::P[i]; break; } }
            q = (bit < 32) ? (q << 1) | (ge ? 1u : 0u) : q;             // (quotient bits above 31 are zero)
            if (ge) { uint64_t bo = 0; for (int i = 0; i < 4; i++) { const uint64_t pi = FP::P[i], d = r[i] - pi - bo; bo = (r[i] < pi || (r[i] == pi && bo)) ? 1 : 0; r[i] = d; } r[4] -= bo; }
        }
        return q;
    }
    static ZK_HD u261 reduce_small(const u261& a) {
        constexpr uint32_t MU = mu285();
        const uint32_t q = (uint32_t)(((uint64_t)a.l[8] * MU) >> 53);
        u261 o;
        int64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int64_t t = (int64_t)a.l[i] - (int64_t)((uint64_t)q * p29(i)) + c;
            o.l[i] = (uint32_t)t & M29;
            c = t >> 29;
        }
        o.l[8] = (uint32_t)((int64_t)a.l[8] - (int64_t)((uint64_t)q * p29(8)) + c);
        return o;
    }

    // ---- between the two Montgomery forms ------------------------------------------------------------------------------------------------------------------------
    static constexpr uint64_t pow2_mod_p_words(int e, int w) {          // word w of 2^e mod p (e >= 256), by doubling from R = 2^256 mod p
        uint64_t x[4] = {FP::R[0], FP::R[1], FP::R[2], FP::R[3]};
        for (int s = 256; s < e; s++) {
            uint64_t y[4] = {0, 0, 0, 0}, c = 0;
            for (int i = 0; i < 4; i++) { y[i] = (x[i] << 1) | c; c = x[i] >> 63; }
            bool ge = true;                                          // y >= p ? (p < 2^254 and x < p, so y < 2^255: no word overflow)
            for (int i = 3; i >= 0; i--) { if (y[i] != FP::P[i]) { ge = y[i] > FP::P[i]; break; } }
            if (ge) { uint64_t bo = 0; for (int i = 0; i < 4; i++) { const uint64_t pi = FP::P[i], d = y[i] - pi - bo; bo = (y[i] < pi || (y[i] == pi && bo)) ? 1 : 0; y[i] = d; } }
            for (int i = 0; i < 4; i++) x[i] = y[i];
        }
        return x[w];
    }
    static constexpr Limbs9 pow2_mod_p(int e) {
        const uint64_t x[4] = {pow2_mod_p_words(e, 0), pow2_mod_p_words(e, 1), pow2_mod_p_words(e, 2), pow2_mod_p_words(e, 3)};
        return digits_of(x);
    }
    // v = x * 2^256 (the library's form, canonical) -> x * 2^261 mod p, below 2 p: one product with 2^266 mod p
    static ZK_HD u261 enter(const u256& v) {
        constexpr Limbs9 c = pow2_mod_p(266);
        return mul(from32<0>(v), constant(c));
    }
    // x * 2^261 (any representative a product accepts) -> x * 2^256 mod p canonical: one product with 2^256 mod p, then the exact reduction
    static ZK_HD u256 leave(const u261& a) {
        constexpr Limbs9 c = digits_of(FP::R);
        return F32::reduce_once(to32(mul(a, constant(c))));           // the product is below 2 p < 2^256
    }
    static ZK_HD u261 one() { constexpr Limbs9 c = pow2_mod_p(261); return constant(c); }
};

using Fq29 = Field29<FqParams>;
using Fr29 = Field29<FrParams>;

}  // namespace zk
