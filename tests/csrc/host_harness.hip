// TEST-ONLY: runs the product's __host__ __device__ field / curve routines on the CPU so the
// limb logic can be checked against the oracle without a GPU (tests/test_host_logic.py).
// It is not a CPU fallback: the product library never links this file.
#include "../../zk-dcap-verifier_amd/csrc/ec.cuh"
using namespace zk;
template <class F29>
static void f29_raw(int op, const u261* a, const u261* b, const u261* c, const u261* d, u261* o, size_t n) {
    for (size_t i = 0; i < n; i++) {
        switch (op) {
            case 0: o[i] = F29::mul(a[i], b[i]); break;
            case 1: o[i] = F29::sqr(a[i]); break;
            case 2: o[i] = F29::mul2(a[i], b[i], c[i], d[i]); break;
            case 3: o[i] = F29::carry(a[i]); break;
            case 4: o[i] = F29::template sub_bias<8, 30>(a[i], b[i]); break;
            case 5: o[i] = F29::template sub_bias<3, 30>(a[i], b[i]); break;
            case 6: o[i] = F29::template sub_bias<5, 31>(a[i], b[i]); break;
            case 7: o[i] = F29::template neg_bias<3, 30>(a[i]); break;
            case 8: o[i] = F29::add(a[i], F29::dbl(b[i])); break;
            case 10: o[i] = F29::mul_shoup(a[i], b[i], c[i]); break;                                               // a * w with w's precomputed quotient
            default: o[i] = F29::one(); break;
        }
    }
}
template <class F29>
static void f29_forms(int op, const u256* a, const u256* b, u256* o, u261* o9, size_t n) {
    for (size_t i = 0; i < n; i++) {
        switch (op) {
            case 0: o9[i] = F29::enter(a[i]); o[i] = F29::leave(o9[i]); break;                                  // there and back
            case 1: o9[i] = F29::mul(F29::enter(a[i]), F29::enter(b[i])); o[i] = F29::leave(o9[i]); break;     // a * b in the library's form
            case 2: o9[i] = F29::mul(F29::template from32<5>(a[i]), F29::enter(b[i])); o[i] = F29::leave(o9[i]); break;   // the shifted conversion as one operand
            case 3: o9[i] = F29::template from32<0>(a[i]); o[i] = F29::to32(o9[i]); break;                      // limb conversion alone
            case 5: o9[i] = F29::shoup_quotient(a[i]); o[i] = a[i]; break;                                        // floor(a 2^261 / p)
            default: o9[i] = F29::sqr(F29::enter(a[i])); o[i] = F29::leave(o9[i]); break;
        }
    }
}
extern "C" {
void hh_fr_mul(const u256* a, const u256* b, u256* o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = Fr::mul(a[i], b[i]); }
void hh_fr_add(const u256* a, const u256* b, u256* o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = Fr::add(a[i], b[i]); }
void hh_fr_sub(const u256* a, const u256* b, u256* o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = Fr::sub(a[i], b[i]); }
void hh_fr_neg(const u256* a, u256* o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = Fr::neg(a[i]); }
void hh_fr_inv(const u256* a, u256* o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = Fr::inv(a[i]); }
void hh_fr_from_mont(const u256* a, u256* o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = Fr::from_mont(a[i]); }
void hh_fq_mul(const u256* a, const u256* b, u256* o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = Fq::mul(a[i], b[i]); }
void hh_fq_mul2_sub(const u256* a, const u256* b, const u256* c, const u256* d, u256* o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = Fq::mul2_sub(a[i], b[i], c[i], d[i]); }
void hh_fq_add(const u256* a, const u256* b, u256* o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = Fq::add(a[i], b[i]); }
void hh_fq_sub(const u256* a, const u256* b, u256* o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = Fq::sub(a[i], b[i]); }
// acc (XYZZ) += sign * p for a list of affine points; acc starts at identity
void hh_xyzz_sum(const Affine* pts, const uint8_t* neg, size_t n, XYZZ* out) {
    XYZZ acc = xyzz_identity();
    for (size_t i = 0; i < n; i++) xyzz_madd_signed(acc, pts[i], neg[i] != 0);
    *out = acc;
}
// the redundant-range forms (field.cuh): inputs may be anywhere in the range each function documents
void hh_fq_lazy(int op, const u256* a, const u256* b, const u256* c, const u256* d, u256* o, size_t n) {
    for (size_t i = 0; i < n; i++) {
        switch (op) {
            case 0: o[i] = Fq::mul_lazy(a[i], b[i]); break;
            case 1: o[i] = Fq::sqr_lazy(a[i]); break;
            case 2: o[i] = Fq::sub2(a[i], b[i]); break;
            case 3: o[i] = Fq::dbl2(a[i]); break;
            case 4: o[i] = Fq::neg2(a[i]); break;
            case 5: o[i] = Fq::mul2_add_2p(a[i], b[i], c[i], d[i]); break;
            case 6: o[i] = Fq::red2p(a[i]); break;
            case 7: o[i] = Fq::add_lazy(a[i], b[i]); break;
            case 8: o[i] = Fq::sub_lazy(a[i], b[i]); break;
            case 9: o[i] = Fq::normalize(a[i]); break;
            case 10: o[i] = Fq::zero(); o[i].v[0] = Fq::is_zero_mod(a[i]) ? 1 : 0; break;
            case 14: o[i] = Fq::shoup_quotient(a[i]); break;                // floor(w 2^256 / p) from w's library form
            case 15: o[i] = Fr::shoup_quotient(a[i]); break;
            case 12: o[i] = Fq::mul_shoup_lazy(a[i], b[i], c[i]); break;   // a * w with wq = floor(w 2^256 / p): no Montgomery factor, [0, 2p)
            case 13: o[i] = Fr::mul_shoup_lazy(a[i], b[i], c[i]); break;
            default: o[i] = Fq::mul(a[i], b[i]); break;              // 11: the full product on inputs up to 4p
        }
    }
}
void hh_xyzz_sum_lazy(const Affine* pts, const uint8_t* neg, size_t n, XYZZ* out) {
    XYZZ acc = xyzz_identity();
    for (size_t i = 0; i < n; i++) xyzz_madd_signed_lazy(acc, pts[i], neg[i] != 0);
    xyzz_normalize(acc);
    *out = acc;
}
void hh_xyzz_add_lazy(const XYZZ* a, const XYZZ* b, XYZZ* out) { XYZZ t = *a; xyzz_add_lazy(t, *b); xyzz_add_lazy(t, *b); xyzz_normalize(t); *out = t; }   // a + b + b: the second addition meets lazy coordinates
void hh_xyzz_add(const XYZZ* a, const XYZZ* b, XYZZ* out) { XYZZ t = *a; xyzz_add(t, *b); *out = t; }
void hh_xyzz_dbl(const XYZZ* a, XYZZ* out) { *out = xyzz_dbl(*a); }

// ---- field29.cuh: the carry-free 29-bit-limb arithmetic, on raw limbs (the Python model of tests/test_field29.py recomputes every column sum exactly and checks it fits
// 64 bits, so an overflow here would show as a difference) and through the two Montgomery forms
void hh_f29_raw(int field, int op, const u261* a, const u261* b, const u261* c, const u261* d, u261* o, size_t n) {
    if (field == 0) f29_raw<Fq29>(op, a, b, c, d, o, n); else f29_raw<Fr29>(op, a, b, c, d, o, n);
}
void hh_f29_forms(int field, int op, const u256* a, const u256* b, u256* o, u261* o9, size_t n) {
    if (field == 0) f29_forms<Fq29>(op, a, b, o, o9, n); else f29_forms<Fr29>(op, a, b, o, o9, n);
}
// the bucket chain on 29-bit limbs, complete form (rare cases through the canonical formulas), result in canonical coordinates
void hh_xyzz29_sum(const Affine* pts, const uint8_t* neg, size_t n, XYZZ* out, uint32_t* n_rare) {
    XYZZ29 acc = xyzz29_identity();
    uint32_t rare = 0;
    for (size_t i = 0; i < n; i++) {
        if (affine_is_identity(pts[i])) continue;
        const u256 y = neg[i] ? Fq::neg(pts[i].y) : pts[i].y;
        if (acc.ident || !xyzz29_madd_fast(acc, pts[i].x, y)) { rare++; xyzz29_madd(acc, pts[i].x, y); }
    }
    *out = xyzz29_leave(acc);
    *n_rare = rare;
}
}
