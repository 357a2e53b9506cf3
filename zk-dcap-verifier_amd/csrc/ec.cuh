// BN254 G1 (y^2 = x^3 + 3 over Fq) group law for the MSM kernels.
//
// Bases cross the FFI as halo2curves G1Affine {x, y} (Montgomery Fq, identity = (0,0));
// accumulators live in extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2,
// identity = ZZ == 0) because the mixed addition is the cheapest complete-enough formula
// without inversions: 8M + 2S (EFD madd-2008-s; the two products of Y3 share one Montgomery reduction), vs 7M + 4S for Jacobian madd-2007-bl that
// halo2curves uses on the CPU.  Results are canonical after normalisation, so the choice of
// coordinates is invisible to the caller (SURVEY.md App. C.1 "GPU freedom").
#pragma once
#include "field.cuh"
#include "field29.cuh"

namespace zk {

struct Affine {
    u256 x, y;
};
struct XYZZ {
    u256 x, y, zz, zzz;
};

ZK_HD bool affine_is_identity(const Affine& p) { return Fq::is_zero(p.x) && Fq::is_zero(p.y); }
ZK_HD bool xyzz_is_identity(const XYZZ& p) { return Fq::is_zero(p.zz); }
ZK_HD XYZZ xyzz_identity() {
    XYZZ o;
    o.x = Fq::zero(); o.y = Fq::zero(); o.zz = Fq::zero(); o.zzz = Fq::zero();
    return o;
}
ZK_HD XYZZ xyzz_from_affine(const Affine& p) {
    XYZZ o;
    if (affine_is_identity(p)) return xyzz_identity();
    o.x = p.x; o.y = p.y; o.zz = Fq::one(); o.zzz = Fq::one();
    return o;
}

// 2*(x1, y1) for an affine point (mdbl-2008-s-1, a = 0)
ZK_HD XYZZ xyzz_mdbl(const u256& x1, const u256& y1) {
    XYZZ o;
    u256 U = Fq::dbl(y1);
    u256 V = Fq::sqr(U);
    u256 W = Fq::mul(U, V);
    u256 S = Fq::mul(x1, V);
    u256 X2 = Fq::sqr(x1);
    u256 M = Fq::add(Fq::dbl(X2), X2);
    o.x = Fq::sub(Fq::sqr(M), Fq::dbl(S));
    o.y = Fq::mul2_sub(M, Fq::sub(S, o.x), W, y1);           // M*(S - X3) - W*Y1, one reduction
    o.zz = V;
    o.zzz = W;
    return o;
}
// dbl-2008-s-1
ZK_HD XYZZ xyzz_dbl(const XYZZ& p) {
    if (xyzz_is_identity(p)) return p;
    XYZZ o;
    u256 U = Fq::dbl(p.y);
    u256 V = Fq::sqr(U);
    u256 W = Fq::mul(U, V);
    u256 S = Fq::mul(p.x, V);
    u256 X2 = Fq::sqr(p.x);
    u256 M = Fq::add(Fq::dbl(X2), X2);
    o.x = Fq::sub(Fq::sqr(M), Fq::dbl(S));
    o.y = Fq::mul2_sub(M, Fq::sub(S, o.x), W, p.y);
    o.zz = Fq::mul(V, p.zz);
    o.zzz = Fq::mul(W, p.zzz);
    return o;
}
// acc += (x2, y2)    (madd-2008-s; y2 already carries the sign)
ZK_HD void xyzz_madd(XYZZ& acc, const u256& x2, const u256& y2) {
    if (xyzz_is_identity(acc)) {
        acc.x = x2; acc.y = y2; acc.zz = Fq::one(); acc.zzz = Fq::one();
        return;
    }
    u256 U2 = Fq::mul(x2, acc.zz);
    u256 S2 = Fq::mul(y2, acc.zzz);
    u256 P = Fq::sub(U2, acc.x);
    u256 R = Fq::sub(S2, acc.y);
    if (Fq::is_zero(P)) {  // same x: doubling or cancellation (rare; bases repeat or P + (-P))
        if (Fq::is_zero(R)) acc = xyzz_mdbl(x2, y2);
        else acc = xyzz_identity();
        return;
    }
    u256 PP = Fq::sqr(P);
    u256 PPP = Fq::mul(P, PP);
    u256 Q = Fq::mul(acc.x, PP);
    u256 X3 = Fq::sub(Fq::sub(Fq::sqr(R), PPP), Fq::dbl(Q));
    u256 Y3 = Fq::mul2_sub(R, Fq::sub(Q, X3), acc.y, PPP);   // R*(Q - X3) - Y1*PPP, one reduction
    acc.x = X3;
    acc.y = Y3;
    acc.zz = Fq::mul(acc.zz, PP);
    acc.zzz = Fq::mul(acc.zzz, PPP);
}
// The same addition for long chains (the bucket accumulation): the accumulator's coordinates live in [0, 2q) — products skip their final subtraction
// (field.cuh, "one notch tighter") — and (x2, y2) is canonical.  acc.zz is exactly 0 for the identity (it is only ever SET to zero); the caller
// brings the coordinates back to [0, q) once, after the last addition (xyzz_normalize).
ZK_HD void xyzz_madd_lazy(XYZZ& acc, const u256& x2, const u256& y2) {
    if (xyzz_is_identity(acc)) {
        acc.x = x2; acc.y = y2; acc.zz = Fq::one(); acc.zzz = Fq::one();
        return;
    }
    const u256 U2 = Fq::mul_lazy(x2, acc.zz);
    const u256 S2 = Fq::mul_lazy(y2, acc.zzz);
    const u256 P = Fq::sub2(U2, acc.x);
    const u256 R = Fq::sub2(S2, acc.y);
    if (Fq::is_zero_mod(P)) {  // same x: doubling or cancellation (rare; bases repeat or P + (-P))
        if (Fq::is_zero_mod(R)) acc = xyzz_mdbl(x2, y2);
        else acc = xyzz_identity();
        return;
    }
    const u256 PP = Fq::sqr_lazy(P);
    const u256 PPP = Fq::mul_lazy(P, PP);
    const u256 Q = Fq::mul_lazy(acc.x, PP);
    const u256 X3 = Fq::sub2(Fq::sub2(Fq::sqr_lazy(R), PPP), Fq::dbl2(Q));
    const u256 Y3 = Fq::mul2_add_2p(R, Fq::sub2(Q, X3), acc.y, Fq::neg2(PPP));   // R*(Q - X3) - Y1*PPP, one reduction
    acc.x = X3;
    acc.y = Y3;
    acc.zz = Fq::mul_lazy(acc.zz, PP);
    acc.zzz = Fq::mul_lazy(acc.zzz, PPP);
}
ZK_HD void xyzz_madd_signed_lazy(XYZZ& acc, const Affine& p, bool negate) {
    if (affine_is_identity(p)) return;
    const u256 y = negate ? Fq::neg(p.y) : p.y;
    xyzz_madd_lazy(acc, p.x, y);
}
// ---- the same chain on carry-free limbs (field29.cuh): 9 x 29-bit limbs, Montgomery radix 2^261 -------------------------------------------------------------------
// The accumulator of a sub-bucket lives in XYZZ29 — coordinates congruent to x * 2^261, limbs in N-form — from its first point to its last; the table's points are
// the library's canonical 2^256 form and enter a product through the limb conversion (x * 32 = a shift of the limb boundaries, below 32 p: fine as one operand).
// Bounds along the chain, in multiples of p (a product returns below a*b/151 + 1):
//   X1 < 7, Y1 < 2, ZZ1, ZZZ1 < 2;  U2, S2 < 32*2/151 + 1 = 1.43;  P = U2 - X1 + 8p < 9.43;  R = S2 - Y1 + 3p < 4.43;  PP < 1.59;  PPP < 1.10;  Q < 1.08;
//   R^2 < 1.13;  X3 = R^2 - (PPP + 2Q) + 5p < 6.2 (< 7);  D1 = Q - X3 + 8p < 9.1;  D2 = 3p - PPP < 3;  Y3 = (R*D1 + Y1*D2)/2^261 + p < 1.31 (< 2);  ZZ3, ZZZ3 < 1.03.
// Limb bounds: P, R, X3, D1 take one carry round each (they feed a squaring, an 18-term column or the next step's subtraction); D2 goes in as it is (limbs below
// 2^30 + 2^29 against the N-form Y1).  No modular correction anywhere in the step.  P = 0 mod p (the same x: a doubling or a cancellation — repeated bases, P + (-P)) is
// caught by a one-limb filter — P is a multiple of p only if (P mod 2^29) * p^-1 mod 2^29 is that small multiple — and sent through the canonical formulas of the
// 32-bit form; a false alarm (2^-25 per addition) costs time, never correctness, because that path is complete.
struct XYZZ29 {
    u261 x, y, zz, zzz;
    bool ident;
};
ZK_HD XYZZ29 xyzz29_identity() {
    XYZZ29 o;
    o.x = o.y = o.zz = o.zzz = Fq29::zero();
    o.ident = true;
    return o;
}
ZK_HD XYZZ xyzz29_leave(const XYZZ29& a) {                             // -> canonical coordinates of the library's form
    if (a.ident) return xyzz_identity();
    XYZZ o;
    o.x = Fq29::leave(a.x); o.y = Fq29::leave(a.y); o.zz = Fq29::leave(a.zz); o.zzz = Fq29::leave(a.zzz);
    return o;
}
ZK_HD XYZZ29 xyzz29_enter(const XYZZ& a) {
    XYZZ29 o;
    o.ident = xyzz_is_identity(a);
    if (o.ident) return xyzz29_identity();
    o.x = Fq29::enter(a.x); o.y = Fq29::enter(a.y); o.zz = Fq29::enter(a.zz); o.zzz = Fq29::enter(a.zzz);
    return o;
}
// acc += (x2, y2) for an accumulator that is NOT the identity; (x2, y2) canonical, not the identity.  Returns false — acc untouched — when the one-limb filter sees
// P = U2 - X1 as a small multiple of p (the same x: a doubling or a cancellation; 2^-25 false alarms): the caller then continues the chain in the 32-bit form,
// whose formulas are complete (msm_accumulate_kernel), so the hot loop carries no rare-case code at all.
// `late` runs once between the first seven products and the last three, where few values are live: the bucket kernel issues its next point's loads there.
struct Xyzz29Nothing { ZK_HD void operator()() const {} };
template <class Late = Xyzz29Nothing>
ZK_HD bool xyzz29_madd_fast(XYZZ29& acc, const u256& x2, const u256& y2, Late late = Late()) {
    const u261 U2 = Fq29::mul(Fq29::from32<5>(x2), acc.zz);
    const u261 Pw = Fq29::sub_bias<8, 30>(U2, acc.x);
    {   // P = j p for a small j  <=>  the points share their x
        constexpr uint32_t PINV29 = (0u - Fq29::INV29) & Fq29::M29;    // p^-1 mod 2^29
        if ((((Pw.l[0] & Fq29::M29) * PINV29) & Fq29::M29) <= 10u) return false;
    }
    const u261 S2 = Fq29::mul(Fq29::from32<5>(y2), acc.zzz);
    const u261 P = Fq29::carry(Pw);
    const u261 R = Fq29::carry(Fq29::sub_bias<3, 30>(S2, acc.y));
    const u261 PP = Fq29::sqr(P);
    const u261 PPP = Fq29::mul(P, PP);
    const u261 Q = Fq29::mul(acc.x, PP);
    const u261 X3 = Fq29::carry(Fq29::sub_bias<5, 31>(Fq29::sqr(R), Fq29::add(PPP, Fq29::dbl(Q))));
    const u261 D1 = Fq29::carry(Fq29::sub_bias<8, 30>(Q, X3));
    const u261 D2 = Fq29::neg_bias<3, 30>(PPP);
    acc.x = X3;
    late();
    acc.y = Fq29::mul2(R, D1, acc.y, D2);                                // R*(Q - X3) - Y1*PPP, one reduction
    acc.zz = Fq29::mul(acc.zz, PP);
    acc.zzz = Fq29::mul(acc.zzz, PPP);
    return true;
}
// the complete step (host code, tests, the microbenchmark): rare cases through the canonical 32-bit formulas
ZK_HD void xyzz29_madd(XYZZ29& acc, const u256& x2, const u256& y2) {
    if (acc.ident) {
        acc.x = Fq29::enter(x2); acc.y = Fq29::enter(y2); acc.zz = Fq29::one(); acc.zzz = Fq29::one();
        acc.ident = false;
        return;
    }
    if (xyzz29_madd_fast(acc, x2, y2)) return;
    XYZZ c = xyzz29_leave(acc);
    xyzz_madd(c, x2, y2);
    acc = xyzz29_enter(c);
}
ZK_HD void xyzz29_madd_signed(XYZZ29& acc, const Affine& p, bool negate) {
    if (affine_is_identity(p)) return;
    const u256 y = negate ? Fq::neg(p.y) : p.y;
    xyzz29_madd(acc, p.x, y);
}

ZK_HD void xyzz_normalize(XYZZ& acc) {                                // coordinates in [0, 2q) -> [0, q)
    acc.x = Fq::reduce_once(acc.x); acc.y = Fq::reduce_once(acc.y); acc.zz = Fq::reduce_once(acc.zz); acc.zzz = Fq::reduce_once(acc.zzz);
}
ZK_HD void xyzz_madd_signed(XYZZ& acc, const Affine& p, bool negate) {
    if (affine_is_identity(p)) return;
    u256 y = negate ? Fq::neg(p.y) : p.y;
    xyzz_madd(acc, p.x, y);
}
// acc += q   (add-2008-s)
ZK_HD void xyzz_add(XYZZ& acc, const XYZZ& q) {
    if (xyzz_is_identity(q)) return;
    if (xyzz_is_identity(acc)) { acc = q; return; }
    u256 U1 = Fq::mul(acc.x, q.zz);
    u256 U2 = Fq::mul(q.x, acc.zz);
    u256 S1 = Fq::mul(acc.y, q.zzz);
    u256 S2 = Fq::mul(q.y, acc.zzz);
    u256 P = Fq::sub(U2, U1);
    u256 R = Fq::sub(S2, S1);
    if (Fq::is_zero(P)) {
        if (Fq::is_zero(R)) acc = xyzz_dbl(acc);
        else acc = xyzz_identity();
        return;
    }
    u256 PP = Fq::sqr(P);
    u256 PPP = Fq::mul(P, PP);
    u256 Q = Fq::mul(U1, PP);
    u256 X3 = Fq::sub(Fq::sub(Fq::sqr(R), PPP), Fq::dbl(Q));
    u256 Y3 = Fq::mul2_sub(R, Fq::sub(Q, X3), S1, PPP);
    acc.x = X3;
    acc.y = Y3;
    acc.zz = Fq::mul(Fq::mul(acc.zz, q.zz), PP);
    acc.zzz = Fq::mul(Fq::mul(acc.zzz, q.zzz), PPP);
}

// acc += q with both points' coordinates in [0, 2q) (the bucket-reduction chains: merge levels, row / column sums, weight-bit classes); xyzz_normalize at the end
ZK_HD void xyzz_add_lazy(XYZZ& acc, const XYZZ& q) {
    if (xyzz_is_identity(q)) return;
    if (xyzz_is_identity(acc)) { acc = q; return; }
    const u256 U1 = Fq::mul_lazy(acc.x, q.zz);
    const u256 U2 = Fq::mul_lazy(q.x, acc.zz);
    const u256 S1 = Fq::mul_lazy(acc.y, q.zzz);
    const u256 S2 = Fq::mul_lazy(q.y, acc.zzz);
    const u256 P = Fq::sub2(U2, U1);
    const u256 R = Fq::sub2(S2, S1);
    if (Fq::is_zero_mod(P)) {
        if (Fq::is_zero_mod(R)) { xyzz_normalize(acc); acc = xyzz_dbl(acc); }
        else acc = xyzz_identity();
        return;
    }
    const u256 PP = Fq::sqr_lazy(P);
    const u256 PPP = Fq::mul_lazy(P, PP);
    const u256 Q = Fq::mul_lazy(U1, PP);
    const u256 X3 = Fq::sub2(Fq::sub2(Fq::sqr_lazy(R), PPP), Fq::dbl2(Q));
    const u256 Y3 = Fq::mul2_add_2p(R, Fq::sub2(Q, X3), S1, Fq::neg2(PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = Fq::mul_lazy(Fq::mul_lazy(acc.zz, q.zz), PP);
    acc.zzz = Fq::mul_lazy(Fq::mul_lazy(acc.zzz, q.zzz), PPP);
}

ZK_HD Affine load_affine(const void* base, size_t idx) {
    Affine p;
    p.x = load_u256(base, 2 * idx);
    p.y = load_u256(base, 2 * idx + 1);
    return p;
}
ZK_HD void store_affine(void* base, size_t idx, const Affine& p) {
    store_u256(base, 2 * idx, p.x);
    store_u256(base, 2 * idx + 1, p.y);
}
ZK_HD XYZZ load_xyzz(const void* base, size_t idx) {
    XYZZ p;
    p.x = load_u256(base, 4 * idx);
    p.y = load_u256(base, 4 * idx + 1);
    p.zz = load_u256(base, 4 * idx + 2);
    p.zzz = load_u256(base, 4 * idx + 3);
    return p;
}
ZK_HD void store_xyzz(void* base, size_t idx, const XYZZ& p) {
    store_u256(base, 4 * idx, p.x);
    store_u256(base, 4 * idx + 1, p.y);
    store_u256(base, 4 * idx + 2, p.zz);
    store_u256(base, 4 * idx + 3, p.zzz);
}

}  // namespace zk
