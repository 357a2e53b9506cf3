/*
 * TEST INFRASTRUCTURE ONLY — CPU oracle for the BN254 halo2 prover hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link
 * or call this file.  The product (zk-dcap-verifier_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned" by the reference.  /root/reference holds no
 * source for this path (it only calls create_proof, keygen_vk, keygen_pk, gen_srs:
 * circuits/src/sgx_dcap_verifier.rs:799-822, crates/p256-ecdsa/src/base.rs:134,
 * 145,193-212) and none of its tests pins an MSM/NTT/h(X) value.  The algorithms
 * below are a plain-C RESTATEMENT of the published algorithms of the pinned,
 * un-vendored dependencies:
 *   halo2_proofs 0.2.0  git zkwebauthn/halo2 @ c254c75   (Cargo.lock:1314-1327)
 *       src/arithmetic.rs      best_multiexp / multiexp_serial / best_fft /
 *                              recursive_butterfly_arithmetic
 *       src/poly/domain.rs     EvaluationDomain::{new, lagrange_to_coeff,
 *                              coeff_to_extended, extended_to_coeff,
 *                              divide_by_vanishing_poly}
 *       src/plonk/evaluation.rs  GraphEvaluator / Evaluator::evaluate_h
 *   halo2curves 0.3.1   git zkwebauthn/halo2curves @ bdb2e66 (Cargo.lock:1329-1344)
 *       src/bn256/{fr.rs,fq.rs,curve.rs}: 4 x u64 Montgomery limbs, R = 2^256,
 *       Jacobian G1 {x,y,z}, affine identity = (0,0).
 * It is pinned against (a) the pure-integer definitions in oracle/pyref.py,
 * (b) the KATs of SURVEY.md App. A, (c) the reference's only shipped proof
 * bytes (bin/assets/proof.bin) for curve/field membership.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>
#include "bn254_consts.h"

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;            /* field element, Montgomery form */
typedef struct { fe x, y; } g1a;                 /* affine, (0,0) = identity       */
typedef struct { fe x, y, z; } g1j;              /* Jacobian, z = 0 identity       */

/* ------------------------------------------------------------------ fields */
typedef struct { fe p, r, r2; uint64_t inv; } fparams;
static const fparams FQ = { {BN254_FQ_MODULUS}, {BN254_FQ_R}, {BN254_FQ_R2}, BN254_FQ_INV64 };
static const fparams FR = { {BN254_FR_MODULUS}, {BN254_FR_R}, {BN254_FR_R2}, BN254_FR_INV64 };

static inline int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) { return memcmp(a, b, 32) == 0; }
static inline int geq(const fe *a, const fe *p) {
    for (int i = 3; i >= 0; i--) { if (a->l[i] > p->l[i]) return 1; if (a->l[i] < p->l[i]) return 0; }
    return 1;
}
static inline void sub_nored(fe *r, const fe *a, const fe *b) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) { u128 t = (u128)a->l[i] - b->l[i] - br; r->l[i] = (uint64_t)t; br = (t >> 64) & 1; }
}
static inline void f_add(const fparams *F, fe *r, const fe *a, const fe *b) {
    u128 c = 0; fe t;
    for (int i = 0; i < 4; i++) { c += (u128)a->l[i] + b->l[i]; t.l[i] = (uint64_t)c; c >>= 64; }
    if (geq(&t, &F->p)) sub_nored(&t, &t, &F->p);   /* moduli < 2^254: no carry out */
    *r = t;
}
static inline void f_sub(const fparams *F, fe *r, const fe *a, const fe *b) {
    u128 br = 0; fe t;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a->l[i] - b->l[i] - br; t.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)t.l[i] + F->p.l[i]; t.l[i] = (uint64_t)c; c >>= 64; } }
    *r = t;
}
static inline void f_neg(const fparams *F, fe *r, const fe *a) {
    if (fe_is_zero(a)) { *r = *a; return; }
    sub_nored(r, &F->p, a);
}
/* Montgomery product a*b*2^-256 mod p (coarsely-integrated operand scanning) */
static inline void f_mul(const fparams *F, fe *r, const fe *a, const fe *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F->inv;
        c = ((u128)m * F->p.l[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * F->p.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fe o = { { t[0], t[1], t[2], t[3] } };
    if (t[4] || geq(&o, &F->p)) sub_nored(&o, &o, &F->p);
    *r = o;
}
static inline void f_sqr(const fparams *F, fe *r, const fe *a) { f_mul(F, r, a, a); }
static inline void f_dbl(const fparams *F, fe *r, const fe *a) { f_add(F, r, a, a); }
static void f_pow(const fparams *F, fe *r, const fe *a, const fe *e) {
    fe acc = F->r;
    for (int i = 255; i >= 0; i--) {
        f_sqr(F, &acc, &acc);
        if ((e->l[i >> 6] >> (i & 63)) & 1) f_mul(F, &acc, &acc, a);
    }
    *r = acc;
}
static void f_inv(const fparams *F, fe *r, const fe *a) {   /* 0 -> 0 */
    fe e = F->p; e.l[0] -= 2;                                  /* low limb of both moduli >= 2 */
    f_pow(F, r, a, &e);
}
static inline void f_to_mont(const fparams *F, fe *r, const fe *a) { f_mul(F, r, a, &F->r2); }
static inline void f_from_mont(const fparams *F, fe *r, const fe *a) { fe one = { {1, 0, 0, 0} }; f_mul(F, r, a, &one); }

/* -------------------------------------------------- exported field helpers */
#define VEC_OP(name, F, expr)                                                    \
    void name(const fe *a, const fe *b, fe *out, size_t n) {                     \
        for (size_t i = 0; i < n; i++) { expr; }                                 \
    }
VEC_OP(orc_fr_mul_vec, FR, f_mul(&FR, &out[i], &a[i], &b[i]))
VEC_OP(orc_fr_add_vec, FR, f_add(&FR, &out[i], &a[i], &b[i]))
VEC_OP(orc_fr_sub_vec, FR, f_sub(&FR, &out[i], &a[i], &b[i]))
VEC_OP(orc_fq_mul_vec, FQ, f_mul(&FQ, &out[i], &a[i], &b[i]))
VEC_OP(orc_fq_add_vec, FQ, f_add(&FQ, &out[i], &a[i], &b[i]))
VEC_OP(orc_fq_sub_vec, FQ, f_sub(&FQ, &out[i], &a[i], &b[i]))
void orc_fr_to_mont(const fe *a, fe *out, size_t n) { for (size_t i = 0; i < n; i++) f_to_mont(&FR, &out[i], &a[i]); }
void orc_fr_from_mont(const fe *a, fe *out, size_t n) { for (size_t i = 0; i < n; i++) f_from_mont(&FR, &out[i], &a[i]); }
void orc_fq_to_mont(const fe *a, fe *out, size_t n) { for (size_t i = 0; i < n; i++) f_to_mont(&FQ, &out[i], &a[i]); }
void orc_fq_from_mont(const fe *a, fe *out, size_t n) { for (size_t i = 0; i < n; i++) f_from_mont(&FQ, &out[i], &a[i]); }
void orc_fr_inv(const fe *a, fe *out, size_t n) { for (size_t i = 0; i < n; i++) f_inv(&FR, &out[i], &a[i]); }
void orc_fq_inv(const fe *a, fe *out, size_t n) { for (size_t i = 0; i < n; i++) f_inv(&FQ, &out[i], &a[i]); }
void orc_fr_pow(const fe *a, const fe *e_canonical, fe *out) { f_pow(&FR, out, a, e_canonical); }

/* ---------------------------------------------------------------- G1 curve */
/* halo2curves new_curve_impl! formulas (Jacobian, a = 0): dbl-2009-l,
 * add-2007-bl, madd-2007-bl.  Any correct formula gives the same normalised
 * point; these are the ones the pinned crate uses. */
static inline int j_is_id(const g1j *p) { return fe_is_zero(&p->z); }
static inline int a_is_id(const g1a *p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static void j_set_id(g1j *p) { memset(p, 0, sizeof *p); }
static void j_from_affine(g1j *r, const g1a *p) {
    if (a_is_id(p)) { j_set_id(r); return; }
    r->x = p->x; r->y = p->y; r->z = FQ.r;
}
static void j_double(g1j *r, const g1j *p) {
    if (j_is_id(p)) { j_set_id(r); return; }
    fe a, b, c, d, e, f, t, x3, y3, z3;
    f_sqr(&FQ, &a, &p->x); f_sqr(&FQ, &b, &p->y); f_sqr(&FQ, &c, &b);
    f_add(&FQ, &d, &p->x, &b); f_sqr(&FQ, &d, &d); f_sub(&FQ, &d, &d, &a); f_sub(&FQ, &d, &d, &c); f_dbl(&FQ, &d, &d);
    f_dbl(&FQ, &e, &a); f_add(&FQ, &e, &e, &a);
    f_sqr(&FQ, &f, &e);
    f_mul(&FQ, &z3, &p->z, &p->y); f_dbl(&FQ, &z3, &z3);
    f_dbl(&FQ, &t, &d); f_sub(&FQ, &x3, &f, &t);
    f_dbl(&FQ, &c, &c); f_dbl(&FQ, &c, &c); f_dbl(&FQ, &c, &c);
    f_sub(&FQ, &t, &d, &x3); f_mul(&FQ, &y3, &e, &t); f_sub(&FQ, &y3, &y3, &c);
    r->x = x3; r->y = y3; r->z = z3;
}
static void j_add(g1j *r, const g1j *p, const g1j *q) {
    if (j_is_id(p)) { *r = *q; return; }
    if (j_is_id(q)) { *r = *p; return; }
    fe z1z1, z2z2, u1, u2, s1, s2;
    f_sqr(&FQ, &z1z1, &p->z); f_sqr(&FQ, &z2z2, &q->z);
    f_mul(&FQ, &u1, &p->x, &z2z2); f_mul(&FQ, &u2, &q->x, &z1z1);
    f_mul(&FQ, &s1, &p->y, &z2z2); f_mul(&FQ, &s1, &s1, &q->z);
    f_mul(&FQ, &s2, &q->y, &z1z1); f_mul(&FQ, &s2, &s2, &p->z);
    if (fe_eq(&u1, &u2)) {
        if (fe_eq(&s1, &s2)) { j_double(r, p); } else { j_set_id(r); }
        return;
    }
    fe h, i, j, rr, v, x3, y3, z3, t;
    f_sub(&FQ, &h, &u2, &u1);
    f_dbl(&FQ, &i, &h); f_sqr(&FQ, &i, &i);
    f_mul(&FQ, &j, &h, &i);
    f_sub(&FQ, &rr, &s2, &s1); f_dbl(&FQ, &rr, &rr);
    f_mul(&FQ, &v, &u1, &i);
    f_sqr(&FQ, &x3, &rr); f_sub(&FQ, &x3, &x3, &j); f_sub(&FQ, &x3, &x3, &v); f_sub(&FQ, &x3, &x3, &v);
    f_mul(&FQ, &t, &s1, &j); f_dbl(&FQ, &t, &t);
    f_sub(&FQ, &y3, &v, &x3); f_mul(&FQ, &y3, &y3, &rr); f_sub(&FQ, &y3, &y3, &t);
    f_add(&FQ, &z3, &p->z, &q->z); f_sqr(&FQ, &z3, &z3); f_sub(&FQ, &z3, &z3, &z1z1); f_sub(&FQ, &z3, &z3, &z2z2);
    f_mul(&FQ, &z3, &z3, &h);
    r->x = x3; r->y = y3; r->z = z3;
}
static void j_add_affine(g1j *r, const g1j *p, const g1a *q) {
    if (a_is_id(q)) { *r = *p; return; }
    if (j_is_id(p)) { j_from_affine(r, q); return; }
    fe z1z1, u2, s2;
    f_sqr(&FQ, &z1z1, &p->z);
    f_mul(&FQ, &u2, &q->x, &z1z1);
    f_mul(&FQ, &s2, &q->y, &z1z1); f_mul(&FQ, &s2, &s2, &p->z);
    if (fe_eq(&p->x, &u2)) {
        if (fe_eq(&p->y, &s2)) { j_double(r, p); } else { j_set_id(r); }
        return;
    }
    fe h, hh, i, j, rr, v, x3, y3, z3, t;
    f_sub(&FQ, &h, &u2, &p->x);
    f_sqr(&FQ, &hh, &h);
    f_dbl(&FQ, &i, &hh); f_dbl(&FQ, &i, &i);
    f_mul(&FQ, &j, &h, &i);
    f_sub(&FQ, &rr, &s2, &p->y); f_dbl(&FQ, &rr, &rr);
    f_mul(&FQ, &v, &p->x, &i);
    f_sqr(&FQ, &x3, &rr); f_sub(&FQ, &x3, &x3, &j); f_sub(&FQ, &x3, &x3, &v); f_sub(&FQ, &x3, &x3, &v);
    f_mul(&FQ, &t, &p->y, &j); f_dbl(&FQ, &t, &t);
    f_sub(&FQ, &y3, &v, &x3); f_mul(&FQ, &y3, &y3, &rr); f_sub(&FQ, &y3, &y3, &t);
    f_add(&FQ, &z3, &p->z, &h); f_sqr(&FQ, &z3, &z3); f_sub(&FQ, &z3, &z3, &z1z1); f_sub(&FQ, &z3, &z3, &hh);
    r->x = x3; r->y = y3; r->z = z3;
}
static void j_to_affine(g1a *r, const g1j *p) {
    if (j_is_id(p)) { memset(r, 0, sizeof *r); return; }
    fe zi, zi2, zi3;
    f_inv(&FQ, &zi, &p->z); f_sqr(&FQ, &zi2, &zi); f_mul(&FQ, &zi3, &zi2, &zi);
    f_mul(&FQ, &r->x, &p->x, &zi2); f_mul(&FQ, &r->y, &p->y, &zi3);
}
static void j_mul(g1j *r, const g1a *p, const fe *k_canonical) {
    g1j acc; j_set_id(&acc);
    for (int i = 255; i >= 0; i--) {
        j_double(&acc, &acc);
        if ((k_canonical->l[i >> 6] >> (i & 63)) & 1) j_add_affine(&acc, &acc, p);
    }
    *r = acc;
}

void orc_g1_to_affine(const g1j *p, g1a *out, size_t n) { for (size_t i = 0; i < n; i++) j_to_affine(&out[i], &p[i]); }
void orc_g1_add(const g1j *a, const g1j *b, g1j *out) { j_add(out, a, b); }
void orc_g1_add_affine(const g1j *a, const g1a *b, g1j *out) { j_add_affine(out, a, b); }
void orc_g1_double(const g1j *a, g1j *out) { j_double(out, a); }
/* scalar in Montgomery form (as Fr lives in memory) */
void orc_g1_mul(const g1a *p, const fe *k_mont, g1j *out) { fe k; f_from_mont(&FR, &k, k_mont); j_mul(out, p, &k); }
int orc_g1_is_on_curve(const g1a *p) {
    if (a_is_id(p)) return 1;
    fe y2, x3, b = { BN254_FQ_THREE_M };
    f_sqr(&FQ, &y2, &p->y); f_sqr(&FQ, &x3, &p->x); f_mul(&FQ, &x3, &x3, &p->x); f_add(&FQ, &x3, &x3, &b);
    return fe_eq(&y2, &x3);
}
void orc_g1_generator(g1a *out) { out->x = FQ.r; fe two = { BN254_FQ_TWO_M }; out->y = two; }

/* bases P_i = [a0 + i*delta] G, i < n  (closed-form MSM check, SURVEY 8d cfg 3) */
typedef struct { g1a *out; size_t lo, hi; fe a0, delta; } genjob;
static void fe_from_u64(fe *r, uint64_t v) { fe t = { {v, 0, 0, 0} }; f_to_mont(&FR, r, &t); }
static void *gen_worker(void *arg) {
    genjob *jb = (genjob *)arg;
    if (jb->lo >= jb->hi) return 0;
    g1a g; orc_g1_generator(&g);
    fe idx, k; fe_from_u64(&idx, jb->lo); f_mul(&FR, &k, &idx, &jb->delta); f_add(&FR, &k, &k, &jb->a0);
    g1j cur, q; orc_g1_mul(&g, &k, &cur); orc_g1_mul(&g, &jb->delta, &q);
    g1a qa; j_to_affine(&qa, &q);
    const size_t B = 1024;
    g1j *buf = malloc(B * sizeof(g1j)); fe *pre = malloc(B * sizeof(fe));
    for (size_t s = jb->lo; s < jb->hi; s += B) {
        size_t m = jb->hi - s < B ? jb->hi - s : B;
        for (size_t i = 0; i < m; i++) { buf[i] = cur; j_add_affine(&cur, &cur, &qa); }
        /* batch normalise (Montgomery's trick) */
        fe acc = FQ.r;
        for (size_t i = 0; i < m; i++) { pre[i] = acc; if (!j_is_id(&buf[i])) f_mul(&FQ, &acc, &acc, &buf[i].z); }
        fe inv; f_inv(&FQ, &inv, &acc);
        for (size_t i = m; i-- > 0;) {
            if (j_is_id(&buf[i])) { memset(&jb->out[s + i], 0, sizeof(g1a)); continue; }
            fe zi, zi2, zi3; f_mul(&FQ, &zi, &inv, &pre[i]); f_mul(&FQ, &inv, &inv, &buf[i].z);
            f_sqr(&FQ, &zi2, &zi); f_mul(&FQ, &zi3, &zi2, &zi);
            f_mul(&FQ, &jb->out[s + i].x, &buf[i].x, &zi2); f_mul(&FQ, &jb->out[s + i].y, &buf[i].y, &zi3);
        }
    }
    free(buf); free(pre);
    return 0;
}
void orc_gen_bases_arith(const fe *a0_mont, const fe *delta_mont, size_t n, int threads, g1a *out) {
    if (threads < 1) threads = 1;
    pthread_t th[64]; genjob jb[64]; if (threads > 64) threads = 64;
    size_t per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        size_t lo = per * t, hi = lo + per; if (lo > n) lo = n; if (hi > n) hi = n;
        jb[t] = (genjob){ out, lo, hi, *a0_mont, *delta_mont };
        pthread_create(&th[t], 0, gen_worker, &jb[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], 0);
}

/* ------------------------------------------------- best_multiexp (App. C.1) */
/* get_at: c-bit unsigned digit #segment of the canonical little-endian repr. */
static inline size_t get_at(size_t segment, size_t c, const uint8_t bytes[32]) {
    size_t skip_bits = segment * c, skip_bytes = skip_bits / 8;
    if (skip_bytes >= 32) return 0;
    uint8_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < 8 && skip_bytes + i < 32; i++) v[i] = bytes[skip_bytes + i];
    uint64_t tmp; memcpy(&tmp, v, 8);
    tmp >>= skip_bits - skip_bytes * 8;
    tmp %= (1ULL << c);
    return (size_t)tmp;
}
/* lazily upgraded bucket: 0 none, 1 affine, 2 projective */
typedef struct { int kind; g1a a; g1j j; } bucket_t;

static void multiexp_serial(const fe *coeffs, const g1a *bases, size_t n, g1j *acc) {
    fe *repr = malloc(n * sizeof(fe));
    for (size_t i = 0; i < n; i++) f_from_mont(&FR, &repr[i], &coeffs[i]);   /* to_repr() */
    size_t c;
    if (n < 4) c = 1; else if (n < 32) c = 3; else c = (size_t)ceil(log((double)(uint32_t)n));
    size_t segments = 256 / c + 1, nb = ((size_t)1 << c) - 1;
    bucket_t *buckets = malloc(nb * sizeof(bucket_t));
    for (size_t seg = segments; seg-- > 0;) {
        for (size_t d = 0; d < c; d++) j_double(acc, acc);
        for (size_t b = 0; b < nb; b++) buckets[b].kind = 0;
        for (size_t i = 0; i < n; i++) {
            size_t d = get_at(seg, c, (const uint8_t *)&repr[i]);
            if (!d) continue;
            bucket_t *bk = &buckets[d - 1];
            if (bk->kind == 0) { bk->a = bases[i]; bk->kind = 1; }
            else if (bk->kind == 1) { g1j t; j_from_affine(&t, &bk->a); j_add_affine(&bk->j, &t, &bases[i]); bk->kind = 2; }
            else j_add_affine(&bk->j, &bk->j, &bases[i]);
        }
        g1j running; j_set_id(&running);          /* summation by parts */
        for (size_t b = nb; b-- > 0;) {
            if (buckets[b].kind == 1) j_add_affine(&running, &running, &buckets[b].a);
            else if (buckets[b].kind == 2) j_add(&running, &running, &buckets[b].j);
            j_add(acc, acc, &running);
        }
    }
    free(buckets); free(repr);
}
typedef struct { const fe *c; const g1a *b; size_t n; g1j acc; } mejob;
static void *me_worker(void *arg) { mejob *j = (mejob *)arg; j_set_id(&j->acc); multiexp_serial(j->c, j->b, j->n, &j->acc); return 0; }

/* out: Jacobian (x,y,z) exactly as the chunked algorithm leaves it */
void orc_best_multiexp(const fe *coeffs, const g1a *bases, size_t n, int num_threads, g1j *out) {
    if (num_threads < 1) num_threads = 1;
    if (n > (size_t)num_threads) {
        size_t chunk = n / num_threads, nchunks = (n + chunk - 1) / chunk;
        mejob *jobs = malloc(nchunks * sizeof(mejob)); pthread_t *th = malloc(nchunks * sizeof(pthread_t));
        for (size_t k = 0; k < nchunks; k++) {
            size_t lo = k * chunk, len = n - lo < chunk ? n - lo : chunk;
            jobs[k].c = coeffs + lo; jobs[k].b = bases + lo; jobs[k].n = len;
            pthread_create(&th[k], 0, me_worker, &jobs[k]);
        }
        g1j acc; j_set_id(&acc);
        for (size_t k = 0; k < nchunks; k++) { pthread_join(th[k], 0); j_add(&acc, &acc, &jobs[k].acc); }
        *out = acc; free(jobs); free(th);
    } else {
        g1j acc; j_set_id(&acc); multiexp_serial(coeffs, bases, n, &acc); *out = acc;
    }
}

/* ------------------------------------------------------ best_fft (App. C.2) */
static size_t bitreverse(size_t n, size_t l) { size_t r = 0; for (size_t i = 0; i < l; i++) { r = (r << 1) | (n & 1); n >>= 1; } return r; }

static inline void butterfly_pair(fe *a, fe *b, const fe *tw) {
    fe t; if (tw) f_mul(&FR, &t, b, tw); else t = *b;
    f_sub(&FR, b, a, &t); f_add(&FR, a, a, &t);
}
typedef struct { fe *a; size_t n, twiddle_chunk; const fe *tw; int depth; } fftjob;
static void recursive_butterfly(fe *a, size_t n, size_t twiddle_chunk, const fe *tw, int depth);
static void *fft_worker(void *arg) { fftjob *j = (fftjob *)arg; recursive_butterfly(j->a, j->n, j->twiddle_chunk, j->tw, j->depth); return 0; }
static void recursive_butterfly(fe *a, size_t n, size_t twiddle_chunk, const fe *tw, int depth) {
    if (n == 2) { butterfly_pair(&a[0], &a[1], 0); return; }
    fe *left = a, *right = a + n / 2;
    if (depth > 0) {                      /* rayon::join */
        pthread_t th; fftjob jb = { right, n / 2, twiddle_chunk * 2, tw, depth - 1 };
        pthread_create(&th, 0, fft_worker, &jb);
        recursive_butterfly(left, n / 2, twiddle_chunk * 2, tw, depth - 1);
        pthread_join(th, 0);
    } else {
        recursive_butterfly(left, n / 2, twiddle_chunk * 2, tw, 0);
        recursive_butterfly(right, n / 2, twiddle_chunk * 2, tw, 0);
    }
    butterfly_pair(&left[0], &right[0], 0);           /* twiddle factor one */
    for (size_t i = 1; i < n / 2; i++) butterfly_pair(&left[i], &right[i], &tw[i * twiddle_chunk]);
}
void orc_best_fft(fe *a, const fe *omega_mont, uint32_t log_n, int num_threads) {
    size_t n = (size_t)1 << log_n;
    if (n == 1) return;
    int log_threads = 0; while ((2 << log_threads) <= num_threads) log_threads++;
    for (size_t k = 0; k < n; k++) { size_t rk = bitreverse(k, log_n); if (k < rk) { fe t = a[rk]; a[rk] = a[k]; a[k] = t; } }
    fe *tw = malloc((n / 2) * sizeof(fe)); fe w = FR.r;
    for (size_t i = 0; i < n / 2; i++) { tw[i] = w; f_mul(&FR, &w, &w, omega_mont); }
    if ((int)log_n <= log_threads) {
        size_t chunk = 2, twiddle_chunk = n / 2;
        for (uint32_t s = 0; s < log_n; s++) {
            for (size_t base = 0; base < n; base += chunk) {
                fe *left = a + base, *right = a + base + chunk / 2;
                butterfly_pair(&left[0], &right[0], 0);
                for (size_t i = 1; i < chunk / 2; i++) butterfly_pair(&left[i], &right[i], &tw[i * twiddle_chunk]);
            }
            chunk *= 2; twiddle_chunk /= 2;
        }
    } else {
        recursive_butterfly(a, n, 1, tw, log_threads);
    }
    free(tw);
}

/* ---------------------------------------------- EvaluationDomain (App. C.3) */
typedef struct {
    uint32_t k, extended_k, quotient_poly_degree, n_t;
    fe omega, omega_inv, extended_omega, extended_omega_inv;
    fe ifft_divisor, extended_ifft_divisor;
    fe t_evaluations[64];
} orc_domain;

static void omega_for(fe *w, uint32_t k) {       /* ROOT_OF_UNITY^(2^(S-k)) */
    fe r = { BN254_FR_ROOT_OF_UNITY_M };
    for (uint32_t i = k; i < BN254_FR_S; i++) f_sqr(&FR, &r, &r);
    *w = r;
}
int orc_domain_new(orc_domain *d, uint32_t j, uint32_t k) {
    memset(d, 0, sizeof *d);
    d->k = k; d->quotient_poly_degree = j - 1;
    uint32_t ek = k; while (((uint64_t)1 << ek) < ((uint64_t)1 << k) * (uint64_t)(j - 1)) ek++;
    if (ek > BN254_FR_S || ek - k > 6) return -1;
    d->extended_k = ek;
    omega_for(&d->extended_omega, ek);
    d->omega = d->extended_omega; for (uint32_t i = k; i < ek; i++) f_sqr(&FR, &d->omega, &d->omega);
    f_inv(&FR, &d->omega_inv, &d->omega); f_inv(&FR, &d->extended_omega_inv, &d->extended_omega);
    fe two_inv = { BN254_FR_TWO_INV_M };
    d->ifft_divisor = FR.r; for (uint32_t i = 0; i < k; i++) f_mul(&FR, &d->ifft_divisor, &d->ifft_divisor, &two_inv);
    d->extended_ifft_divisor = FR.r; for (uint32_t i = 0; i < ek; i++) f_mul(&FR, &d->extended_ifft_divisor, &d->extended_ifft_divisor, &two_inv);
    /* t_evaluations[i] = (zeta^n * (ext_omega^n)^i - 1)^-1 */
    fe zn = { BN254_FR_ZETA_M }, won = d->extended_omega;
    for (uint32_t i = 0; i < k; i++) { f_sqr(&FR, &zn, &zn); f_sqr(&FR, &won, &won); }
    d->n_t = 1u << (ek - k);
    fe cur = zn;
    for (uint32_t i = 0; i < d->n_t; i++) {
        fe t; f_sub(&FR, &t, &cur, &FR.r); f_inv(&FR, &d->t_evaluations[i], &t);
        f_mul(&FR, &cur, &cur, &won);
    }
    return 0;
}
void orc_lagrange_to_coeff(const orc_domain *d, fe *a, int threads) {
    size_t n = (size_t)1 << d->k;
    orc_best_fft(a, &d->omega_inv, d->k, threads);
    for (size_t i = 0; i < n; i++) f_mul(&FR, &a[i], &a[i], &d->ifft_divisor);
}
void orc_coeff_to_lagrange(const orc_domain *d, fe *a, int threads) { orc_best_fft(a, &d->omega, d->k, threads); }
/* in: n coeffs; out: 2^extended_k evaluations on the ZETA-coset */
void orc_coeff_to_extended(const orc_domain *d, const fe *coeff, fe *out, int threads) {
    size_t n = (size_t)1 << d->k, en = (size_t)1 << d->extended_k;
    fe z[3] = { FR.r, { BN254_FR_ZETA_M }, { BN254_FR_ZETA2_M } };
    for (size_t i = 0; i < n; i++) { if (i % 3) f_mul(&FR, &out[i], &coeff[i], &z[i % 3]); else out[i] = coeff[i]; }
    memset(out + n, 0, (en - n) * sizeof(fe));
    orc_best_fft(out, &d->extended_omega, d->extended_k, threads);
}
/* in place on 2^extended_k values; first n*(j-1) entries are the result */
void orc_extended_to_coeff(const orc_domain *d, fe *a, int threads) {
    size_t en = (size_t)1 << d->extended_k;
    orc_best_fft(a, &d->extended_omega_inv, d->extended_k, threads);
    fe zi[3] = { FR.r, { BN254_FR_ZETA2_M }, { BN254_FR_ZETA_M } };
    for (size_t i = 0; i < en; i++) { f_mul(&FR, &a[i], &a[i], &d->extended_ifft_divisor); if (i % 3) f_mul(&FR, &a[i], &a[i], &zi[i % 3]); }
}
void orc_divide_by_vanishing_poly(const orc_domain *d, fe *a) {
    size_t en = (size_t)1 << d->extended_k;
    for (size_t i = 0; i < en; i++) f_mul(&FR, &a[i], &a[i], &d->t_evaluations[i % d->n_t]);
}
void orc_eval_polynomial(const fe *poly, size_t n, const fe *x, fe *out) {
    fe acc; memset(&acc, 0, sizeof acc);
    for (size_t i = n; i-- > 0;) { f_mul(&FR, &acc, &acc, x); f_add(&FR, &acc, &acc, &poly[i]); }
    *out = acc;
}

/* best_fft over the group G1 (the instance ParamsKZG::setup uses for g_to_lagrange): bit-reverse + iterative butterflies with
 * group_scale = scalar multiplication; then an optional scaling of every output (the n^-1 of the inverse transform) */
void orc_g1_fft(const g1a *in, uint32_t log_n, const fe *omega_mont, const fe *scale_mont_or_null, g1a *out) {
    size_t n = (size_t)1 << log_n;
    g1j *a = malloc(n * sizeof(g1j));
    for (size_t k = 0; k < n; k++) j_from_affine(&a[bitreverse(k, log_n)], &in[k]);
    fe *tw = malloc((n / 2 + 1) * sizeof(fe)); fe w = FR.r;
    for (size_t i = 0; i < n / 2; i++) { f_from_mont(&FR, &tw[i], &w); f_mul(&FR, &w, &w, omega_mont); }   /* canonical twiddles */
    size_t chunk = 2, twiddle_chunk = n / 2;
    for (uint32_t s = 0; s < log_n; s++) {
        for (size_t base = 0; base < n; base += chunk)
            for (size_t i = 0; i < chunk / 2; i++) {
                g1j t = a[base + chunk / 2 + i];
                if (i) { g1a ta; j_to_affine(&ta, &t); j_mul(&t, &ta, &tw[i * twiddle_chunk]); }
                g1j nt = t; f_neg(&FQ, &nt.y, &t.y);
                g1j x = a[base + i];
                j_add(&a[base + i], &x, &t);
                j_add(&a[base + chunk / 2 + i], &x, &nt);
            }
        chunk *= 2; twiddle_chunk /= 2;
    }
    for (size_t k = 0; k < n; k++) {
        if (scale_mont_or_null) { g1a t; j_to_affine(&t, &a[k]); fe sc; f_from_mont(&FR, &sc, scale_mont_or_null); j_mul(&a[k], &t, &sc); }
        j_to_affine(&out[k], &a[k]);
    }
    free(a); free(tw);
}

/* halo2_proofs src/arithmetic.rs kate_division(a, b): quotient of a(X) by (X - b); q has n-1 coefficients */
void orc_kate_division(const fe *a, size_t n, const fe *b, fe *q) {
    fe nb; f_neg(&FR, &nb, b);
    fe tmp; memset(&tmp, 0, sizeof tmp);
    for (size_t i = n - 1; i >= 1; i--) {
        fe lead; f_sub(&FR, &lead, &a[i], &tmp);
        q[i - 1] = lead;
        f_mul(&FR, &tmp, &lead, &nb);
    }
}

/* ------------------------------------------------ grand products */
/* halo2_proofs src/plonk/permutation/prover.rs Argument::commit, the body of ONE column set:
 * modified_values = prod_j (beta*sigma_j + gamma + v_j); batch_invert; *= prod_j (deltaomega_j*beta + gamma + v_j)
 * with deltaomega_j = delta^(j0+j) * omega^row; z[0] = last_z, z[row] = z[row-1]*modified[row-1];
 * z[n-blinding..] = randomness (drawn by the caller here).  delta_start = delta^(j0) (Montgomery). */
void orc_permutation_product(const fe *const *values, const fe *const *sigmas, size_t count, uint32_t k, const fe *beta, const fe *gamma,
                             const fe *delta_start, const fe *last_z, const fe *blinding, uint32_t blinding_factors, fe *z, fe *last_z_out) {
    size_t n = (size_t)1 << k;
    fe *mv = malloc(n * sizeof(fe)), *inv = malloc(n * sizeof(fe));
    for (size_t i = 0; i < n; i++) mv[i] = FR.r;
    for (size_t j = 0; j < count; j++)
        for (size_t i = 0; i < n; i++) { fe t; f_mul(&FR, &t, beta, &sigmas[j][i]); f_add(&FR, &t, &t, gamma); f_add(&FR, &t, &t, &values[j][i]); f_mul(&FR, &mv[i], &mv[i], &t); }
    /* batch_invert (zero stays zero) */
    { fe acc = FR.r; for (size_t i = 0; i < n; i++) { inv[i] = acc; if (!fe_is_zero(&mv[i])) f_mul(&FR, &acc, &acc, &mv[i]); }
      fe ai; f_inv(&FR, &ai, &acc);
      for (size_t i = n; i-- > 0;) { if (fe_is_zero(&mv[i])) continue; fe t; f_mul(&FR, &t, &ai, &inv[i]); f_mul(&FR, &ai, &ai, &mv[i]); mv[i] = t; } }
    fe omega; omega_for(&omega, k);
    fe delta = { BN254_FR_DELTA_M }, deltaomega0 = *delta_start;
    for (size_t j = 0; j < count; j++) {
        fe dw = deltaomega0;
        for (size_t i = 0; i < n; i++) { fe t; f_mul(&FR, &t, &dw, beta); f_add(&FR, &t, &t, gamma); f_add(&FR, &t, &t, &values[j][i]); f_mul(&FR, &mv[i], &mv[i], &t); f_mul(&FR, &dw, &dw, &omega); }
        f_mul(&FR, &deltaomega0, &deltaomega0, &delta);
    }
    z[0] = *last_z;
    for (size_t row = 1; row < n; row++) f_mul(&FR, &z[row], &z[row - 1], &mv[row - 1]);
    for (uint32_t b = 0; b < blinding_factors; b++) z[n - blinding_factors + b] = blinding[b];
    if (last_z_out) *last_z_out = z[n - (blinding_factors + 1)];
    free(mv); free(inv);
}
/* halo2_proofs src/plonk/lookup/prover.rs Permuted::commit_product */
void orc_lookup_product(const fe *cin, const fe *ctab, const fe *pin, const fe *ptab, uint32_t k, const fe *beta, const fe *gamma,
                        const fe *blinding, uint32_t blinding_factors, fe *z) {
    size_t n = (size_t)1 << k;
    fe *lp = malloc(n * sizeof(fe)), *pre = malloc(n * sizeof(fe));
    for (size_t i = 0; i < n; i++) { fe a, b; f_add(&FR, &a, beta, &pin[i]); f_add(&FR, &b, gamma, &ptab[i]); f_mul(&FR, &lp[i], &a, &b); }
    { fe acc = FR.r; for (size_t i = 0; i < n; i++) { pre[i] = acc; if (!fe_is_zero(&lp[i])) f_mul(&FR, &acc, &acc, &lp[i]); }
      fe ai; f_inv(&FR, &ai, &acc);
      for (size_t i = n; i-- > 0;) { if (fe_is_zero(&lp[i])) continue; fe t; f_mul(&FR, &t, &ai, &pre[i]); f_mul(&FR, &ai, &ai, &lp[i]); lp[i] = t; } }
    for (size_t i = 0; i < n; i++) { fe a, b; f_add(&FR, &a, &cin[i], beta); f_add(&FR, &b, &ctab[i], gamma); f_mul(&FR, &lp[i], &lp[i], &a); f_mul(&FR, &lp[i], &lp[i], &b); }
    fe state = FR.r;
    size_t keep = n - blinding_factors;
    for (size_t i = 0; i < keep; i++) { if (i > 0) f_mul(&FR, &state, &state, &lp[i - 1]); z[i] = state; }   /* once(1).chain(lp).scan */
    for (uint32_t b = 0; b < blinding_factors; b++) z[keep + b] = blinding[b];
    free(lp); free(pre);
}

/* halo2_proofs src/plonk/lookup/prover.rs permute_expression_pair.  Returns 0, or -1 for
 * Error::ConstraintSystemFailure (an input value that is not in the table).  blind_*: blinding_factors + 1 rows each. */
static int cmp_canon(const void *a, const void *b) {
    const fe *x = a, *y = b;
    for (int i = 3; i >= 0; i--) { if (x->l[i] < y->l[i]) return -1; if (x->l[i] > y->l[i]) return 1; }
    return 0;
}
int orc_lookup_permute(const fe *input, const fe *table, uint32_t k, uint32_t blinding_factors, const fe *blind_in, const fe *blind_tab,
                       fe *out_in, fe *out_tab) {
    size_t n = (size_t)1 << k, usable = n - (blinding_factors + 1);
    fe *pin = malloc(usable * sizeof(fe)), *tab = malloc(usable * sizeof(fe));
    for (size_t i = 0; i < usable; i++) { f_from_mont(&FR, &pin[i], &input[i]); f_from_mont(&FR, &tab[i], &table[i]); }
    qsort(pin, usable, sizeof(fe), cmp_canon);                 /* permuted_input_expression.sort() */
    qsort(tab, usable, sizeof(fe), cmp_canon);                 /* BTreeMap<value, count> == sorted multiset */
    char *taken = calloc(usable, 1);
    size_t *repeated = malloc(usable * sizeof(size_t)), nrep = 0;
    fe *ptab = calloc(usable, sizeof(fe));
    size_t tpos = 0;
    int rc = 0;
    for (size_t row = 0; row < usable; row++) {
        if (row == 0 || cmp_canon(&pin[row], &pin[row - 1]) != 0) {
            while (tpos < usable && cmp_canon(&tab[tpos], &pin[row]) < 0) tpos++;
            if (tpos >= usable || cmp_canon(&tab[tpos], &pin[row]) != 0) { rc = -1; break; }
            taken[tpos++] = 1;                                   /* *count -= 1 */
            ptab[row] = pin[row];
        } else repeated[nrep++] = row;
    }
    if (!rc) {
        for (size_t t = 0; t < usable; t++)                      /* leftover table values ascending, rows popped from the back */
            if (!taken[t]) ptab[repeated[--nrep]] = tab[t];
        for (size_t i = 0; i < usable; i++) { f_to_mont(&FR, &out_in[i], &pin[i]); f_to_mont(&FR, &out_tab[i], &ptab[i]); }
        for (size_t i = 0; i < blinding_factors + 1; i++) { out_in[usable + i] = blind_in[i]; out_tab[usable + i] = blind_tab[i]; }
    }
    free(pin); free(tab); free(taken); free(repeated); free(ptab);
    return rc;
}

#include "evaluate_h_oracle.inc"
