"""csrc/field29.cuh — the carry-free arithmetic on 9 limbs of 29 bits (Montgomery radix 2^261) that the bucket accumulation runs on — compiled for the CPU and driven
on RAW limbs against an exact Python model: the model recomputes every column sum of a product as a Python integer, asserts it fits the 64-bit accumulator for the
operand shapes the kernel produces (xyzz29_madd_fast: ec.cuh lists them), and the C result must equal the model's, limb for limb.  Then the chain itself: a bucket's
mixed additions on 29-bit limbs give the coordinates of the canonical chain, special cases included."""
import ctypes as C
import random

import numpy as np
import pytest

from conftest import HOST_SO

M29 = (1 << 29) - 1
RAD = 1 << 261


@pytest.fixture(scope="module")
def hh(built):
    return C.CDLL(HOST_SO)


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def limbs_of(x, loose=None, rnd=None):
    """x as 9 limbs; `loose` (bits) re-distributes value between neighbours so that limbs use up to that many bits (same integer)"""
    l = [(x >> (29 * i)) & M29 for i in range(8)] + [x >> 232]
    if loose:
        for i in range(8, 0, -1):
            room = int(2 ** loose) - 1 - l[i - 1]
            take = min(l[i], room >> 29, rnd.randrange(0, 8))
            l[i] -= take
            l[i - 1] += take << 29
    return l


def value(l):
    return sum(int(v) << (29 * i) for i, v in enumerate(l))


def arr(rows):
    return np.array(rows, dtype=np.uint32).reshape(len(rows), 9)


def model_mul(mod, pairs):
    """sum of a*b over `pairs` of limb vectors, times 2^-261: exact product scanning, every column checked against 2^64"""
    pl = limbs_of(mod)
    inv = (-pow(mod, -1, 1 << 29)) % (1 << 29)
    acc, m, r = 0, [], [0] * 9
    for k in range(17):
        for a, b in pairs:
            for i in range(max(0, k - 8), min(k, 8) + 1):
                acc += a[i] * b[k - i]
        for i in range(max(0, k - 8), min(k, 9)):
            if i < len(m) and k - i <= 8:
                acc += m[i] * pl[k - i]
        if k < 9:
            m.append((acc & M29) * inv & M29)
            acc += m[k] * pl[0]
            assert acc & M29 == 0
        else:
            r[k - 9] = acc & M29
        assert acc < 1 << 64, ("column overflows the 64-bit accumulator", k, acc.bit_length())
        acc >>= 29
    r[8] = acc
    assert acc < 1 << 32
    return r


def bias(mod, K, LOG):
    t = K * mod - sum(1 << (LOG + 29 * i) for i in range(8))
    assert t > 0
    return [(1 << LOG) + ((t >> (29 * i)) & M29) for i in range(8)] + [t >> 232]


def run_raw(hh, field, op, *ops, n=None):
    n = len(ops[0])
    bufs = [arr(o) for o in ops] + [arr([[0] * 9] * n)] * (4 - len(ops))
    out = np.zeros((n, 9), dtype=np.uint32)
    hh.hh_f29_raw(C.c_int(field), C.c_int(op), P(bufs[0]), P(bufs[1]), P(bufs[2]), P(bufs[3]), P(out), C.c_size_t(n))
    return [[int(v) for v in row] for row in out]


@pytest.mark.parametrize("field", [0, 1])
def test_products_on_raw_limbs_match_the_exact_model_at_the_limb_bounds(hh, pyref, field):
    mod = pyref.P if field == 0 else pyref.R
    rnd = random.Random(29 + field)
    vals = [0, 1, mod - 1, mod, 2 * mod - 1, 7 * mod - 1, 10 * mod - 3, 32 * mod - 1, (1 << 261) - 1] + [rnd.randrange(0, 12 * mod) for _ in range(40)]
    # mul: N-form x N-form, 2^30 x 2^30 (one un-normalised sum each), 2^31 x N-form (a biased difference against a product output)
    for la, lb in ((None, None), (30, 30), (31, None)):
        A = [limbs_of(rnd.choice(vals), la, rnd) for _ in range(60)]
        B = [limbs_of(rnd.choice(vals), lb, rnd) for _ in range(60)]
        A[0] = [(1 << (la or 29)) - 1] * 8 + [(1 << 27) - 1]            # every limb at its bound
        B[0] = [(1 << (lb or 29)) - 1] * 8 + [(1 << 27) - 1]
        got = run_raw(hh, field, 0, A, B)
        for a, b, g in zip(A, B, got):
            want = model_mul(mod, [(a, b)])
            assert g == want and value(g) * RAD % mod == value(a) * value(b) % mod and value(g) < value(a) * value(b) // RAD + mod + 1
    # sqr: limbs below 2^30
    A = [limbs_of(rnd.choice(vals), 30, rnd) for _ in range(60)] + [[(1 << 30) - 1] * 8 + [(1 << 27) - 1]]
    for a, g in zip(A, run_raw(hh, field, 1, A)):
        assert g == model_mul(mod, [(a, a)])
    # mul2 as the mixed addition uses it: R (N-form) * D1 (N-form) + Y1 (N-form product output) * D2 (limbs below 2^30 + 2^29)
    top = [M29 + 8] * 8 + [(1 << 26) - 1]
    rows = [(limbs_of(rnd.choice(vals)), limbs_of(rnd.choice(vals)), limbs_of(rnd.choice(vals)), limbs_of(rnd.choice(vals), 30, rnd)) for _ in range(40)]
    rows.append((top, top, [M29] * 8 + [(1 << 23) - 1], [(1 << 30) + (1 << 29) - 1] * 8 + [(1 << 24) - 1]))
    got = run_raw(hh, field, 2, *[[r[j] for r in rows] for j in range(4)])
    for (a, b, c, d), g in zip(rows, got):
        assert g == model_mul(mod, [(a, b), (c, d)])
        assert value(g) * RAD % mod == (value(a) * value(b) + value(c) * value(d)) % mod


@pytest.mark.parametrize("field", [0, 1])
def test_biased_differences_and_the_carry_round(hh, pyref, field):
    mod = pyref.P if field == 0 else pyref.R
    rnd = random.Random(31 + field)
    for op, K, LOG, sub_bits in ((4, 8, 30, 29), (5, 3, 30, 29), (6, 5, 31, 31)):
        kp = bias(mod, K, LOG)
        assert value(kp) == K * mod and all((1 << LOG) <= v < (1 << LOG) + (1 << 29) for v in kp[:8])
        A = [limbs_of(rnd.randrange(0, 2 * mod)) for _ in range(50)]
        # subtrahends up to the documented bound: value below (K - 1) p, limbs up to 2^LOG (a sum PPP + 2 Q for LOG = 31, an N-form value otherwise)
        B = [limbs_of(rnd.randrange(0, (K - 1) * mod), 30 if sub_bits == 29 else 31, rnd) for _ in range(50)]
        B = [[min(v, (1 << LOG)) for v in b[:8]] + [b[8]] for b in B]
        B[0] = limbs_of((K - 1) * mod - 1)
        got = run_raw(hh, field, op, A, B)
        for a, b, g in zip(A, B, got):
            assert all(0 <= x + k - y < 1 << 32 for x, k, y in zip(a, kp, b))                # limb-wise non-negative, no wrap
            assert g == [x + k - y for x, k, y in zip(a, kp, b)] and value(g) == value(a) + K * mod - value(b)
    kp = bias(mod, 3, 30)
    B = [limbs_of(rnd.randrange(0, 2 * mod)) for _ in range(20)]
    for b, g in zip(B, run_raw(hh, field, 7, B)):
        assert value(g) == 3 * mod - value(b) and all(v < (1 << 30) + (1 << 29) for v in g[:8])
    # carry: any limbs below 2^32 -> N-form, same integer
    A = [[rnd.randrange(0, 1 << 32) for _ in range(8)] + [rnd.randrange(0, 1 << 26)] for _ in range(50)] + [[(1 << 32) - 1] * 8 + [5]]
    for a, g in zip(A, run_raw(hh, field, 3, A)):
        assert value(g) == value(a) and all(v < (1 << 29) + 8 for v in g[:8])
    assert value(run_raw(hh, field, 9, [[0] * 9])[0]) == RAD % mod                            # one() = 2^261 mod p


@pytest.mark.parametrize("field", [0, 1])
def test_between_the_two_montgomery_forms(hh, orc, pyref, field):
    mod = pyref.P if field == 0 else pyref.R
    rnd = random.Random(33 + field)
    R256 = 1 << 256
    xs = [0, 1, mod - 1, mod - 2, (1 << 253) + 5] + [rnd.randrange(0, mod) for _ in range(60)]
    ys = [mod - 1, 0, 1, mod - 2, 7] + [rnd.randrange(0, mod) for _ in range(60)]
    A, B = orc.ints_to_limbs(xs), orc.ints_to_limbs(ys)

    def run(op):
        o, o9 = np.zeros_like(A), np.zeros((len(xs), 9), dtype=np.uint32)
        hh.hh_f29_forms(C.c_int(field), C.c_int(op), P(A), P(B), P(o), P(o9), C.c_size_t(len(xs)))
        return orc.limbs_to_ints(o), [[int(v) for v in row] for row in o9]
    back, ent = run(0)
    assert back == xs and all(value(e) % mod == x * 32 % mod and value(e) < 2 * mod for e, x in zip(ent, xs))       # enter: x 2^256 -> x 2^261, below 2 p
    rinv = pow(R256, -1, mod)
    for op in (1, 2):
        got, _ = run(op)
        assert got == [x * y * rinv % mod for x, y in zip(xs, ys)]                                                 # the library's Montgomery product, canonical
    got, _ = run(4)
    assert got == [x * x * rinv % mod for x in xs]
    got, l9 = run(3)
    assert got == xs and all(value(l) == x and all(v <= M29 for v in l[:8]) for l, x in zip(l9, xs))               # limb conversion is exact


def model_shoup(mod, a, w, wq):
    """the generated body, column for column: q from columns 7 .. 16 of a * wq, r = low 261 bits of a * w + q * (2^261 - p)"""
    npl = limbs_of(RAD - mod)
    acc, q = 0, [0] * 9
    for k in range(7, 17):
        acc += sum(a[i] * wq[k - i] for i in range(max(0, k - 8), min(k, 8) + 1))
        assert acc < 1 << 64
        if k >= 9:
            q[k - 9] = acc & M29
        acc >>= 29
    q[8] = acc
    assert acc < 1 << 32
    acc, r = 0, [0] * 9
    for k in range(9):
        acc += sum(a[i] * w[k - i] + q[i] * npl[k - i] for i in range(k + 1))
        assert acc < 1 << 64
        r[k] = acc & M29
        acc >>= 29
    return r, q


@pytest.mark.parametrize("field", [0, 1])
def test_shoup_product_with_a_precomputed_quotient(hh, orc, pyref, field):
    """mul_shoup: a * w mod p for a constant w, wq = floor(w 2^261 / p) — what the NTT butterflies multiply their twiddles with.  shoup_quotient is exact; the product equals the
    column model limb for limb, is congruent to a * w, below 3 p, for every a below 2^261 with limbs up to 3 * 2^30 — a biased difference of N-form values, the loosest operand a
    butterfly multiplies (q is the true quotient or one below).  mul_shoup = shoup_r(shoup_q(..)), the two halves the NTT kernels call."""
    mod = pyref.P if field == 0 else pyref.R
    rnd = random.Random(41 + field)
    ws = [0, 1, 2, mod - 1, mod - 2, (mod + 1) // 2, 1 << 253] + [rnd.randrange(0, mod) for _ in range(80)]
    W = orc.ints_to_limbs(ws)
    o, o9 = np.zeros_like(W), np.zeros((len(ws), 9), dtype=np.uint32)
    hh.hh_f29_forms(C.c_int(field), C.c_int(5), P(W), P(W), P(o), P(o9), C.c_size_t(len(ws)))
    wq = [[int(v) for v in row] for row in o9]
    assert [value(l) for l in wq] == [w * RAD // mod for w in ws] and all(v <= M29 for l in wq for v in l)
    avals = [0, 1, mod - 1, mod, 3 * mod - 1, 32 * mod, 150 * mod, RAD - 1] + [rnd.randrange(0, RAD) for _ in range(40)] + [rnd.randrange(0, 8 * mod) for _ in range(40)]
    A, Wl, Wq = [], [], []
    for j in range(len(ws)):
        for loose in (None, 30, 31.58):
            a = rnd.choice(avals)
            A.append(limbs_of(a, loose, rnd)); Wl.append(limbs_of(ws[j])); Wq.append(wq[j])
    A.append([(1 << 30) - 1] * 8 + [(1 << 28) - 1]); Wl.append(limbs_of(mod - 1)); Wq.append(limbs_of((mod - 1) * RAD // mod))     # every limb at its bound (the integer is still below 2^261)
    assert value(A[-1]) < RAD
    A.append([3 * (1 << 30)] * 8 + [(1 << 26)]); Wl.append(limbs_of(mod - 1)); Wq.append(limbs_of((mod - 1) * RAD // mod))           # ... and at the bound of a biased difference
    assert value(A[-1]) < RAD
    got = run_raw(hh, field, 10, A, Wl, Wq)
    low = 0
    for a, w, q_, g in zip(A, Wl, Wq, got):
        want, q = model_shoup(mod, a, w, q_)
        va, vw = value(a), value(w)
        assert g == want and all(v <= M29 for v in g)
        assert value(g) % mod == va * vw % mod and value(g) < 3 * mod
        exact = va * value(q_) // RAD
        assert value(q) in (exact, exact - 1) and value(g) == va * vw - value(q) * mod
        low += value(q) != exact
    print("quotient one below the exact one in", low, "of", len(A))


def _affine(orc, p, xyzz16):
    x, y, zz, zzz = orc.limbs_to_ints(xyzz16.reshape(4, 4))
    if zz == 0:
        return None
    rinv = pow(1 << 256, -1, p.P)
    x, y, zz, zzz = (v * rinv % p.P for v in (x, y, zz, zzz))
    return x * pow(zz, -1, p.P) % p.P, y * pow(zzz, -1, p.P) % p.P


def test_bucket_chain_on_29_bit_limbs_equals_the_canonical_chain(hh, orc, pyref):
    """the coordinates — not only the point — of the 29-bit chain equal the canonical chain's (the same rational formulas over the same field), through the fast step
    and through every rare case: identity bases, the first point, a doubling, P then -P (cancellation to the identity) and a chain that continues after it"""
    p, rnd = pyref, random.Random(11)
    pts = [p.g1_mul(p.G1_GEN, rnd.randrange(1, p.R)) for _ in range(48)]
    cases = [(pts[:40], [rnd.randrange(2) for _ in range(40)], 1),                              # the plain chain: only the first point is "rare"
             ([pts[0], pts[0]] + pts[1:9], [0] * 10, 2),                                         # doubling at step 2
             ([pts[0], pts[1], None, pts[2], None], [0, 1, 0, 0, 0], 1),                         # identity bases are skipped
             ([pts[3], pts[4], pts[5], pts[5], pts[6]], [0, 0, 0, 1, 0], 1),                     # P5 then -P5 is not the same x as the accumulator: plain steps
             ([pts[7], pts[7], pts[8]], [0, 1, 0], 3),                                           # P - P = identity, then the chain restarts from the identity
             ([pts[9], pts[10], pts[9], pts[10], pts[11]], [0, 0, 1, 1, 0], 3)]                  # ... + P + Q - P - Q: cancellation at the last-but-one step
    for seq, neg, rare_want in cases:
        a, ng = orc.g1_affine_from_ints(seq), np.array(neg, dtype=np.uint8)
        got, canon, rare = np.zeros(16, dtype=np.uint64), np.zeros(16, dtype=np.uint64), C.c_uint32()
        hh.hh_xyzz29_sum(P(a), P(ng), C.c_size_t(len(seq)), P(got), C.byref(rare))
        hh.hh_xyzz_sum(P(a), P(ng), C.c_size_t(len(seq)), P(canon))
        want = None
        for q_, s_ in zip(seq, neg):
            want = p.g1_add(want, p.g1_neg(q_) if s_ else q_)
        assert _affine(orc, p, got) == want == _affine(orc, p, canon)
        assert (got == canon).all(), "coordinates differ from the canonical chain's"
        assert rare.value == rare_want, (rare.value, rare_want)
