"""The product's __host__ __device__ limb routines (field.cuh / ec.cuh) compiled for the CPU,
checked against the oracle.  Covers the arithmetic the kernels inline; no GPU needed."""
import ctypes as C
import random

import numpy as np
import pytest

from conftest import HOST_SO


@pytest.fixture(scope="module")
def hh(built):
    return C.CDLL(HOST_SO)


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def v2(hh, name, A, B):
    out = np.empty_like(A)
    getattr(hh, name)(P(A), P(B), P(out), C.c_size_t(len(A)))
    return out


def test_field_limb_ops(hh, orc, pyref):
    rnd = random.Random(2)
    for mod, pre in ((pyref.R, "fr"), (pyref.P, "fq")):
        edge = [0, 1, mod - 1, mod - 2, 1 << 253, pyref.mont_r(mod), mod >> 1]
        a = [rnd.randrange(mod) for _ in range(3000)] + edge + edge
        b = [rnd.randrange(mod) for _ in range(3000)] + edge + edge[::-1]
        A, B = orc.ints_to_limbs(a), orc.ints_to_limbs(b)
        for op in ("mul", "add", "sub"):
            assert (v2(hh, f"hh_{pre}_{op}", A, B) == getattr(orc, f"{pre}_{op}")(A, B)).all(), (pre, op)
    A = orc.ints_to_limbs([rnd.randrange(pyref.R) for _ in range(16)] + [0])
    out = np.empty_like(A)
    hh.hh_fr_inv(P(A), P(out), C.c_size_t(len(A)))
    assert (out == orc.fr_inv(A)).all()
    hh.hh_fr_neg(P(A), P(out), C.c_size_t(len(A)))
    assert (out == orc.fr_sub(np.zeros_like(A), A)).all()


def test_fused_two_product_reduction(hh, orc, pyref):
    rnd = random.Random(9)
    P_ = pyref.P
    edge = [0, 1, P_ - 1, P_ - 2, P_ >> 1]
    vals = [[rnd.randrange(P_) for _ in range(500)] + edge for _ in range(4)]
    vals[1] = vals[1][:500] + edge[::-1]
    A, B, C_, D = (orc.ints_to_limbs(v) for v in vals)
    out = np.empty_like(A)
    hh.hh_fq_mul2_sub(P(A), P(B), P(C_), P(D), P(out), C.c_size_t(len(A)))
    assert (out == orc.fq_sub(orc.fq_mul(A, B), orc.fq_mul(C_, D))).all()


def xyzz_to_affine(orc, pyref, x):
    X, Y, ZZ, ZZZ = orc.fq_to_ints(np.asarray(x).reshape(4, 4))
    if ZZ == 0:
        return None
    return (X * pow(ZZ, -1, pyref.P) % pyref.P, Y * pow(ZZZ, -1, pyref.P) % pyref.P)


def test_xyzz_group_law_including_special_cases(hh, orc, pyref):
    p, rnd = pyref, random.Random(6)
    pts = [p.g1_mul(p.G1_GEN, rnd.randrange(1, p.R)) for _ in range(24)]
    seq = [pts[0], pts[0]] + pts[1:] + [None, pts[3], pts[2]]   # doubling first, identity base, repeats
    neg = [0, 0] + [rnd.randrange(2) for _ in pts[1:]] + [0, 1, 0]
    arr, ng = orc.g1_affine_from_ints(seq), np.array(neg, dtype=np.uint8)
    out = np.zeros(16, dtype=np.uint64)
    hh.hh_xyzz_sum(P(arr), P(ng), C.c_size_t(len(seq)), P(out))
    want = None
    for q, s in zip(seq, neg):
        want = p.g1_add(want, p.g1_neg(q) if s else q)
    assert xyzz_to_affine(orc, p, out) == want
    # P + (-P) = identity through the mixed-add path
    two = orc.g1_affine_from_ints([pts[5], pts[5]])
    o2 = np.zeros(16, dtype=np.uint64)
    hh.hh_xyzz_sum(P(two), P(np.array([0, 1], dtype=np.uint8)), C.c_size_t(2), P(o2))
    assert xyzz_to_affine(orc, p, o2) is None
    # full add: doubling branch, identity operands
    o3 = np.zeros(16, dtype=np.uint64)
    hh.hh_xyzz_add(P(out), P(out), P(o3))
    assert xyzz_to_affine(orc, p, o3) == p.g1_add(want, want)
    o4 = np.zeros(16, dtype=np.uint64)
    hh.hh_xyzz_add(P(o3), P(o2), P(o4))
    assert xyzz_to_affine(orc, p, o4) == p.g1_add(want, want)
    o5 = np.zeros(16, dtype=np.uint64)
    hh.hh_xyzz_dbl(P(out), P(o5))
    assert xyzz_to_affine(orc, p, o5) == p.g1_add(want, want)


def test_redundant_range_arithmetic_on_the_range_boundaries(hh, orc, pyref):
    """field.cuh's lazy forms (the NTT butterflies in [0, 4p), the bucket accumulation in [0, 2p)): for inputs ON the boundaries of the documented ranges —
    0, 1, p - 1, p, p + 1, 2p - 1, 2p, 4p - 1 — and random ones, every result lies in the documented output range and is congruent to the exact answer."""
    q, rnd = pyref.P, random.Random(11)
    Rinv = pow(1 << 256, -1, q)
    lim = lambda xs: orc.ints_to_limbs(xs)
    ints = lambda arr: orc.limbs_to_ints(arr)

    def run(op, a, b=None, c=None, d=None):
        A = lim(a)
        B, C_, D = (lim(x) if x is not None else A for x in (b, c, d))
        out = np.empty_like(A)
        hh.hh_fq_lazy(C.c_int(op), P(A), P(B), P(C_), P(D), P(out), C.c_size_t(len(a)))
        return ints(out)
    e2 = [0, 1, q - 1, q, q + 1, 2 * q - 1]                           # [0, 2q)
    e4 = e2 + [2 * q, 2 * q + 1, 3 * q, 4 * q - 1]                    # [0, 4q)
    r2 = e2 + [rnd.randrange(2 * q) for _ in range(400)]
    r4 = e4 + [rnd.randrange(4 * q) for _ in range(400)]
    canon = [0, 1, q - 1] + [rnd.randrange(q) for _ in range(len(r4) - 3)]
    pairs2 = [(a, b) for a in e2 for b in e2] + [(rnd.randrange(2 * q), rnd.randrange(2 * q)) for _ in range(300)]
    a2, b2 = [p_[0] for p_ in pairs2], [p_[1] for p_ in pairs2]
    for got, a, b in zip(run(0, r4, canon), r4, canon):               # mul_lazy: [0, 4q) x [0, q) -> [0, 2q)
        assert got < 2 * q and got % q == a * b * Rinv % q
    for got, a, b in zip(run(0, a2, b2), a2, b2):                     # ... and [0, 2q) x [0, 2q) -> [0, 2q) (the accumulate chain)
        assert got < 2 * q and got % q == a * b * Rinv % q
    for got, a in zip(run(1, r2), r2):
        assert got < 2 * q and got % q == a * a * Rinv % q
    for got, a, b in zip(run(2, a2, b2), a2, b2):
        assert got < 2 * q and got % q == (a - b) % q
    for got, a in zip(run(3, r2), r2):
        assert got < 2 * q and got % q == 2 * a % q
    for got, a in zip(run(4, r2), r2):
        assert got <= 2 * q and got % q == -a % q
    top = [2 * q] * 8 + [rnd.randrange(2 * q + 1) for _ in range(300)]   # mul2_add_2p takes the closed range [0, 2q]
    aa, bb, cc, dd = ([rnd.choice(top) for _ in range(400)] for _ in range(4))
    aa[0] = bb[0] = cc[0] = dd[0] = 2 * q
    for got, a, b, c, d in zip(run(5, aa, bb, cc, dd), aa, bb, cc, dd):
        assert got < 2 * q and got % q == (a * b + c * d) * Rinv % q
    for got, a in zip(run(6, r4), r4):
        assert got < 2 * q and got % q == a % q
    for got, a, b in zip(run(7, a2, b2), a2, b2):
        assert got < 4 * q and got == a + b
    for got, a, b in zip(run(8, a2, b2), a2, b2):
        assert 0 < got < 4 * q and got == a + 2 * q - b
    for got, a in zip(run(9, r4), r4):
        assert got == a % q
    for got, a in zip(run(10, r2), r2):
        assert got == (1 if a % q == 0 else 0)
    for got, a, b in zip(run(11, r4, canon), r4, canon):              # the full product accepts a redundant left operand (ntt_post)
        assert got == a * b * Rinv % q
    # mul_shoup_lazy (the final NTT pass's twiddle products): a * w mod p, no Montgomery factor, for ANY a below 2^256 (the butterflies hand it [0, 4p)) and canonical w
    # with wq = floor(w 2^256 / p); result in [0, 2p).  Both fields; the edges of a's range and w in {0, 1, p - 1, ...} included.
    for op, mod in ((12, pyref.P), (13, pyref.R)):
        ws = [0, 1, 2, mod - 1, mod - 2, (mod + 1) // 2] + [rnd.randrange(mod) for _ in range(300)]
        As = [0, 1, mod - 1, mod, 2 * mod - 1, 2 * mod, 4 * mod - 1, (1 << 256) - 1] + [rnd.randrange(4 * mod) for _ in range(200)] + [rnd.randrange(1 << 256) for _ in range(98)]
        wqs = [w * (1 << 256) // mod for w in ws]
        assert run(op + 2, [w * (1 << 256) % mod for w in ws]) == wqs                               # shoup_quotient: from w's library form, exact
        for got, a, w in zip(run(op, As, ws, wqs), As, ws):
            assert got < 2 * mod and got % mod == a * w % mod, (op, a, w)


def test_lazy_mixed_addition_chain_equals_the_canonical_one(hh, orc, pyref):
    p, rnd = pyref, random.Random(8)
    pts = [p.g1_mul(p.G1_GEN, rnd.randrange(1, p.R)) for _ in range(40)]
    seq = [pts[0], pts[0]] + pts[1:] + [None, pts[3], pts[2], pts[7], pts[7]]       # doubling first, identity base, repeats, P then P again late in the chain
    neg = [0, 0] + [rnd.randrange(2) for _ in pts[1:]] + [0, 1, 0, 0, 1]
    arr, ng = orc.g1_affine_from_ints(seq), np.array(neg, dtype=np.uint8)
    lazy, canon = np.zeros(16, dtype=np.uint64), np.zeros(16, dtype=np.uint64)
    hh.hh_xyzz_sum_lazy(P(arr), P(ng), C.c_size_t(len(seq)), P(lazy))
    hh.hh_xyzz_sum(P(arr), P(ng), C.c_size_t(len(seq)), P(canon))
    want = None
    for q_, s_ in zip(seq, neg):
        want = p.g1_add(want, p.g1_neg(q_) if s_ else q_)
    assert xyzz_to_affine(orc, p, lazy) == xyzz_to_affine(orc, p, canon) == want
    assert all(v < p.P for v in orc.limbs_to_ints(lazy.reshape(4, 4)))             # normalised coordinates
    # the full addition in the lazy range: a + b + b, a + a + a (doubling branch first), identity operands
    o3, o4 = np.zeros(16, dtype=np.uint64), np.zeros(16, dtype=np.uint64)
    other = np.zeros(16, dtype=np.uint64)
    hh.hh_xyzz_sum(P(orc.g1_affine_from_ints(pts[20:30])), P(np.zeros(10, dtype=np.uint8)), C.c_size_t(10), P(other))
    w2 = None
    for q_ in pts[20:30]:
        w2 = p.g1_add(w2, q_)
    hh.hh_xyzz_add_lazy(P(canon), P(other), P(o3))
    assert xyzz_to_affine(orc, p, o3) == p.g1_add(p.g1_add(want, w2), w2)
    hh.hh_xyzz_add_lazy(P(canon), P(canon), P(o4))
    assert xyzz_to_affine(orc, p, o4) == p.g1_add(p.g1_add(want, want), want)
    ident = np.zeros(16, dtype=np.uint64)
    hh.hh_xyzz_add_lazy(P(ident), P(other), P(o4))
    assert xyzz_to_affine(orc, p, o4) == p.g1_add(w2, w2)
    two = orc.g1_affine_from_ints([pts[5], pts[9], pts[5], pts[9]])                 # ... + P + Q - P - Q = identity through the lazy path
    o2 = np.zeros(16, dtype=np.uint64)
    hh.hh_xyzz_sum_lazy(P(two), P(np.array([0, 0, 1, 1], dtype=np.uint8)), C.c_size_t(4), P(o2))
    assert xyzz_to_affine(orc, p, o2) is None


def test_chained_copy_rows_equals_scalar_copy_and_is_a_bijection():
    """Assembly.copy_rows must only take its vectorised path for cells that are their own (singleton) cycle: `sizes` is kept for cycle
    representatives only, so after copy_rows(A, B) the B cells still read size 1 — a chained copy_rows(B, C) has to fall back to copy()."""
    import numpy as np
    from zk_dcap_verifier_amd import plonk
    from zk_dcap_verifier_amd.plonk import ADVICE
    cs = plonk.ConstraintSystem(num_advice_columns=4)
    for c in range(4):
        cs.enable_equality(ADVICE, c)
    rows = [0, 1, 2, 3, 5]
    fast, slow = plonk.Assembly(cs, 3), plonk.Assembly(cs, 3)
    fast.copy_rows((ADVICE, 0), (ADVICE, 1), rows)
    fast.copy_rows((ADVICE, 1), (ADVICE, 2), rows)
    fast.copy_rows((ADVICE, 3), (ADVICE, 0), [5, 6])
    for r in rows:
        slow.copy((ADVICE, 0, r), (ADVICE, 1, r))
    for r in rows:
        slow.copy((ADVICE, 1, r), (ADVICE, 2, r))
    for r in (5, 6):
        slow.copy((ADVICE, 3, r), (ADVICE, 0, r))
    assert (fast.map_c == slow.map_c).all() and (fast.map_r == slow.map_r).all()
    succ = set(zip(fast.map_c.ravel().tolist(), fast.map_r.ravel().tolist()))
    assert len(succ) == 4 * 8                                       # the successor map is a permutation of the cells
    # the cycle through (0, 0) is A -> B -> C -> A in some rotation: three cells
    seen, cell = [], (0, 0)
    while cell not in seen:
        seen.append(cell)
        cell = (int(fast.map_c[cell]), int(fast.map_r[cell]))
    assert sorted(seen) == [(0, 0), (1, 0), (2, 0)]


def test_rand_fr_array_is_uniform_below_r_and_reproducible():
    import numpy as np
    from zk_dcap_verifier_amd import fields as F
    a = F.rand_fr_array(np.random.default_rng(5), 4000)
    b = F.rand_fr_array(np.random.default_rng(5), 4000)
    assert (a == b).all()
    vals = [F.unlimbs(x) for x in a]
    assert all(v < F.R_MOD for v in vals)
    top = sum(v >= (1 << 253) for v in vals) / len(vals)            # (r - 2^253) / r = 0.339 of the field lies above 2^253
    assert 0.30 < top < 0.38
    edge = np.array([F.limbs(F.R_MOD), F.limbs(F.R_MOD - 1), F.limbs(F.R_MOD + 1), F.limbs(0)], dtype=np.uint64)
    assert F._below_r(edge).tolist() == [False, True, False, True]
    c = F.rand_fr_array(F.OsRng(), 100)
    assert c.shape == (100, 4) and all(F.unlimbs(x) < F.R_MOD for x in c)
