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
