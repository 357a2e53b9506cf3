"""Evaluation phase: eval_polynomial and kate_division — oracle vs definition, product vs oracle."""
import random

import numpy as np
import pytest

import parity_cases as pc
import zk_dcap_verifier_amd as z


def test_oracle_kate_division_and_eval_match_definition(orc, pyref):
    p, rnd = pyref, random.Random(8)
    n, R = 37, p.R
    a = [rnd.randrange(R) for _ in range(n)]
    b = rnd.randrange(R)
    M = orc.fr_from_ints
    q = orc.fr_to_ints(orc.kate_division(M(a), M([b])[0]))
    # (X - b) * q(X) + a(b) == a(X)
    ab = p.poly_eval(a, b)
    assert orc.fr_to_ints(orc.eval_polynomial(M(a), M([b])[0]).reshape(1, 4))[0] == ab
    prod = [0] * n
    for i, c in enumerate(q):
        prod[i + 1] = (prod[i + 1] + c) % R
        prod[i] = (prod[i] - b * c) % R
    prod[0] = (prod[0] + ab) % R
    assert prod == a


def _check(be, orc, pyref, n, count, seed):
    polys = [pc.rand_fr(orc, pyref, n, seed + i) for i in range(count)]
    pts = pc.rand_fr(orc, pyref, count, seed + 100)
    pts[0] = 0
    if count > 1:
        pts[1] = orc.fr_from_ints([1])[0]
    d = [be.to_device(c) for c in polys]
    got = be.eval_polynomial_batch_dev(d, n, pts)
    for i in range(count):
        assert (got[i] == orc.eval_polynomial(polys[i], pts[i])).all(), i
    for dd in d:
        dd.free()
    if n >= 2:
        assert (z.arithmetic.kate_division(polys[0], pts[-1], backend=be) == orc.kate_division(polys[0], pts[-1])).all()
    assert (z.arithmetic.eval_polynomial(polys[0], pts[-1], backend=be) == orc.eval_polynomial(polys[0], pts[-1])).all()


@pytest.mark.parametrize("n,count", [(1, 1), (2, 2), (100, 3), (4096, 2), (10000, 3)])
def test_emulated_eval_phase(emu, orc, pyref, n, count):
    _check(emu, orc, pyref, n, count, seed=n)


@pytest.mark.gpu
@pytest.mark.parametrize("n,count", [(1, 1), (3, 2), (1 << 12, 5), (100003, 7), (1 << 19, 4), ((1 << 20) + 4099, 2)])   # the last one: kate_division carries over two rounds of workgroups
def test_gpu_eval_phase(gpu, orc, pyref, n, count):
    _check(gpu, orc, pyref, n, count, seed=n)


def _check_lincomb(be, orc, pyref, n, count, seed):
    """zk_fr_lincomb_dev (SHPLONK's polynomial combinations): out = sum_j s_j * p_j, a scalar equal to one (skips its product), output aliasing an input"""
    R = pyref.R
    polys = [pc.rand_fr(orc, pyref, n, seed + i) for i in range(count)]
    sc = pc.rand_fr(orc, pyref, count, seed + 50)
    sc[0] = orc.fr_from_ints([1])[0]
    want_ints = [0] * n
    si = orc.fr_to_ints(sc)
    for pj, s_ in zip(polys, si):
        want_ints = [(w + s_ * v) % R for w, v in zip(want_ints, orc.fr_to_ints(pj))]
    want = orc.fr_from_ints(want_ints)
    d = [be.to_device(c) for c in polys]
    out = be.alloc(n * 32)
    be.fr_lincomb_dev(d, sc, n, out)
    assert (out.download((n, 4)) == want).all()
    be.fr_lincomb_dev(d, sc, n, d[-1])                                   # in place on the last input
    assert (d[-1].download((n, 4)) == want).all()
    for dd in d + [out]:
        dd.free()


def test_emulated_lincomb(emu, orc, pyref):
    _check_lincomb(emu, orc, pyref, 777, 4, seed=3)


@pytest.mark.gpu
@pytest.mark.parametrize("n,count", [(1, 1), (100003, 3), (1 << 18, 9)])
def test_gpu_lincomb(gpu, orc, pyref, n, count):
    _check_lincomb(gpu, orc, pyref, n, count, seed=n % 1000)
