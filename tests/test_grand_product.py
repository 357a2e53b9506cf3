"""Permutation / lookup grand products: oracle vs the definition in Python ints, product vs oracle."""
import random

import numpy as np
import pytest

import parity_cases as pc
import zk_dcap_verifier_amd as z


def _inputs(orc, pyref, k, count, seed):
    n = 1 << k
    vals = [pc.rand_fr(orc, pyref, n, seed + i) for i in range(count)]
    sig = [pc.rand_fr(orc, pyref, n, seed + 50 + i) for i in range(count)]
    sc = pc.rand_fr(orc, pyref, 4, seed + 99)
    return vals, sig, sc[0], sc[1]


def test_oracle_permutation_product_matches_definition(orc, pyref):
    p, rnd = pyref, random.Random(3)
    k, count, bf, j0 = 4, 3, 5, 2
    n = 1 << k
    R = p.R
    v = [[rnd.randrange(R) for _ in range(n)] for _ in range(count)]
    s = [[rnd.randrange(R) for _ in range(n)] for _ in range(count)]
    s[1][3] = (-(v[1][3] + 5) * pow(7, -1, R)) % R      # with beta = 7, gamma = 5: a zero denominator -> batch_invert leaves 0
    beta, gamma, z0 = 7, 5, rnd.randrange(R)
    blind = [rnd.randrange(R) for _ in range(bf)]
    M = orc.fr_from_ints
    zz, last = orc.permutation_product([M(c) for c in v], [M(c) for c in s], k, M([beta])[0], M([gamma])[0], M([pow(p.DELTA, j0, R)])[0], M([z0])[0], M(blind))
    got = orc.fr_to_ints(zz)
    w = p.omega(k)
    want = [z0]
    for i in range(n - 1):
        num = den = 1
        for j in range(count):
            den = den * (v[j][i] + beta * s[j][i] + gamma) % R
            num = num * (v[j][i] + pow(p.DELTA, j0 + j, R) * pow(w, i, R) * beta + gamma) % R
        want.append(want[-1] * num * (pow(den, -1, R) if den else 0) % R)
    want[n - bf:] = blind
    assert got == want and orc.fr_to_ints(last)[0] == want[n - bf - 1]


def test_oracle_lookup_product_matches_definition(orc, pyref):
    p, rnd = pyref, random.Random(4)
    k, bf = 4, 5
    n, R = 1 << k, p.R
    cin, ctab, pin, ptab = ([rnd.randrange(R) for _ in range(n)] for _ in range(4))
    beta, gamma = rnd.randrange(R), rnd.randrange(R)
    blind = [rnd.randrange(R) for _ in range(bf)]
    M = orc.fr_from_ints
    got = orc.fr_to_ints(orc.lookup_product(M(cin), M(ctab), M(pin), M(ptab), k, M([beta])[0], M([gamma])[0], M(blind)))
    want = [1]
    for i in range(n - bf - 1):
        f = (cin[i] + beta) * (ctab[i] + gamma) * pow((pin[i] + beta) * (ptab[i] + gamma), -1, R) % R
        want.append(want[-1] * f % R)
    assert got == want + blind


def _check_backend(be, orc, pyref, k, count, seed, bf=5):
    n = 1 << k
    vals, sig, beta, gamma = _inputs(orc, pyref, k, count, seed)
    blind = pc.rand_fr(orc, pyref, bf, seed + 7)
    z0 = pc.rand_fr(orc, pyref, 1, seed + 8)[0]
    dstart = orc.fr_from_ints([pow(pyref.DELTA, 3, pyref.R)])[0]
    want, want_last = orc.permutation_product(vals, sig, k, beta, gamma, dstart, z0, blind)
    dv, ds = [be.to_device(c) for c in vals], [be.to_device(c) for c in sig]
    dz = be.alloc(n * 32)
    last = be.permutation_product_dev(dv, ds, k, beta, gamma, dstart, z0, blind, dz)
    assert (dz.download((n, 4)) == want).all() and (last == want_last).all()
    want_l = orc.lookup_product(vals[0], sig[0], vals[-1], sig[-1], k, beta, gamma, blind)
    be.lookup_product_dev(dv[0], ds[0], dv[-1], ds[-1], k, beta, gamma, blind, dz)
    assert (dz.download((n, 4)) == want_l).all()
    # every lookup of a proof in one call (k >= 11 takes the batched launch sequence, smaller domains the per-lookup one)
    quads = [(dv[0], ds[0], dv[-1], ds[-1]), (ds[0], dv[0], ds[-1], dv[-1]), (dv[-1], ds[-1], dv[0], ds[0])]
    hq = [(vals[0], sig[0], vals[-1], sig[-1]), (sig[0], vals[0], sig[-1], vals[-1]), (vals[-1], sig[-1], vals[0], sig[0])]
    blinds = [pc.rand_fr(orc, pyref, bf, seed + 40 + j) for j in range(3)]
    zs = z.permutation.lookup_commit_products(quads, k, beta, gamma, np.stack(blinds), backend=be)
    for zz, q, b in zip(zs, hq, blinds):
        assert (zz.download((n, 4)) == orc.lookup_product(q[0], q[1], q[2], q[3], k, beta, gamma, b)).all()
    for d in dv + ds + [dz] + zs:
        d.free()


@pytest.mark.parametrize("k,count", [(3, 1), (5, 3), (9, 2), (12, 4)])
def test_emulated_grand_products(emu, orc, pyref, k, count):
    _check_backend(emu, orc, pyref, k, count, seed=10 * k + count)


@pytest.mark.parametrize("k", [5, 12])     # 12: the all-sets-in-one-launch path (domains of at least one scan span), 5: set by set
def test_permutation_commit_chains_sets(emu, orc, pyref, k):
    """permutation_commit(): z of set s starts at the last unblinded value of set s-1, delta powers continue."""
    cs_degree, ncols, bf = 4, 5, 5
    n = 1 << k
    vals, sig, beta, gamma = _inputs(orc, pyref, k, ncols, 21)
    blind = [pc.rand_fr(orc, pyref, bf, 30 + s) for s in range(3)]
    dv, ds = [emu.to_device(c) for c in vals], [emu.to_device(c) for c in sig]
    zs = z.permutation.permutation_commit(dv, ds, k, cs_degree, beta, gamma, blind, backend=emu)
    assert len(zs) == 3
    last = orc.fr_from_ints([1])[0]
    for s, lo in enumerate(range(0, ncols, cs_degree - 2)):
        want, last = orc.permutation_product(vals[lo:lo + 2], sig[lo:lo + 2], k, beta, gamma,
                                             orc.fr_from_ints([pow(pyref.DELTA, lo, pyref.R)])[0], last, blind[s])
        assert (zs[s].download((n, 4)) == want).all(), s


@pytest.mark.gpu
@pytest.mark.parametrize("k,count", [(3, 1), (10, 3), (16, 3), (19, 2)])
def test_gpu_grand_products(gpu, orc, pyref, k, count):
    _check_backend(gpu, orc, pyref, k, count, seed=10 * k + count)
