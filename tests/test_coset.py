"""The units a multi-GPU prover shards the quotient by: the 2^(extended_k - k) cosets of the extended domain.
zk_coeff_to_coset_batch_dev must give exactly every 2^e-th entry of coeff_to_extended, zk_fr_interleave_dev must put them back."""
import numpy as np
import pytest

import parity_cases as pc


def _check(be, orc, pyref, k, e, seed):
    n, ek = 1 << k, k + e
    polys = [pc.rand_fr(orc, pyref, n, seed + i) for i in range(3)]
    d = [be.to_device(p) for p in polys]
    ext = [be.alloc((n << e) * 32) for _ in polys]
    be.coeff_to_extended_batch_dev(d, ext, k, ek)
    full = [x.download((n << e, 4)) for x in ext]
    cos = [[be.alloc(n * 32) for _ in polys] for _ in range(1 << e)]
    for j in range(1 << e):
        be.coeff_to_coset_batch_dev(d, cos[j], k, ek, j)
        for c, f in zip(cos[j], full):
            assert (c.download((n, 4)) == f[j::1 << e]).all(), (j,)
    out = be.alloc((n << e) * 32)
    be.fr_interleave_dev([cos[j][1] for j in range(1 << e)], n, out)
    assert (out.download((n << e, 4)) == full[1]).all()


@pytest.mark.parametrize("k,e", [(4, 1), (5, 2), (6, 3), (7, 2), (8, 1), (9, 3)])      # above the fixture's tile (2^6) a coset transform takes its pre-scaling from ONE table (ntt_coset_table)
def test_emulated_cosets(emu, orc, pyref, k, e):
    _check(emu, orc, pyref, k, e, seed=k)


def test_emulated_coset_table_on_and_off(emu, orc, pyref):
    """the pre-scaling table of a coset transform against the two-level powers it replaces: the same values, coset by coset"""
    k, e = 8, 2
    n = 1 << k
    d = [emu.to_device(pc.rand_fr(orc, pyref, n, 90 + i)) for i in range(2)]
    got = {}
    try:
        for mode in (1, 0):
            emu.tune(ntt_coset_table=mode)
            for j in range(1 << e):
                outs = [emu.alloc(n * 32) for _ in d]
                emu.coeff_to_coset_batch_dev(d, outs, k, k + e, j)
                got[mode, j] = [o.download((n, 4)) for o in outs]
    finally:
        emu.tune(ntt_coset_table=1)
    for j in range(1 << e):
        for a, b in zip(got[1, j], got[0, j]):
            assert (a == b).all(), j


@pytest.mark.gpu
@pytest.mark.parametrize("k,e", [(10, 2), (16, 2), (19, 2), (14, 3)])
def test_gpu_cosets(gpu, orc, pyref, k, e):
    _check(gpu, orc, pyref, k, e, seed=k)


def _check_pieces(be, orc, pyref, k, e, q, seed):
    """random pieces h_0 .. h_{q-1} -> the numerator h * (X^n - 1) on every coset (coset values of the pieces, combined with s_j^i and s_j - 1 in Python-derived scalars) ->
    zk_cosets_to_pieces_dev on cosets 0 .. q-1 must return the pieces; the extended route over ALL cosets (interleave, divide_by_vanishing_poly, extended_to_coeff) must too"""
    n, ek, R = 1 << k, k + e, pyref.R
    pieces = [pc.rand_fr(orc, pyref, n, seed + i) for i in range(q)]
    d = [be.to_device(p) for p in pieces]
    zn, won = pow(pyref.ZETA, n, R), pow(pyref.omega(ek), n, R)
    numer = []
    for j in range(1 << e):
        s = zn * pow(won, j, R) % R
        vals = [be.alloc(n * 32) for _ in range(q)]
        be.coeff_to_coset_batch_dev(d, vals, k, ek, j)
        sc = orc.ints_to_limbs([(s - 1) * pow(s, i, R) % R * (1 << 256) % R for i in range(q)])        # Montgomery form
        out = be.alloc(n * 32)
        be.fr_lincomb_dev(vals, sc, n, out)
        numer.append(out)
    ext = be.alloc((n << e) * 32)
    be.fr_interleave_dev(numer, n, ext)
    be.divide_by_vanishing_poly_dev(ext, k, ek)
    be.extended_to_coeff_dev(ext, k, ek)
    h = ext.download((n << e, 4))
    assert (h[: q * n] == np.concatenate(pieces)).all() and not h[q * n:].any()                          # the test's own construction, through the extended route
    got = [be.alloc(n * 32) for _ in range(q)]
    be.cosets_to_pieces_dev(numer[:q], k, ek, got)
    for g, p in zip(got, pieces):
        assert (g.download((n, 4)) == p).all(), (k, e, q)


@pytest.mark.parametrize("k,e,q", [(4, 1, 1), (4, 1, 2), (5, 2, 3), (5, 2, 4), (6, 3, 5), (5, 3, 7), (4, 3, 8), (5, 4, 8)])
def test_emulated_pieces_from_a_subset_of_cosets(emu, orc, pyref, k, e, q):
    _check_pieces(emu, orc, pyref, k, e, q, seed=7 * k + q)


@pytest.mark.gpu
@pytest.mark.parametrize("k,e,q", [(12, 2, 3), (19, 2, 3), (16, 3, 7), (18, 2, 4)])
def test_gpu_pieces_from_a_subset_of_cosets(gpu, orc, pyref, k, e, q):
    _check_pieces(gpu, orc, pyref, k, e, q, seed=k + q)


def test_coset_arguments_are_checked(emu, orc, pyref):
    import zk_dcap_verifier_amd as z
    d = emu.to_device(pc.rand_fr(orc, pyref, 16, 1))
    o = emu.alloc(16 * 32)
    with pytest.raises(z.ZkError):
        emu.coeff_to_coset_batch_dev([d], [o], 4, 6, 4)              # only cosets 0..3 exist
    with pytest.raises(z.ZkError):
        emu.coeff_to_coset_batch_dev([d], [o], 4, 3, 0)              # extended_k < k
    with pytest.raises(z.ZkError):
        emu.cosets_to_pieces_dev([d, o, d], 4, 5, [o, d, o])         # three pieces on two cosets
    with pytest.raises(z.ZkError):
        emu.cosets_to_pieces_dev([d], 4, 6, [d])                     # in place
