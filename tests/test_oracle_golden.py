"""Pins the oracle: constants vs SURVEY.md App. A, KATs, Python big-int definitions, and the only
proof bytes the reference ships (bin/assets/proof.bin, used by bin/src/main.rs:269-279)."""
import os
import random

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_constants_match_survey_appendix_a(pyref):
    p = pyref
    assert p.P == 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
    assert p.R == 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    assert p.mont_r(p.P) == 0x0E0A77C19A07DF2F666EA36F7879462C0A78EB28F5C70B3DD35D438DC58F0D9D
    assert p.mont_r(p.R) == 0x0E0A77C19A07DF2F666EA36F7879462E36FC76959F60CD29AC96341C4FFFFFFB
    assert p.mont_inv64(p.P) == 0x87D20782E4866389 and p.mont_inv64(p.R) == 0xC2E1F593EFFFFFFF
    assert p.ROOT_OF_UNITY == 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
    assert pow(p.ROOT_OF_UNITY, 1 << 28, p.R) == 1 and pow(p.ROOT_OF_UNITY, 1 << 27, p.R) != 1
    assert p.DELTA == 0x09226B6E22C6F0CA64EC26AAD4C86E715B5F898E5E963F25870E56BBE533E9A2
    assert p.ZETA == 0x0000000000000000B3C4D79D41A917585BFC41088D8DAAA78B17EA66B99C90DD
    assert pow(p.ZETA, 3, p.R) == 1 and p.ZETA != 1
    assert p.omega(19) == 0x0CF1526AAAFAC6BACBB67D11A4077806B123F767E4B0883D14CC0193568FC082
    assert p.omega(21) == 0x032750F8F3C2493D0828C7285D0258E1BDCAA463F4442A52747B5C96639659BB
    assert p.omega(22) == 0x18C95F1AE6514E11A1B30FD7923947C5FFCEC5347F16E91B4DD654168326BEDE
    assert p.g1_mul(p.G1_GEN, p.R) is None and p.g1_mul(p.G1_GEN, p.R - 1) == (1, p.P - 2)


def test_field_ops_vs_bigint(orc, pyref):
    rnd = random.Random(1)
    for mod, mul, add, sub, frm, to in ((pyref.R, orc.fr_mul, orc.fr_add, orc.fr_sub, orc.fr_from_ints, orc.fr_to_ints),
                                        (pyref.P, orc.fq_mul, orc.fq_add, orc.fq_sub, orc.fq_from_ints, orc.fq_to_ints)):
        a = [rnd.randrange(mod) for _ in range(200)] + [0, 1, mod - 1, mod - 1]
        b = [rnd.randrange(mod) for _ in range(200)] + [mod - 1, mod - 1, mod - 1, 1]
        A, B = frm(a), frm(b)
        assert to(mul(A, B)) == [x * y % mod for x, y in zip(a, b)]
        assert to(add(A, B)) == [(x + y) % mod for x, y in zip(a, b)]
        assert to(sub(A, B)) == [(x - y) % mod for x, y in zip(a, b)]
    assert orc.limbs_to_ints(orc.fr_from_ints([1]))[0] == pyref.mont_r(pyref.R)   # transmute(Fr::one()) == R
    assert orc.g1_affine_to_ints(orc.g1_generator()) == [(1, 2)]


def test_msm_kats(orc, pyref):
    p = pyref
    G30 = (0x036083BFA420B15A4C11F66A3CFFD55318B019FEB45F833A876E93848625F5AE,
           0x2630C348C019C3EDB74FE62A7E921361AAE9621988223514D56CA8B36ADC9E36)
    bases = orc.g1_affine_from_ints([p.g1_mul(p.G1_GEN, k) for k in (1, 2, 3, 4)])
    for th in (1, 3, 8):
        res = orc.best_multiexp(orc.fr_from_ints([1, 2, 3, 4]), bases, threads=th)
        assert orc.g1_affine_to_ints(orc.g1_to_affine(res)) == [G30]
    bases = orc.gen_bases_arith(5, 3, 16, threads=3)
    assert orc.g1_affine_to_ints(bases) == [p.g1_mul(p.G1_GEN, 5 + 3 * i) for i in range(16)]
    res = orc.best_multiexp(orc.fr_from_ints([i * i + 7 for i in range(16)]), bases, threads=2)
    assert orc.g1_affine_to_ints(orc.g1_to_affine(res)) == [p.g1_mul(p.G1_GEN, 52480)]


def test_msm_random_vs_definition(orc, pyref):
    p, rnd = pyref, random.Random(3)
    n = 64
    ks = [rnd.randrange(p.R) for _ in range(n)]
    ks[3], ks[7], ks[9] = 0, 1, p.R - 1
    dl = [rnd.randrange(1, p.R) for _ in range(n)]
    pts = [p.g1_mul(p.G1_GEN, d) for d in dl]
    pts[5] = None
    pts[11] = pts[10]
    want = p.msm_naive(ks, pts)
    for th in (1, 8):
        res = orc.best_multiexp(orc.fr_from_ints(ks), orc.g1_affine_from_ints(pts), threads=th)
        assert orc.g1_affine_to_ints(orc.g1_to_affine(res)) == [want]


def test_ntt_kats_and_definition(orc, pyref):
    p, rnd = pyref, random.Random(4)
    out = p.ntt_definition([1, 2, 3, 4], p.omega(2))
    assert out == [0xA, 0x00000000000000016789AF3A83522EB1969386A2F88C094A419FE246C11F9394, p.R - 2,
                   0x30644E72E131A02850C6967BFE2F29AB91A061A5812D67470242134D2EE06C69]
    out8 = p.ntt_definition(list(range(1, 9)), p.omega(3))
    assert out8[0] == 0x24 and out8[4] == p.R - 4
    assert out8[1] == 0x002701A4FD3F1D3E7A309CDC72C7C8FCB5C94AF009CB48E6E51461367A2F1796
    for logn in (1, 2, 3, 6, 10):
        n = 1 << logn
        a = [rnd.randrange(p.R) for _ in range(n)]
        w = p.omega(logn)
        want = p.ntt_definition(a, w) if logn <= 6 else p.ntt_fast(a, w)
        for th in (1, 4, 8):
            got = orc.fr_to_ints(orc.best_fft(orc.fr_from_ints(a), orc.fr_from_ints([w])[0], logn, threads=th))
            assert got == want


def test_domain_vs_definition(orc, pyref):
    p, rnd = pyref, random.Random(5)
    for j, k in ((4, 5), (5, 6), (3, 4), (9, 4), (2, 3)):
        d, pd = orc.Domain(j, k), p.Domain(j, k)
        assert d.extended_k == pd.extended_k
        a = [rnd.randrange(p.R) for _ in range(1 << k)]
        A = orc.fr_from_ints(a)
        assert orc.fr_to_ints(d.lagrange_to_coeff(A)) == pd.lagrange_to_coeff(a)
        ext, pext = d.coeff_to_extended(A), pd.coeff_to_extended(a)
        assert orc.fr_to_ints(ext) == pext
        for i in (0, 1, 3, (1 << pd.extended_k) - 1):           # definition: value of the polynomial on the coset
            assert pext[i] == p.poly_eval(a, pd.extended_point(i))
        assert orc.fr_to_ints(d.divide_by_vanishing_poly(ext)) == pd.divide_by_vanishing_poly(pext)
        xn = pd.extended_point(2)                                  # t_evaluations really is 1/(X^n - 1)
        assert pd.t_evaluations[2 % len(pd.t_evaluations)] == pow(pow(xn, 1 << k, p.R) - 1, -1, p.R)
        back = orc.fr_to_ints(d.extended_to_coeff(ext))
        assert back[: 1 << k] == a and all(v == 0 for v in back[1 << k:])


def test_reference_proof_bin_decodes_on_this_curve(orc, pyref):
    """bin/assets/proof.bin (bin/src/main.rs:275): 47 x 32-byte LE words = 13 G1 + 32 Fr + 2 G1
    (SURVEY.md App. B).  Every point word must be an x-coordinate on y^2 = x^3 + 3 over OUR Fq and
    every scalar word < OUR r — this is the one piece of reference-held data that touches the
    path's field/curve definitions."""
    raw = open(os.path.join(HERE, "golden", "proof.bin")).read().strip()
    assert raw.startswith("0x")
    data = bytes.fromhex(raw[2:])
    assert len(data) == 1504
    words = [int.from_bytes(data[i:i + 32], "little") for i in range(0, 1504, 32)]
    point_words = words[:13] + words[45:]
    for w in point_words:
        x = w & ((1 << 254) - 1)                    # bit 254 carries the y-sign in halo2curves-axiom
        ys = pyref.g1_decompress_x(x)
        assert x < pyref.P and ys is not None
        pt = orc.g1_affine_from_ints([(x, ys[0])])[0]
        assert orc.g1_is_on_curve(pt)
    assert all(w < pyref.R for w in words[13:45])


def test_committed_kats_fixture(orc, pyref):
    """tests/golden/kats.json (tools/gen_golden.py): the committed vectors against the C oracle."""
    import json
    k = json.load(open(os.path.join(HERE, "golden", "kats.json")))
    assert int(k["fq_modulus"], 16) == pyref.P and int(k["fr_modulus"], 16) == pyref.R
    assert orc.limbs_to_ints(orc.fr_from_ints([1]))[0] == int(k["fr_R"], 16)
    assert orc.limbs_to_ints(orc.fq_from_ints([1]))[0] == int(k["fq_R"], 16)
    g = orc.g1_generator()
    for case in k["msm"]:
        bases = np.concatenate([orc.g1_to_affine(orc.g1_mul(g, orc.fr_from_ints([d])[0])) for d in case["base_dlogs"]])
        res = orc.g1_to_affine(orc.best_multiexp(orc.fr_from_ints(case["scalars"]), bases))
        want = [int(c, 16) for c in k["g1_multiples"][str(case["result_dlog"])]]
        assert orc.g1_affine_to_ints(res) == [tuple(want)]
    for case in k["ntt"]:
        w = orc.fr_from_ints([int(k["omega"][str(case["log_n"])], 16)])[0]
        got = orc.fr_to_ints(orc.best_fft(orc.fr_from_ints(case["input"]), w, case["log_n"]))
        assert got == [int(v, 16) for v in case["output"]]
