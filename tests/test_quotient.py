"""evaluate_h: (1) the oracle against the polynomial-identity DEFINITION (pure Python ints);
(2) the product's compiled micro-program (emulator on CPU) against the oracle."""
import random

import pytest

import parity_cases as pc
import quotient_cases as qc
from zk_dcap_verifier_amd import evaluation as ev


def test_oracle_evaluate_h_matches_polynomial_definition(orc, pyref):
    """Random low-degree polynomials; the numerator at extended row i must equal the Horner-in-y fold of
    the identities evaluated at the POINT x_i = ZETA * ext_omega^i, with rotations as x -> omega^rot x
    (SURVEY.md App. C.4 / the identities listed in halo2's evaluate_h)."""
    p, rnd = pyref, random.Random(12)
    k, deg = 3, 4
    dom = p.Domain(deg, k)
    n, size, ek = 1 << k, 1 << dom.extended_k, dom.extended_k
    R = p.R

    def rp():
        return [rnd.randrange(R) for _ in range(n)]              # a polynomial = n coefficients

    fixed, advice = [rp()], [rp(), rp()]
    l0, l_last, l_act = rp(), rp(), rp()
    sigma, zs = [rp(), rp(), rp()], [rp(), rp()]                     # 3 permutation columns, chunk 2 -> 2 sets
    lk_z, lk_a, lk_s = rp(), rp(), rp()
    beta, gamma, theta, y = (rnd.randrange(R) for _ in range(4))
    cst = rnd.randrange(R)
    blinding = 5
    perm_cols = [(0, 0), (0, 1), (1, 0)]
    # gate: previous*y + fixed0 * (advice0(X) * advice1(wX) - cst)
    g = ev.Graph()
    c0 = g.add_constant(orc.fr_from_ints([cst])[0])
    r0, r1 = g.add_rotation(0), g.add_rotation(1)
    m = g.add_calculation(ev.MUL, ev.vs(ev.ADVICE, 0, r0), ev.vs(ev.ADVICE, 1, r1))
    s = g.add_calculation(ev.SUB, m, c0)
    gate = g.add_calculation(ev.MUL, ev.vs(ev.FIXED, 0, r0), s)
    g.add_calculation(ev.HORNER, ev.vs(ev.PREVIOUS), [gate], ev.vs(ev.Y))
    # lookup: (advice0 * theta + advice1 + beta) * (fixed0 + gamma)
    lg = ev.Graph()
    q0 = lg.add_rotation(0)
    ci = lg.add_calculation(ev.HORNER, ev.vs(ev.ADVICE, 0, q0), [ev.vs(ev.ADVICE, 1, q0)], ev.vs(ev.THETA))
    a1 = lg.add_calculation(ev.ADD, ci, ev.vs(ev.BETA))
    b1 = lg.add_calculation(ev.ADD, ev.vs(ev.FIXED, 0, q0), ev.vs(ev.GAMMA))
    lg.add_calculation(ev.MUL, a1, b1)
    prog = ev.Program(k=k, extended_k=ek, n_fixed=1, n_advice=2, n_instance=0, n_challenges=0, blinding_factors=blinding, cs_degree=deg,
                      perm_columns=perm_cols, custom_gates=g, lookups=[lg])
    odom = orc.Domain(deg, k)

    def cos(poly):
        return odom.coeff_to_extended(orc.fr_from_ints(poly))

    def M(v):
        return orc.fr_from_ints([v])[0]

    got = orc.fr_to_ints(orc.evaluate_h(prog.to_blob(), [cos(f) for f in fixed], [cos(a) for a in advice], [], cos(l0), cos(l_last), cos(l_act),
                                        [cos(s_) for s_ in sigma], [cos(z_) for z_ in zs], [cos(lk_z)], [cos(lk_a)], [cos(lk_s)], [],
                                        M(beta), M(gamma), M(theta), M(y), size))
    w = dom.omega
    E = p.poly_eval
    col = {(0, 0): advice[0], (0, 1): advice[1], (1, 0): fixed[0]}
    for i in range(size):
        x = dom.extended_point(i)
        xn, xp, xl = x * w % R, x * pow(w, -1, R) % R, x * pow(w, -(blinding + 1), R) % R

        def fold(v_, t):
            return (v_ * y + t) % R

        v = 0
        v = fold(v, E(fixed[0], x) * (E(advice[0], x) * E(advice[1], xn) - cst))
        v = fold(v, (1 - E(zs[0], x)) * E(l0, x))
        v = fold(v, (E(zs[1], x) ** 2 - E(zs[1], x)) * E(l_last, x))
        v = fold(v, (E(zs[1], x) - E(zs[0], xl)) * E(l0, x))
        dj = 0
        for s_idx, chunk in enumerate(([0, 1], [2])):
            left, right = E(zs[s_idx], xn), E(zs[s_idx], x)
            for j in chunk:
                vv = E(col[perm_cols[j]], x)
                left = left * (vv + beta * E(sigma[j], x) + gamma) % R
                right = right * (vv + pow(p.DELTA, dj, R) * beta * x + gamma) % R
                dj += 1
            v = fold(v, (left - right) * E(l_act, x))
        tv = ((E(advice[0], x) * theta + E(advice[1], x)) + beta) * (E(fixed[0], x) + gamma) % R
        a_, s_, z_ = E(lk_a, x), E(lk_s, x), E(lk_z, x)
        v = fold(v, (1 - z_) * E(l0, x))
        v = fold(v, (z_ * z_ - z_) * E(l_last, x))
        v = fold(v, (E(lk_z, xn) * (a_ + beta) * (s_ + gamma) - z_ * tv) * E(l_act, x))
        v = fold(v, (a_ - s_) * E(l0, x))
        v = fold(v, (a_ - s_) * (a_ - E(lk_a, xp)) * E(l_act, x))
        assert got[i] == v % R, i


SHAPES = [(1, dict(k=3, cs_degree=4, n_fixed=2, n_advice=3, n_instance=1, n_challenges=2, n_perm=5, n_lookups=2)),
          (2, dict(k=4, cs_degree=5, n_fixed=1, n_advice=2, n_instance=0, n_challenges=0, n_perm=3, n_lookups=1)),
          (3, dict(k=3, cs_degree=3, n_fixed=1, n_advice=1, n_instance=0, n_challenges=1, n_perm=0, n_lookups=0)),
          (4, dict(k=2, cs_degree=9, n_fixed=3, n_advice=4, n_instance=2, n_challenges=1, n_perm=8, n_lookups=3))]


@pytest.mark.parametrize("seed,shape", SHAPES)
def test_emulated_quotient_vs_oracle(emu, orc, pyref, seed, shape):
    prog = qc.build_program(orc, pyref, seed=seed, **shape)
    qc.run_case(emu, orc, pyref, pc, prog, seed=seed)


@pytest.mark.parametrize("seed", [1, 2, 3, 5, 8])
def test_emulated_quotient_dense_random_programs(emu, orc, pyref, seed):
    """60 random calculations per gate graph: values with several readers (a product that is folded AND read elsewhere must not be fused away), gate
    polynomials that read PreviousValue themselves (the custom-gate Horner may then not run in the accumulator), Horner chains shared between lookups"""
    prog = qc.build_program(orc, pyref, seed=seed, gate_ops=60, k=4, cs_degree=5, n_fixed=4, n_advice=6, n_instance=1, n_challenges=1, n_perm=7, n_lookups=3)
    qc.run_case(emu, orc, pyref, pc, prog, seed=seed)


def test_quotient_rejects_malformed_programs(emu, orc, pyref):
    import zk_dcap_verifier_amd as z
    with pytest.raises(z.ZkError):
        emu.quotient_program_load(b"\x00" * 64)
    prog = qc.build_program(orc, pyref, k=3, cs_degree=4, n_fixed=1, n_advice=1, n_instance=0, n_challenges=0, n_perm=2, n_lookups=1, seed=9)
    blob = prog.to_blob()
    with pytest.raises(z.ZkError):
        emu.quotient_program_load(blob[: (len(blob) // 2) & ~3])         # truncated
    prog.custom_gates.calculations[0] = (ev.MUL, 0, (ev.vs(ev.ADVICE, 99, 0), ev.vs(ev.BETA)))
    with pytest.raises(z.ZkError):
        emu.quotient_program_load(prog.to_blob())                        # column index out of range


@pytest.mark.gpu
@pytest.mark.parametrize("seed,shape", SHAPES + [(5, dict(k=10, cs_degree=5, n_fixed=4, n_advice=6, n_instance=1, n_challenges=1, n_perm=7, n_lookups=3))])
def test_gpu_quotient_vs_oracle(gpu, orc, pyref, seed, shape):
    prog = qc.build_program(orc, pyref, seed=seed, gate_ops=60 if shape["k"] >= 10 else 24, **shape)
    qc.run_case(gpu, orc, pyref, pc, prog, seed=seed)


def _pk_level_case(be, orc, pyref, seed, shape, form):
    """zk_pk_load + zk_evaluate_h (coefficient polynomials in, h(X) out) against the oracle's composition
    coeff_to_extended -> evaluate_h -> divide_by_vanishing_poly -> extended_to_coeff."""
    import numpy as np
    prog = qc.build_program(orc, pyref, seed=seed, **shape)
    k, deg = shape["k"], shape["cs_degree"]
    dom = orc.Domain(deg, k)
    n, en = 1 << k, 1 << dom.extended_k
    chunk = deg - 2
    n_sets = (len(prog.perm_columns) + chunk - 1) // chunk if prog.perm_columns else 0
    nl = len(prog.lookups)
    P = lambda cnt, sd: [pc.rand_fr(orc, pyref, n, sd + 3 * i) for i in range(cnt)]
    fixed, sigma, ls = P(prog.n_fixed, seed), P(len(prog.perm_columns), seed + 100), P(3, seed + 200)
    advice, inst, zs = P(prog.n_advice, seed + 300), P(prog.n_instance, seed + 400), P(n_sets, seed + 500)
    lz, la, lt = P(nl, seed + 600), P(nl, seed + 700), P(nl, seed + 800)
    chal = pc.rand_fr(orc, pyref, max(prog.n_challenges, 1), seed + 900)[: prog.n_challenges]
    beta, gamma, theta, y = pc.rand_fr(orc, pyref, 4, seed + 901)
    ext = lambda cols: [dom.coeff_to_extended(c) for c in cols]
    num = orc.evaluate_h(prog.to_blob(), ext(fixed), ext(advice), ext(inst), *ext(ls), ext(sigma), ext(zs), ext(lz), ext(la), ext(lt),
                         chal, beta, gamma, theta, y, en)
    want_h = dom.extended_to_coeff(dom.divide_by_vanishing_poly(num))
    h = be.quotient_program_load(prog.to_blob())
    if form == 0:
        pk = be.pk_load(h, fixed, sigma, ls[0], ls[1], ls[2], form=0)
    else:
        pk = be.pk_load(h, ext(fixed), ext(sigma), *ext(ls), form=1)
    kw = dict(advice=advice, instance=inst, perm_products=zs, lookup_product=lz, lookup_input=la, lookup_table=lt, challenges=chal,
              beta=beta, gamma=gamma, theta=theta, y=y)
    got_num = be.evaluate_h(pk, out_rows=en, finish=False, **kw)
    assert (got_num == num).all()
    got_h = be.evaluate_h(pk, out_rows=n * (deg - 1), finish=True, **kw)
    assert (got_h == want_h).all()
    be.pk_release(pk)
    be.quotient_program_release(h)


@pytest.mark.parametrize("form", [0, 1])
def test_emulated_pk_level_evaluate_h(emu, orc, pyref, form):
    _pk_level_case(emu, orc, pyref, 1, SHAPES[0][1], form)


@pytest.mark.gpu
@pytest.mark.parametrize("form,idx", [(0, 0), (1, 1), (0, 4)])
def test_gpu_pk_level_evaluate_h(gpu, orc, pyref, form, idx):
    shapes = SHAPES + [(5, dict(k=10, cs_degree=5, n_fixed=4, n_advice=6, n_instance=1, n_challenges=1, n_perm=7, n_lookups=3))]
    _pk_level_case(gpu, orc, pyref, shapes[idx][0], shapes[idx][1], form)


def _theta_compression_case(be, orc, pyref, k, seed):
    """compress_expressions of a lookup over Lagrange columns (rotations wrap mod n): product vs Python definition."""
    import random
    p, rnd = pyref, random.Random(seed)
    n, R = 1 << k, p.R
    adv = [[rnd.randrange(R) for _ in range(n)] for _ in range(2)]
    fix = [[rnd.randrange(R) for _ in range(n)]]
    theta = rnd.randrange(R)
    g = ev.Graph()
    r0, r1 = g.add_rotation(0), g.add_rotation(-1)
    e0 = g.add_calculation(ev.MUL, ev.vs(ev.FIXED, 0, r0), ev.vs(ev.ADVICE, 0, r0))        # q * a
    e1 = g.add_calculation(ev.MUL, ev.vs(ev.FIXED, 0, r0), ev.vs(ev.ADVICE, 1, r1))        # q * b(omega^-1 X)
    g.add_calculation(ev.HORNER, e0, [e1, ev.vs(ev.ADVICE, 1, r0)], ev.vs(ev.THETA))        # ((e0*theta + e1)*theta + b)
    prog = ev.expression_program(k, 1, 2, 0, 0, g)
    M = orc.fr_from_ints
    d_adv, d_fix = [be.to_device(M(c)) for c in adv], [be.to_device(M(c)) for c in fix]
    out = be.alloc(n * 32)
    e = ev.Evaluator(prog, backend=be)
    one = M([1])[0]
    e.evaluate_h(fixed=d_fix, advice=d_adv, instance=[], l0=d_fix[0], l_last=d_fix[0], l_active_row=d_fix[0], perm_cosets=[], perm_products=[],
                 lookup_product=[], lookup_input=[], lookup_table=[], challenges=[], beta=one, gamma=one, theta=M([theta])[0], y=one, out=out)
    got = orc.fr_to_ints(out.download((n, 4)))
    want = [((fix[0][i] * adv[0][i] * theta + fix[0][i] * adv[1][(i - 1) % n]) * theta + adv[1][i]) % R for i in range(n)]
    assert got == want
    e.release()


def test_emulated_theta_compression(emu, orc, pyref):
    _theta_compression_case(emu, orc, pyref, 5, 3)


@pytest.mark.gpu
def test_gpu_theta_compression(gpu, orc, pyref):
    _theta_compression_case(gpu, orc, pyref, 12, 4)


def _factored_compression_case(be, orc, pyref, k, seed):
    """the compiler's selector factoring (quotient.hip factor_common_horner): Horner(0, [q*a_0 .. q*a_3], theta) = q * Horner(0, [a_j], theta), also with a first part that IS the
    factor (q*a, q*a*2, q*a*3: the factor is the product q*a), and NOT applied when one part lacks the factor.  Same values with the rewrite on and off; fewer products with it."""
    import random
    p, rnd = pyref, random.Random(seed)
    n, R = 1 << k, p.R
    adv = [[rnd.randrange(R) for _ in range(n)] for _ in range(4)]
    fix = [[rnd.randrange(R) for _ in range(n)]]
    theta = rnd.randrange(R)
    M = orc.fr_from_ints
    q = lambda g, r: ev.vs(ev.FIXED, 0, r)

    def shape(which):
        g = ev.Graph()
        zero, r0, r1 = g.add_constant(M([0])[0]), g.add_rotation(0), g.add_rotation(1)
        if which == "products":
            parts = [g.add_calculation(ev.MUL, q(g, r0), ev.vs(ev.ADVICE, j, r1 if j == 2 else r0)) for j in range(4)]
            want = lambda i: fix[0][i] * sum(adv[j][(i + 1) % n if j == 2 else i] * pow(theta, 3 - j, R) for j in range(4)) % R
        elif which == "bare":
            base = g.add_calculation(ev.MUL, q(g, r0), ev.vs(ev.ADVICE, 0, r0))
            cs_ = [g.add_constant(M([c])[0]) for c in (2, 3, 4)]
            parts = [base] + [g.add_calculation(ev.MUL, base, c) for c in cs_]
            want = lambda i: fix[0][i] * adv[0][i] * (pow(theta, 3, R) + 2 * theta * theta + 3 * theta + 4) % R
        else:                                                            # "mixed": the last part has no factor — the chain stays as halo2 wrote it
            parts = [g.add_calculation(ev.MUL, q(g, r0), ev.vs(ev.ADVICE, j, r0)) for j in range(3)] + [ev.vs(ev.ADVICE, 3, r0)]
            want = lambda i: (fix[0][i] * sum(adv[j][i] * pow(theta, 3 - j, R) for j in range(3)) + adv[3][i]) % R
        g.add_calculation(ev.HORNER, zero, parts, ev.vs(ev.THETA))
        return ev.expression_program(k, 1, 4, 0, 0, g), want
    d_adv, d_fix = [be.to_device(M(c)) for c in adv], [be.to_device(M(c)) for c in fix]
    one = M([1])[0]
    muls = {}
    for which in ("products", "bare", "mixed"):
        prog, want = shape(which)
        for on in (0, 1):
            be.tune(quot_factor_horner=on)
            e = ev.Evaluator(prog, backend=be)
            mix = be.quotient_program_opmix(e.handle)
            muls[which, on] = mix["mul"] + mix["muladd"]
            out = be.alloc(n * 32)
            e.evaluate_h(fixed=d_fix, advice=d_adv, instance=[], l0=d_fix[0], l_last=d_fix[0], l_active_row=d_fix[0], perm_cosets=[], perm_products=[],
                         lookup_product=[], lookup_input=[], lookup_table=[], challenges=[], beta=one, gamma=one, theta=M([theta])[0], y=one, out=out)
            assert orc.fr_to_ints(out.download((n, 4))) == [want(i) for i in range(n)], (which, on)
            out.free()
            e.release()
    be.tune(quot_factor_horner=1)
    assert muls["products", 0] == 7 and muls["products", 1] == 4           # 4 products + 3 Horner steps  ->  3 Horner steps + 1 product
    assert muls["bare", 0] == 7 and muls["bare", 1] == 5                   # q*a, 3 scalings, 3 steps  ->  q*a, 3 steps over constants, 1 product
    assert muls["mixed", 0] == muls["mixed", 1] == 6


def test_emulated_factored_theta_compression(emu, orc, pyref):
    _factored_compression_case(emu, orc, pyref, 5, 11)


@pytest.mark.gpu
def test_gpu_factored_theta_compression(gpu, orc, pyref):
    _factored_compression_case(gpu, orc, pyref, 12, 12)


def _bench_program_case(be, orc, pyref, k):
    """The sgx-shaped program that bench.py times (24 degree-3 gates, 16 permutation columns in 6 sets, 11 lookups), at small k."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    import zk_dcap_verifier_amd as z
    prog = bench.sgx_shaped_program(z, k, k + 2, 25, 18, 11, 16, 5)
    qc.run_case(be, orc, pyref, pc, prog, seed=77)


def test_emulated_bench_program(emu, orc, pyref):
    _bench_program_case(emu, orc, pyref, 3)


@pytest.mark.gpu
def test_gpu_bench_program(gpu, orc, pyref):
    _bench_program_case(gpu, orc, pyref, 9)



DENSE = dict(k=4, cs_degree=5, n_fixed=4, n_advice=6, n_instance=1, n_challenges=1, n_perm=7, n_lookups=3)


def test_emulated_quotient_dense_program(emu, orc, pyref):
    """60 gate operations over 6 advice columns, 7 permutation columns, 3 lookups: many live intermediates (LDS slots beyond the register slot)"""
    qc.run_case(emu, orc, pyref, pc, qc.build_program(orc, pyref, seed=5, gate_ops=60, **DENSE), seed=5)


@pytest.mark.gpu
def test_gpu_quotient_dense_program(gpu, orc, pyref):
    qc.run_case(gpu, orc, pyref, pc, qc.build_program(orc, pyref, seed=5, gate_ops=60, **dict(DENSE, k=10)), seed=5)
    qc.run_case(gpu, orc, pyref, pc, qc.build_program(orc, pyref, seed=4, **SHAPES[3][1]), seed=4)
