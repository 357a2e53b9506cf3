"""End-to-end algebra check on a satisfiable toy circuit (custom gate + copy constraints + one lookup).

Everything the prover computes between the witness and h(X) runs through the product API — grand products,
lagrange_to_coeff, pk upload, evaluate_h with the vanishing division — and the result must satisfy what halo2's
verifier ultimately checks:  N(x) = h(x) * (x^n - 1)  at a random point x, where N is the Horner-in-y combination
of all gate / permutation / lookup identities.  This holds only if every identity really vanishes on the domain,
i.e. if the rotations, l_0 / l_last / l_active_row handling, delta powers and product columns of the kernels are
mutually consistent with a VALID witness (an oracle comparison alone cannot show that).
"""
import random

import numpy as np
import pytest

import parity_cases as pc
import zk_dcap_verifier_amd as z
from zk_dcap_verifier_amd import evaluation as ev


def _build_and_check(be, orc, pyref, k, seed, tamper=False):
    p, rnd = pyref, random.Random(seed)
    R = p.R
    n, bf, deg = 1 << k, 5, 4
    u = n - bf - 1                      # usable rows [0, u); row u is l_last; rows (u, n) are blinding
    omega = p.omega(k)
    M = orc.fr_from_ints

    # ---- witness: c = a * b on rows where q = 1; copy constraints a[i] == c[i-1] on a chain; lookup a in table
    table = list(range(1, 9)) + [0] * (n - 8)                  # fixed lookup table column (values 1..8, rest 0)
    a, b, c, q = [0] * n, [0] * n, [0] * n, [0] * n
    a[0] = rnd.randrange(1, 9)
    for i in range(u):
        if i > 0:
            a[i] = c[i - 1] if (i % 3 != 0) else rnd.randrange(1, 9)
        if a[i] not in table[:8]:                               # keep a inside the table: restart the chain
            a[i] = rnd.randrange(1, 9)
        b[i] = 1 if i % 2 else rnd.randrange(1, 3)
        c[i] = a[i] * b[i] % R
        q[i] = 1
    copies = [((2, i - 1), (0, i)) for i in range(1, u) if a[i] == c[i - 1] and i % 3 != 0]      # (col c,row) ~ (col a,row)
    if tamper:
        c[1] = (c[1] + 1) % R                                     # break the gate on one row (negative control)
        copies = [cp for cp in copies if cp[0] != (2, 1)]
    for col in (a, b, c):                                        # blinding rows: arbitrary
        for i in range(u, n):
            col[i] = rnd.randrange(R)

    # ---- permutation sigma over columns [a, b, c] (all advice), usable rows
    cols = [a, b, c]
    ncol = len(cols)
    mapping = {(j, i): (j, i) for j in range(ncol) for i in range(n)}
    for x, y in copies:                                          # merge cycles: swap images
        mapping[x], mapping[y] = mapping[y], mapping[x]
    sigma = [[pow(p.DELTA, mapping[(j, i)][0], R) * pow(omega, mapping[(j, i)][1], R) % R for i in range(n)] for j in range(ncol)]
    for (x, y) in copies:
        assert cols[x[0]][x[1]] == cols[y[0]][y[1]]

    # ---- lookup: input expression = a, table expression = table  (degree 2 + 1 + 1 = cs_degree 4; an input q*a would need degree 5)
    beta, gamma, theta, y_ = (rnd.randrange(1, R) for _ in range(4))
    inp = list(a)
    # permuted input / table on usable rows (halo2 permute_expression_pair): sort inputs, align table
    pin = sorted(inp[:u])
    tab_left = sorted(table[:u])
    ptab = [None] * u
    for i in range(u):
        if i == 0 or pin[i] != pin[i - 1]:
            ptab[i] = pin[i]
            tab_left.remove(pin[i])
    rest = iter(tab_left)
    for i in range(u):
        if ptab[i] is None:
            ptab[i] = next(rest)
    pin += [rnd.randrange(R) for _ in range(n - u)]
    ptab += [rnd.randrange(R) for _ in range(n - u)]

    # ---- device: grand products through the product API
    dA = [be.to_device(M(col)) for col in cols]
    dS = [be.to_device(M(s)) for s in sigma]
    blind = [pc.rand_fr(orc, pyref, bf, seed + 10 + s) for s in range(3)]
    zs = z.permutation.permutation_commit(dA, dS, k, deg, M([beta])[0], M([gamma])[0], blind, backend=be)
    assert len(zs) == 2                                          # chunk = deg - 2 = 2 columns per set
    d_in, d_tab, d_pin, d_ptab = (be.to_device(M(v)) for v in (inp, table, pin, ptab))
    zl = z.permutation.lookup_commit_product(d_in, d_tab, d_pin, d_ptab, k, M([beta])[0], M([gamma])[0], blind[2], backend=be)
    z_host = [zz.download((n, 4)) for zz in zs]
    assert orc.fr_to_ints(z_host[-1][u:u + 1])[0] == 1          # the permutation really closes: z_last(omega^u) = 1
    assert orc.fr_to_ints(zl.download((n, 4))[u:u + 1])[0] == 1

    # ---- Lagrange -> coefficient form for everything (GPU), then the pk and the proof polynomials as host arrays
    l0 = [1] + [0] * (n - 1)
    l_last = [0] * n
    l_last[u] = 1
    l_act = [1 if i < u else 0 for i in range(n)]
    lag = {"q": M(q), "table": M(table), "l0": M(l0), "l_last": M(l_last), "l_act": M(l_act),
           "s0": M(sigma[0]), "s1": M(sigma[1]), "s2": M(sigma[2]), "a": M(a), "b": M(b), "c": M(c),
           "z0": z_host[0], "z1": z_host[1], "zl": zl.download((n, 4)), "pin": M(pin), "ptab": M(ptab)}
    names = list(lag)
    dcols = [be.to_device(lag[nm]) for nm in names]
    be.lagrange_to_coeff_batch_dev(dcols, k)
    coef = {nm: d.download((n, 4)) for nm, d in zip(names, dcols)}

    # ---- program: gate q*(a*b - c); lookup input a, table `table`
    g = ev.Graph()
    r0 = g.add_rotation(0)
    t = g.add_calculation(ev.MUL, ev.vs(ev.ADVICE, 0, r0), ev.vs(ev.ADVICE, 1, r0))
    t = g.add_calculation(ev.SUB, t, ev.vs(ev.ADVICE, 2, r0))
    gate = g.add_calculation(ev.MUL, ev.vs(ev.FIXED, 0, r0), t)
    g.add_calculation(ev.HORNER, ev.vs(ev.PREVIOUS), [gate], ev.vs(ev.Y))
    lg = ev.Graph()
    q0 = lg.add_rotation(0)
    a1 = lg.add_calculation(ev.ADD, ev.vs(ev.ADVICE, 0, q0), ev.vs(ev.BETA))
    b1 = lg.add_calculation(ev.ADD, ev.vs(ev.FIXED, 1, q0), ev.vs(ev.GAMMA))
    lg.add_calculation(ev.MUL, a1, b1)
    dom = p.Domain(deg, k)
    prog = ev.Program(k=k, extended_k=dom.extended_k, n_fixed=2, n_advice=3, n_instance=0, n_challenges=0, blinding_factors=bf, cs_degree=deg,
                      perm_columns=[(0, 0), (0, 1), (0, 2)], custom_gates=g, lookups=[lg])
    e = ev.Evaluator(prog, backend=be)
    e.load_pk([coef["q"], coef["table"]], [coef["s0"], coef["s1"], coef["s2"]], coef["l0"], coef["l_last"], coef["l_act"])
    kw = dict(advice=[coef["a"], coef["b"], coef["c"]], instance=[], perm_products=[coef["z0"], coef["z1"]], lookup_product=[coef["zl"]],
              lookup_input=[coef["pin"]], lookup_table=[coef["ptab"]], challenges=[], beta=M([beta])[0], gamma=M([gamma])[0],
              theta=M([theta])[0], y=M([y_])[0])
    h = e.evaluate_h_polys(finish=True, **kw)                         # (deg-1)*n coefficients of h(X)
    num_ext = e.evaluate_h_polys(finish=False, **kw)                  # numerator on the extended coset
    # numerator in coefficient form (all 2^ek coefficients)
    dnum = be.to_device(num_ext)
    be.extended_to_coeff_dev(dnum, k, dom.extended_k)
    num_coef = dnum.download((1 << dom.extended_k, 4))
    # ---- the verifier's identity at a random point
    x = rnd.randrange(2, R)
    dh, dn = be.to_device(h), be.to_device(num_coef)
    hx = orc.fr_to_ints(be.eval_polynomial_batch_dev([dh], h.shape[0], M([x]))[0:1])[0]
    nx = orc.fr_to_ints(be.eval_polynomial_batch_dev([dn], num_coef.shape[0], M([x]))[0:1])[0]
    if tamper:                                                   # an unsatisfied gate must NOT divide
        assert nx != hx * (pow(x, n, R) - 1) % R
        e.release()
        return
    assert nx == hx * (pow(x, n, R) - 1) % R
    assert nx != 0
    # and h is a genuine polynomial of degree < (deg-1)*n: the vanishing division left no remainder, i.e. the numerator
    # (degree < deg*n) reconstructs exactly from h
    nz = orc.fr_to_ints(num_coef)
    hz = orc.fr_to_ints(h)
    recon = [0] * (len(hz) + n)
    for i, cf in enumerate(hz):
        recon[i + n] = (recon[i + n] + cf) % R
        recon[i] = (recon[i] - cf) % R
    assert recon == nz[: len(recon)] and all(v == 0 for v in nz[len(recon):])
    e.release()


@pytest.mark.parametrize("k", [4, 5])
def test_toy_circuit_on_emulator(emu, orc, pyref, k):
    _build_and_check(emu, orc, pyref, k, seed=k)


def test_toy_circuit_negative_control(emu, orc, pyref):
    _build_and_check(emu, orc, pyref, 4, seed=4, tamper=True)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [5, 9])
def test_toy_circuit_on_gpu(gpu, orc, pyref, k):
    _build_and_check(gpu, orc, pyref, k, seed=100 + k)
