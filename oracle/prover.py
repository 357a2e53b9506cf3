"""TEST INFRASTRUCTURE ONLY — an independent CPU prover: halo2's keygen_vk / keygen_pk / create_proof / ProverSHPLONK over Python integers.

Why it exists: the reference's prover IS a CPU `create_proof` (circuits/src/sgx_dcap_verifier.rs:799-823: gen_srs, keygen_vk, keygen_pk,
create_proof with a Blake2b transcript; crates/p256-ecdsa/src/base.rs:193-212 for stack B) and its only pin on the prover is that
`verify_proof` accepts (:826-844).  The GPU prover (zk-dcap-verifier_amd/plonk/prover.py) was so far compared with proofs made by its own
kernel sources on the emulator — a self-comparison.  This file produces the golden proofs instead (tools/gen_golden_proof.py): same SRS, same
witness, same seeded RNG stream, same transcript => the GPU must emit THESE bytes (tests/test_create_proof.py).

Restated from the published protocol of halo2_proofs 0.2.0 (zkwebauthn @ c254c75, Cargo.lock:1314-1327): src/plonk/{keygen,prover}.rs,
src/plonk/{permutation,lookup,vanishing}/prover.rs, src/plonk/permutation/keygen.rs (Assembly), src/poly/domain.rs,
src/poly/kzg/{commitment.rs, multiopen/shplonk/prover.rs} ([3P-MEM]: the pinned crate is not on this machine — SURVEY.md §3.1, App. C; parity
with the Rust prover's bytes stays UNPINNED).  It shares no code with the product: polynomials are lists of canonical
Python ints, NTTs are oracle/pyref.py's, the quotient is evaluated from its DEFINITION row by row (no GraphEvaluator, no ZKQ1 program),
lookups are permuted with sorted() and a dict, grand products use pow(x, -1, r); only the multi-scalar multiplications go through the C
oracle (oracle/bn254_oracle.c, halo2's best_multiexp restated).  The circuit description (`cs`: column counts, gate / lookup expression
trees, equality columns, query lists) is read as DATA, the way oracle/verifier.py reads it.  Sized for k <= 10.

Spec points that are this repo's (mirror = oracle, neither checkable against Rust here): vk.transcript_repr (Blake2b of the key's Debug-like
rendering), the rejection sampler standing in for Fr::random, the order of the random draws.
"""
from __future__ import annotations

import hashlib

import numpy as np

import oracle as orc
import pyref as p

R = p.R
ADVICE, FIXED, INSTANCE = 0, 1, 2
_RINV = pow(1 << 256, -1, R)


# ---- Fr::random stand-in (the mirror's sampler, restated): uniform raw 254-bit draws rejected at r, rejected rows redrawn; a raw limb
# pattern v is the Montgomery form of v / R --------------------------------------------------------------------------------------------------
def rand_fr(rng: np.random.Generator, n: int) -> list:
    mask = np.uint64((1 << 62) - 1)
    out = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    out[:, 3] &= mask
    vals = [sum(int(out[i, j]) << (64 * j) for j in range(4)) for i in range(n)]
    bad = [i for i, v in enumerate(vals) if v >= R]
    while bad:
        a = rng.integers(0, 1 << 64, size=(len(bad), 4), dtype=np.uint64)
        a[:, 3] &= mask
        nxt = []
        for t, i in enumerate(bad):
            vals[i] = sum(int(a[t, j]) << (64 * j) for j in range(4))
            if vals[i] >= R:
                nxt.append(i)
        bad = nxt
    return [v * _RINV % R for v in vals]


# ---- transcript (Blake2bWrite / Challenge255) ---------------------------------------------------------------------------
class _Writer:
    def __init__(self):
        self.h = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.out = bytearray()

    def squeeze(self) -> int:
        self.h.update(b"\x00")
        return int.from_bytes(self.h.copy().digest(), "little") % R

    def common_scalar(self, s: int):
        self.h.update(b"\x02" + (s % R).to_bytes(32, "little"))

    def write_scalar(self, s: int):
        self.common_scalar(s)
        self.out += (s % R).to_bytes(32, "little")

    def write_point(self, pt):
        if pt is None:                                               # halo2's common_point refuses the identity (it has no coordinates)
            raise ValueError("cannot write points at infinity to the transcript")
        x, y = pt
        self.h.update(b"\x01" + x.to_bytes(32, "little") + y.to_bytes(32, "little"))
        b = bytearray(x.to_bytes(32, "little"))
        b[31] |= (y & 1) << 7
        self.out += b


# ---- polynomial helpers over lists of ints -------------------------------------------------------------------------------------------------
def _intt(vals, k):
    n = 1 << k
    ninv = pow(n, -1, R)
    return [v * ninv % R for v in p.ntt_fast(list(vals), pow(p.omega(k), -1, R))]


def _poly_eval(c, x):
    acc = 0
    for v in reversed(c):
        acc = (acc * x + v) % R
    return acc


def _kate_division(a, b):
    """(a(X) - a(b)) / (X - b): q[n-2] = a[n-1]; q[i-1] = a[i] + b q[i]"""
    q = [0] * (len(a) - 1)
    carry = 0
    for i in range(len(a) - 1, 0, -1):
        carry = (a[i] + b * carry) % R
        q[i - 1] = carry
    return q


def _interpolate(points, evals):
    n = len(points)
    coeffs = [0] * n
    for j in range(n):
        num, den = [1], 1
        for m in range(n):
            if m != j:
                num = [(-points[m] * num[0]) % R] + [(num[i - 1] - points[m] * num[i]) % R for i in range(1, len(num))] + [num[-1]]
                den = den * (points[j] - points[m]) % R
        sc = evals[j] * pow(den, -1, R) % R
        for i, c in enumerate(num):
            coeffs[i] = (coeffs[i] + c * sc) % R
    return coeffs


def _eval_expr(e, fx, ad, ins):
    """Expression::evaluate; fx / ad / ins: (column, rotation) -> int"""
    t = type(e).__name__
    if t == "Constant":
        return e.value % R
    if t == "Fixed":
        return fx(e.column, e.rotation)
    if t == "Advice":
        return ad(e.column, e.rotation)
    if t == "Instance":
        return ins(e.column, e.rotation)
    if t == "Negated":
        return -_eval_expr(e.a, fx, ad, ins) % R
    if t == "Sum":
        return (_eval_expr(e.a, fx, ad, ins) + _eval_expr(e.b, fx, ad, ins)) % R
    if t == "Product":
        return _eval_expr(e.a, fx, ad, ins) * _eval_expr(e.b, fx, ad, ins) % R
    if t == "Scaled":
        return _eval_expr(e.a, fx, ad, ins) * e.f % R
    raise TypeError(t)


# ---- permutation::keygen::Assembly (cycle merge by swapping successors) -------------------------------------------------------------------
class Assembly:
    def __init__(self, columns, n):
        self.columns = list(columns)
        m = len(self.columns)
        self.mapping = [[(j, i) for i in range(n)] for j in range(m)]
        self.aux = [[(j, i) for i in range(n)] for j in range(m)]
        self.sizes = [[1] * n for _ in range(m)]

    def copy(self, left, right):
        lc, lr = self.columns.index((left[0], left[1])), left[2]
        rc, rr = self.columns.index((right[0], right[1])), right[2]
        lcyc, rcyc = self.aux[lc][lr], self.aux[rc][rr]
        if lcyc == rcyc:
            return
        if self.sizes[lcyc[0]][lcyc[1]] < self.sizes[rcyc[0]][rcyc[1]]:
            lcyc, rcyc = rcyc, lcyc
            lc, lr, rc, rr = rc, rr, lc, lr
        self.sizes[lcyc[0]][lcyc[1]] += self.sizes[rcyc[0]][rcyc[1]]
        i, j = rcyc
        while True:
            self.aux[i][j] = lcyc
            i, j = self.mapping[i][j]
            if (i, j) == rcyc:
                break
        self.mapping[lc][lr], self.mapping[rc][rr] = self.mapping[rc][rr], self.mapping[lc][lr]


# ---- SRS with a known trapdoor (tests only): g[i] = [tau^i] G, g_lagrange[i] = [l_i(tau)] G straight from the definition of l_i ---------------
def _g1_table(scalars):
    gen = orc.g1_generator()
    sm = orc.fr_from_ints(scalars)
    return np.stack([orc.g1_to_affine(orc.g1_mul(gen, sm[i]))[0] for i in range(len(scalars))])


class Params:
    def __init__(self, k: int, tau: int):
        self.k, self.n, self.tau = k, 1 << k, tau % R
        n, w = self.n, p.omega(k)
        self.g = _g1_table([pow(self.tau, i, R) for i in range(n)])
        tn1 = (pow(self.tau, n, R) - 1) % R
        self.g_lagrange = _g1_table([pow(w, i, R) * tn1 % R * pow(n * (self.tau - pow(w, i, R)) % R, -1, R) % R for i in range(n)])

    def _msm(self, scalars, bases):
        jac = orc.best_multiexp(orc.fr_from_ints(scalars), bases[:len(scalars)])
        return orc.g1_affine_to_ints(orc.g1_to_affine(jac))[0]

    def commit_lagrange(self, values):
        assert len(values) == self.n
        return self._msm(values, self.g_lagrange)

    def commit(self, coeffs):
        assert len(coeffs) <= self.n
        return self._msm(coeffs, self.g)


class Keys:
    """what keygen_vk + keygen_pk leave behind (ints)"""


def keygen(params: Params, cs, fixed_columns, copies=()) -> Keys:
    """fixed_columns: cs.num_fixed_columns lists of n canonical ints (selectors included); copies: ((type, index, row), (type, index, row)) pairs in
    the order the circuit's synthesize issued them."""
    k, n = params.k, params.n
    keys = Keys()
    keys.cs, keys.k = cs, k
    keys.fixed_values = [[int(v) % R for v in col] for col in fixed_columns]
    assert len(keys.fixed_values) == cs.num_fixed_columns and all(len(c) == n for c in keys.fixed_values)
    asm = Assembly(cs.permutation_columns, n)
    for left, right in copies:
        asm.copy(left, right)
    w = p.omega(k)
    wp = [pow(w, i, R) for i in range(n)]
    keys.sigma_values = [[pow(p.DELTA, cj, R) * wp[ri] % R for (cj, ri) in asm.mapping[j]] for j in range(len(asm.columns))]
    keys.fixed_commitments = [params.commit_lagrange(c) for c in keys.fixed_values]
    keys.permutation_commitments = [params.commit_lagrange(c) for c in keys.sigma_values]
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
    h.update(repr((k, cs.num_fixed_columns, cs.num_advice_columns, cs.num_instance_columns, len(cs.gates), len(cs.lookups),
                   cs.permutation_columns, cs.degree(), keys.fixed_commitments, keys.permutation_commitments)).encode())
    keys.transcript_repr = int.from_bytes(h.digest(), "little") % R
    keys.fixed_polys = [_intt(c, k) for c in keys.fixed_values]
    keys.sigma_polys = [_intt(c, k) for c in keys.sigma_values]
    return keys


def _extended_k(cs, k):
    ek = k
    while (1 << ek) < (1 << k) * (cs.degree() - 1):
        ek += 1
    return ek


def _coset_evals(coeffs, k, ek):
    """EvaluationDomain::coeff_to_extended, by definition: evaluations at ZETA * extended_omega^idx"""
    ext_w = pow(p.ROOT_OF_UNITY, 1 << (p.S - ek), R)
    a = [c * pow(p.ZETA, i % 3, R) % R for i, c in enumerate(coeffs)] + [0] * ((1 << ek) - len(coeffs))
    return p.ntt_fast(a, ext_w)


def _permute_expression_pair(inp, tab, usable, blind_in, blind_tab):
    """lookup::prover::permute_expression_pair: A' = sorted(A); S'[row] = A'[row] at the first row of every run (taken out of the table multiset),
    the remaining (repeated-input) rows, LAST row first, filled with the leftover table values in ascending order."""
    a = sorted(inp[:usable])
    left = {}
    for v in tab[:usable]:
        left[v] = left.get(v, 0) + 1
    s, repeated = [None] * usable, []
    for row, v in enumerate(a):
        if row == 0 or v != a[row - 1]:
            if not left.get(v):
                raise ValueError("ConstraintSystemFailure: lookup input not in table")
            s[row] = v
            left[v] -= 1
        else:
            repeated.append(row)
    rest = [v for v in sorted(left) for _ in range(left[v])]
    assert len(rest) == len(repeated)
    for v in rest:                                                   # `permuted_table_coeffs[repeated_input_rows.pop().unwrap()] = *coeff`: rows from the back
        s[repeated.pop()] = v
    return a + list(blind_in), s + list(blind_tab)


def create_proof(params: Params, keys: Keys, advice, instances, rng: np.random.Generator, require_satisfied: bool = True) -> bytes:
    """advice: cs.num_advice_columns lists of n canonical ints (rows past the usable ones are overwritten with blinding); instances: lists of
    canonical ints.  Returns the proof bytes (Blake2b transcript).  require_satisfied = False: behave as halo2 does on a witness that violates a gate —
    no check, extended_to_coeff silently truncates h(X) to (d-1) n coefficients (poly/domain.rs) — instead of stopping (differential tests on random circuits).
    Every Fr::random of halo2's provers is drawn in place, including the Blind each commitment draws (`blind()` below: drawn and dropped, as KZG does)."""

    def blind(times=1):                                              # `Blind(Scheme::Scalar::random(&mut rng))`: the stream advances, the value is unused under KZG
        for _ in range(times):
            rand_fr(rng, 1)
    cs, k, n = keys.cs, params.k, params.n
    w = p.omega(k)
    ek = _extended_k(cs, k)
    en, step = 1 << ek, 1 << (ek - k)
    bf = cs.blinding_factors()
    usable = n - (bf + 1)
    L = len(cs.lookups)
    tr = _Writer()
    # 1 ---------------------------------------------------------------------------------------------------------------------------------------
    tr.common_scalar(keys.transcript_repr)
    inst_values = []
    for col in instances:
        for v in col:
            tr.common_scalar(v)
        inst_values.append([int(v) % R for v in col] + [0] * (n - len(col)))
    # 2 ---------------------------------------------------------------------------------------------------------------------------------------
    adv_values = []
    for col in advice:
        col = [int(v) % R for v in col]
        assert len(col) == n
        col[usable:] = rand_fr(rng, n - usable)
        adv_values.append(col)
    blind(len(adv_values))                                           # plonk/prover.rs: `let blinds: Vec<_> = advice_values.iter().map(|_| Blind(random)).collect()`
    for col in adv_values:
        tr.write_point(params.commit_lagrange(col))
    # 3 ---------------------------------------------------------------------------------------------------------------------------------------
    theta = tr.squeeze()
    lag = {ADVICE: adv_values, FIXED: keys.fixed_values, INSTANCE: inst_values}

    def compress(exprs):
        out = []
        for i in range(n):
            fx = lambda c, r: keys.fixed_values[c][(i + r) % n]
            ad = lambda c, r: adv_values[c][(i + r) % n]
            ins = lambda c, r: inst_values[c][(i + r) % n]
            acc = 0
            for e in exprs:
                acc = (acc * theta + _eval_expr(e, fx, ad, ins)) % R
            out.append(acc)
        return out
    compressed = [(compress(lk.input_expressions), compress(lk.table_expressions)) for lk in cs.lookups]
    permuted = []                                                    # lookup/prover.rs commit_permuted, lookup by lookup: permute_expression_pair extends the input
    for cin_, ctab_ in compressed:                                   # then the table with random rows, commit_values draws one Blind per commitment
        bi_ = rand_fr(rng, bf + 1)
        bt_ = rand_fr(rng, bf + 1)
        permuted.append(_permute_expression_pair(cin_, ctab_, usable, bi_, bt_))
        blind(2)
    for a_, s_ in permuted:
        tr.write_point(params.commit_lagrange(a_))
        tr.write_point(params.commit_lagrange(s_))
    # 4 ---------------------------------------------------------------------------------------------------------------------------------------
    beta, gamma = tr.squeeze(), tr.squeeze()
    perm_cols = [lag[t][i] for t, i in cs.permutation_columns]
    chunk = cs.permutation_chunk_len()
    n_sets = (len(perm_cols) + chunk - 1) // chunk if perm_cols else 0
    perm_blind, lookup_blind = [], []
    for _ in range(n_sets):                                          # permutation/prover.rs: the set's blinding rows, then its Blind
        perm_blind.append(rand_fr(rng, bf))
        blind()
    for _ in range(L):                                               # lookup/prover.rs commit_product: the same per lookup
        lookup_blind.append(rand_fr(rng, bf))
        blind()
    random_poly = rand_fr(rng, n)                                    # vanishing/prover.rs commit: n coefficients, one Blind; construct: one Blind per piece of h(X)
    blind(1 + (cs.degree() - 1))
    wp = [pow(w, i, R) for i in range(n)]
    zs, last_z = [], 1
    for s in range(n_sets):
        z = [last_z]
        for i in range(n - 1):
            num = den = 1
            for j in range(s * chunk, min((s + 1) * chunk, len(perm_cols))):
                v = perm_cols[j][i]
                num = num * (v + pow(p.DELTA, j, R) * beta % R * wp[i] + gamma) % R
                den = den * (v + beta * keys.sigma_values[j][i] + gamma) % R
            z.append(z[-1] * num % R * pow(den, -1, R) % R)
        z[n - bf:] = perm_blind[s]
        last_z = z[n - (bf + 1)]
        zs.append(z)
    lzs = []
    for j, ((cin, ctab), (pin, ptab)) in enumerate(zip(compressed, permuted)):
        z = [1]
        for i in range(n - 1):
            num = (cin[i] + beta) * (ctab[i] + gamma) % R
            den = (pin[i] + beta) * (ptab[i] + gamma) % R
            z.append(z[-1] * num % R * pow(den, -1, R) % R)
        z[n - bf:] = lookup_blind[j]
        lzs.append(z)
    for z in zs + lzs:
        tr.write_point(params.commit_lagrange(z))
    # 5 ---------------------------------------------------------------------------------------------------------------------------------------
    tr.write_point(params.commit(random_poly))
    # 6 ---------------------------------------------------------------------------------------------------------------------------------------
    y = tr.squeeze()
    adv_polys = [_intt(c, k) for c in adv_values]
    inst_polys = [_intt(c, k) for c in inst_values]
    z_polys = [_intt(c, k) for c in zs]
    lz_polys = [_intt(c, k) for c in lzs]
    perm_polys = [(_intt(a_, k), _intt(s_, k)) for a_, s_ in permuted]
    ce = lambda c: _coset_evals(c, k, ek)
    A, F, I = [ce(c) for c in adv_polys], [ce(c) for c in keys.fixed_polys], [ce(c) for c in inst_polys]
    SG = [ce(c) for c in keys.sigma_polys]
    Z, LZ = [ce(c) for c in z_polys], [ce(c) for c in lz_polys]
    PA, PS = [ce(a_) for a_, _ in perm_polys], [ce(s_) for _, s_ in perm_polys]
    unit = lambda rows: ce(_intt([1 if i in rows else 0 for i in range(n)], k))
    l0, l_last = unit({0}), unit({n - bf - 1})
    l_active = unit(set(range(n - bf - 1)))
    ext_w = pow(p.ROOT_OF_UNITY, 1 << (p.S - ek), R)
    ext_cols = {ADVICE: A, FIXED: F, INSTANCE: I}
    last_rot = -(bf + 1)
    h_ext = []
    zeta_n = pow(p.ZETA, n, R)
    ext_w_n = pow(ext_w, n, R)
    for idx in range(en):
        at = lambda col, r: col[(idx + r * step) % en]
        fx = lambda c, r: at(F[c], r)
        ad = lambda c, r: at(A[c], r)
        ins = lambda c, r: at(I[c], r)
        X = p.ZETA * pow(ext_w, idx, R) % R
        value = 0
        for g in cs.gates:
            value = (value * y + _eval_expr(g, fx, ad, ins)) % R
        if n_sets:
            value = (value * y + (1 - Z[0][idx]) * l0[idx]) % R
            zl = Z[-1][idx]
            value = (value * y + (zl * zl - zl) * l_last[idx]) % R
            for s in range(1, n_sets):
                value = (value * y + (Z[s][idx] - at(Z[s - 1], last_rot)) * l0[idx]) % R
            for s in range(n_sets):
                left, right = at(Z[s], 1), Z[s][idx]
                for j in range(s * chunk, min((s + 1) * chunk, len(perm_cols))):
                    t, ci = cs.permutation_columns[j]
                    v = ext_cols[t][ci][idx]
                    left = left * (v + beta * SG[j][idx] + gamma) % R
                    right = right * (v + pow(p.DELTA, j, R) * beta % R * X + gamma) % R
                value = (value * y + (left - right) * l_active[idx]) % R
        for j, lk in enumerate(cs.lookups):
            def comp(es):
                acc = 0
                for e in es:
                    acc = (acc * theta + _eval_expr(e, fx, ad, ins)) % R
                return acc
            z_, a_, s_ = LZ[j][idx], PA[j][idx], PS[j][idx]
            value = (value * y + (1 - z_) * l0[idx]) % R
            value = (value * y + (z_ * z_ - z_) * l_last[idx]) % R
            left = at(LZ[j], 1) * (a_ + beta) % R * (s_ + gamma) % R
            right = z_ * (comp(lk.input_expressions) + beta) % R * (comp(lk.table_expressions) + gamma) % R
            value = (value * y + (left - right) * l_active[idx]) % R
            value = (value * y + (a_ - s_) * l0[idx]) % R
            value = (value * y + (a_ - s_) * (a_ - at(PA[j], -1)) % R * l_active[idx]) % R
        t_inv = pow((zeta_n * pow(ext_w_n, idx, R) - 1) % R, -1, R)           # 1 / (X^n - 1) on the coset
        h_ext.append(value * t_inv % R)
    # 7: back to coefficients (extended_to_coeff), split, commit -------------------------------------------------------------------------------
    en_inv = pow(en, -1, R)
    hc = [v * en_inv % R for v in p.ntt_fast(h_ext, pow(ext_w, -1, R))]
    zinv = pow(p.ZETA, -1, R)
    hc = [c * pow(zinv, i % 3, R) % R for i, c in enumerate(hc)]
    n_pieces = cs.degree() - 1
    assert not (require_satisfied and any(hc[n_pieces * n:])), "the witness does not satisfy the circuit: h(X) has degree >= (d-1) n"
    pieces = [hc[i * n:(i + 1) * n] for i in range(n_pieces)]
    for pc in pieces:
        tr.write_point(params.commit(pc))
    # 8 ---------------------------------------------------------------------------------------------------------------------------------------
    x = tr.squeeze()
    xn = pow(x, n, R)
    rot = lambda r: x * pow(w, r % n, R) % R
    h_poly = [sum(pow(xn, i, R) * pieces[i][c] for i in range(n_pieces)) % R for c in range(n)]
    x_last = rot(last_rot)
    Q = []                                                             # (key, poly, point) in evaluation order
    for c, r in cs.advice_queries():
        Q.append((("adv", c), adv_polys[c], rot(r)))
    for c, r in cs.fixed_queries():
        Q.append((("fix", c), keys.fixed_polys[c], rot(r)))
    Q.append((("rand",), random_poly, x))
    for j, sp in enumerate(keys.sigma_polys):
        Q.append((("sig", j), sp, x))
    for i, zp in enumerate(z_polys):
        Q.append((("pz", i), zp, x))
        Q.append((("pz", i), zp, rot(1)))
        if i + 1 < len(z_polys):
            Q.append((("pz", i), zp, x_last))
    for j in range(L):
        Q.append((("lz", j), lz_polys[j], x))
        Q.append((("lz", j), lz_polys[j], rot(1)))
        Q.append((("la", j), perm_polys[j][0], x))
        Q.append((("la", j), perm_polys[j][0], rot(-1)))
        Q.append((("ls", j), perm_polys[j][1], x))
    Q.append((("h",), h_poly, x))
    evals = [_poly_eval(poly, pt) for _, poly, pt in Q]
    for e in evals[:-1]:
        tr.write_scalar(e)
    # 9: ProverSHPLONK — queries in the multi-open order -------------------------------------------------------------------------------------------
    it = iter([(key, poly, pt, e) for (key, poly, pt), e in zip(Q, evals)])
    take = lambda: next(it)
    q_adv = [take() for _ in cs.advice_queries()]
    q_fix = [take() for _ in cs.fixed_queries()]
    q_rand = take()
    q_sig = [take() for _ in keys.sigma_polys]
    q_pa, q_pl = [], []
    for i in range(len(z_polys)):
        q_pa += [take(), take()]
        if i + 1 < len(z_polys):
            q_pl.append(take())
    q_lk = []
    for _ in range(L):
        pz, pzn, pa, pai, ps = take(), take(), take(), take(), take()
        q_lk += [pz, pa, ps, pai, pzn]
    q_h = take()
    queries = q_adv + q_pa + list(reversed(q_pl)) + q_lk + q_fix + q_sig + [q_h, q_rand]
    yy = tr.squeeze()
    super_points = sorted({q[2] for q in queries})
    order, info = [], {}
    for key, poly, pt, e in queries:
        if key not in info:
            info[key] = {"poly": poly, "pts": {}}
            order.append(key)
        info[key]["pts"].setdefault(pt, e)
    sets = []
    for key in order:
        pts = tuple(sorted(info[key]["pts"]))
        for s in sets:
            if s[0] == pts:
                s[1].append(key)
                break
        else:
            sets.append((pts, [key]))
    v = tr.squeeze()
    padn = lambda c: list(c) + [0] * (n - len(c))
    quotients, low = [], []
    for pts, ks in sets:
        acc, ypow, rs = [0] * n, 1, []
        for key in ks:
            r_x = _interpolate(list(pts), [info[key]["pts"][q] for q in pts])
            rs.append(r_x)
            poly = padn(info[key]["poly"])
            for i in range(n):
                acc[i] = (acc[i] + ypow * (poly[i] - (r_x[i] if i < len(r_x) else 0))) % R
            ypow = ypow * yy % R
        low.append(rs)
        for pt in pts:                                                 # exact division by the set's vanishing polynomial
            acc = _kate_division(acc, pt)
        quotients.append(padn(acc))
    h_x = [sum(pow(v, i, R) * q[c] for i, q in enumerate(quotients)) % R for c in range(n)]
    tr.write_point(params.commit(h_x))
    u = tr.squeeze()
    vanish = lambda roots, z_: __import__("functools").reduce(lambda a_, r_: a_ * (z_ - r_) % R, roots, 1)
    z_diffs = [vanish([q for q in super_points if q not in pts], u) for pts, _ in sets]
    z0_inv = pow(z_diffs[0], -1, R)
    lx = [0] * n
    for i, (pts, ks) in enumerate(sets):
        ypow = 1
        for key, r_x in zip(ks, low[i]):
            wgt = pow(v, i, R) * z_diffs[i] % R * ypow % R
            poly = padn(info[key]["poly"])
            for c in range(n):
                lx[c] = (lx[c] + wgt * poly[c]) % R
            lx[0] = (lx[0] - wgt * _poly_eval(r_x, u)) % R
            ypow = ypow * yy % R
    zt = vanish(super_points, u)
    lx = [(c - zt * hv) % R * z0_inv % R for c, hv in zip(lx, h_x)]
    tr.write_point(params.commit(padn(_kate_division(lx, u))))
    return bytes(tr.out)
