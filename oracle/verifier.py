"""TEST INFRASTRUCTURE ONLY — pure-Python halo2 verifier (plonk::verify_proof + VerifierSHPLONK) over big integers.

The reference's own tests pin the prover hot path in exactly one way: the proof produced by `create_proof` must be
ACCEPTED by `verify_proof` (circuits/src/sgx_dcap_verifier.rs:826-844; crates/p256-ecdsa/src/base.rs:214-247 —
SURVEY.md §4, §8a row a6).  This file is that acceptance oracle for the GPU prover mirror
(zk-dcap-verifier_amd/plonk/prover.py): an independent restatement of halo2_proofs 0.2.0 (zkwebauthn @ c254c75,
Cargo.lock:1314-1327) src/plonk/verifier.rs, src/plonk/{permutation,lookup,vanishing}/verifier.rs and
src/poly/kzg/multiopen/shplonk/verifier.rs ([3P-MEM]: restated from the published protocol, the crate is not on this
machine).  It shares NO arithmetic with the product: field / curve operations are Python ints (oracle/pyref.py), the
transcript reader and the expression walker are restated here.

The final pairing check e(h2, [tau]G2) = e(outer, G2) is replaced by the equivalent G1 identity [tau] h2 == outer,
using the SRS trapdoor tau that the TEST generated (ParamsKZG.setup(k, tau)): G1 has prime order and the pairing is
non-degenerate, so the two statements are equivalent whenever tau is known — which is only ever true in tests.
Nothing in the product path may import this module.
"""
from __future__ import annotations

import hashlib

import pyref as p

R, P = p.R, p.P


# ---- transcript (Blake2bRead / Challenge255) ------------------------------------------------------------
class _Reader:
    def __init__(self, proof: bytes):
        self.h = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.proof, self.pos = bytes(proof), 0

    def squeeze(self) -> int:
        self.h.update(b"\x00")
        return int.from_bytes(self.h.copy().digest(), "little") % R

    def common_scalar(self, s: int):
        self.h.update(b"\x02" + (s % R).to_bytes(32, "little"))

    def common_point(self, pt):
        if pt is None:                                               # Blake2bRead::common_point fails on the identity, so verify_proof returns an error
            raise ValueError("cannot write points at infinity to the transcript")
        x, y = pt
        self.h.update(b"\x01" + x.to_bytes(32, "little") + y.to_bytes(32, "little"))

    def _take(self) -> bytes:
        if self.pos + 32 > len(self.proof):
            raise ValueError("proof too short")
        self.pos += 32
        return self.proof[self.pos - 32:self.pos]

    def read_point(self):
        b = self._take()
        if b == bytes(32):
            pt = None
        else:
            x = int.from_bytes(b, "little") & ((1 << 255) - 1)
            ys = p.g1_decompress_x(x) if x < P else None
            if ys is None:
                raise ValueError("commitment not on the curve")
            y = ys[0] if (ys[0] & 1) == (b[31] >> 7) else ys[1]
            pt = (x, y)
        self.common_point(pt)
        return pt

    def read_scalar(self) -> int:
        s = int.from_bytes(self._take(), "little")
        if s >= R:
            raise ValueError("evaluation not canonical")
        self.common_scalar(s)
        return s


# ---- expression walker over the circuit description (dataclasses of zk-dcap-verifier_amd/plonk/expression.py) ---------------
def _eval_expr(e, fixed, advice, instance) -> int:
    t = type(e).__name__
    if t == "Constant":
        return e.value % R
    if t == "Fixed":
        return fixed[(e.column, e.rotation)]
    if t == "Advice":
        return advice[(e.column, e.rotation)]
    if t == "Instance":
        return instance[(e.column, e.rotation)]
    if t == "Negated":
        return -_eval_expr(e.a, fixed, advice, instance) % R
    if t == "Sum":
        return (_eval_expr(e.a, fixed, advice, instance) + _eval_expr(e.b, fixed, advice, instance)) % R
    if t == "Product":
        return _eval_expr(e.a, fixed, advice, instance) * _eval_expr(e.b, fixed, advice, instance) % R
    if t == "Scaled":
        return _eval_expr(e.a, fixed, advice, instance) * e.f % R
    raise TypeError(t)


def _lagrange_at(i: int, x: int, xn: int, k: int) -> int:
    """l_i(x) = omega^i (x^n - 1) / (n (x - omega^i))."""
    n, w = 1 << k, pow(p.omega(k), i % (1 << k), R)
    return w * (xn - 1) % R * pow(n * (x - w) % R, -1, R) % R


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


def verify_proof(vk, tau: int, instances, proof: bytes, reader=None) -> bool:
    """plonk::verify_proof with VerifierSHPLONK and a single circuit instance.  vk: the keygen output (k, cs, fixed and
    permutation commitments as canonical affine points, transcript_repr).  Returns True / False; malformed proofs raise ValueError.
    `reader`: transcript reader class (squeeze / common_scalar / common_point / read_point / read_scalar / pos / proof)."""
    cs, k = vk.cs, vk.k
    n = 1 << k
    w = p.omega(k)
    bf = cs.blinding_factors()
    L = len(cs.lookups)
    chunk = cs.permutation_chunk_len()
    n_perm = len(cs.permutation_columns)
    n_sets = (n_perm + chunk - 1) // chunk if n_perm else 0
    tr = (reader or _Reader)(proof)                                    # default: Blake2bRead (stack A); oracle/poseidon_ref.Reader for stack B's PoseidonTranscript
    tr.common_scalar(vk.transcript_repr)
    assert len(instances) == cs.num_instance_columns
    for col in instances:
        for v in col:
            tr.common_scalar(v)
    advice_c = [tr.read_point() for _ in range(cs.num_advice_columns)]
    theta = tr.squeeze()
    permuted_c = [(tr.read_point(), tr.read_point()) for _ in range(L)]
    beta, gamma = tr.squeeze(), tr.squeeze()
    perm_z_c = [tr.read_point() for _ in range(n_sets)]
    lookup_z_c = [tr.read_point() for _ in range(L)]
    random_c = tr.read_point()
    y = tr.squeeze()
    h_c = [tr.read_point() for _ in range(cs.degree() - 1)]
    x = tr.squeeze()
    xn = pow(x, n, R)
    rot = lambda r: x * pow(w, r % n, R) % R

    aq, fq, iq = cs.advice_queries(), cs.fixed_queries(), cs.instance_queries()
    advice_evals = {q: tr.read_scalar() for q in aq}
    fixed_evals = {q: tr.read_scalar() for q in fq}
    random_eval = tr.read_scalar()
    sigma_evals = [tr.read_scalar() for _ in range(n_perm)]
    perm_evals = []
    for i in range(n_sets):
        e = {"z": tr.read_scalar(), "z_next": tr.read_scalar()}
        if i + 1 < n_sets:
            e["z_last"] = tr.read_scalar()
        perm_evals.append(e)
    lookup_evals = [dict(zip(("z", "z_next", "a", "a_inv", "s"), [tr.read_scalar() for _ in range(5)])) for _ in range(L)]

    # instance evaluations are the verifier's own (KZG: the prover does not send them)
    instance_evals = {}
    for (c, r) in iq:
        pt = rot(r)
        ptn = pow(pt, n, R)
        instance_evals[(c, r)] = sum(v * _lagrange_at(i, pt, ptn, k) for i, v in enumerate(instances[c])) % R

    l_0 = _lagrange_at(0, x, xn, k)
    l_last = _lagrange_at(n - bf - 1, x, xn, k)
    l_blind = sum(_lagrange_at(n - bf + i, x, xn, k) for i in range(bf)) % R
    l_active = (1 - l_last - l_blind) % R

    exprs = [_eval_expr(g, fixed_evals, advice_evals, instance_evals) for g in cs.gates]
    # permutation argument
    col_eval = lambda t, i: {0: advice_evals, 1: fixed_evals, 2: instance_evals}[t][(i, 0)]
    if n_sets:
        exprs.append(l_0 * (1 - perm_evals[0]["z"]) % R)
        zl = perm_evals[-1]["z"]
        exprs.append(l_last * (zl * zl - zl) % R)
        for i in range(1, n_sets):
            exprs.append(l_0 * (perm_evals[i]["z"] - perm_evals[i - 1]["z_last"]) % R)
        for i in range(n_sets):
            cols = cs.permutation_columns[i * chunk:(i + 1) * chunk]
            left = perm_evals[i]["z_next"]
            right = perm_evals[i]["z"]
            cur_delta = beta * x % R * pow(p.DELTA, i * chunk, R) % R
            for j, (t, ci) in enumerate(cols):
                v = col_eval(t, ci)
                left = left * (v + beta * sigma_evals[i * chunk + j] + gamma) % R
                right = right * (v + cur_delta + gamma) % R
                cur_delta = cur_delta * p.DELTA % R
            exprs.append((left - right) * l_active % R)
    # lookup arguments (App. C.4 (iv))
    for lk, e in zip(cs.lookups, lookup_evals):
        def compress(es):
            acc = 0
            for ex_ in es:
                acc = (acc * theta + _eval_expr(ex_, fixed_evals, advice_evals, instance_evals)) % R
            return acc
        exprs.append(l_0 * (1 - e["z"]) % R)
        exprs.append(l_last * (e["z"] * e["z"] - e["z"]) % R)
        left = e["z_next"] * (e["a"] + beta) % R * (e["s"] + gamma) % R
        right = e["z"] * (compress(lk.input_expressions) + beta) % R * (compress(lk.table_expressions) + gamma) % R
        exprs.append((left - right) * l_active % R)
        exprs.append(l_0 * (e["a"] - e["s"]) % R)
        exprs.append((e["a"] - e["s"]) * (e["a"] - e["a_inv"]) % R * l_active % R)
    acc = 0
    for v in exprs:
        acc = (acc * y + v) % R
    expected_h_eval = acc * pow(xn - 1, -1, R) % R
    h_commitment = None
    for c in reversed(h_c):                                          # sum_i xn^i H_i
        h_commitment = p.g1_add(p.g1_mul(h_commitment, xn), c)

    # ---- queries, in the prover's order --------------------------------------------------------------------------------------
    x_next, x_inv, x_last = rot(1), rot(-1), rot(-(bf + 1))
    Q = []                                                           # (commitment key, commitment point, point, eval)
    for (c, r) in aq:
        Q.append((("adv", c), advice_c[c], rot(r), advice_evals[(c, r)]))
    for i in range(n_sets):
        Q.append((("pz", i), perm_z_c[i], x, perm_evals[i]["z"]))
        Q.append((("pz", i), perm_z_c[i], x_next, perm_evals[i]["z_next"]))
    for i in reversed(range(n_sets - 1)):
        Q.append((("pz", i), perm_z_c[i], x_last, perm_evals[i]["z_last"]))
    for i, e in enumerate(lookup_evals):
        Q.append((("lz", i), lookup_z_c[i], x, e["z"]))
        Q.append((("la", i), permuted_c[i][0], x, e["a"]))
        Q.append((("ls", i), permuted_c[i][1], x, e["s"]))
        Q.append((("la", i), permuted_c[i][0], x_inv, e["a_inv"]))
        Q.append((("lz", i), lookup_z_c[i], x_next, e["z_next"]))
    for (c, r) in fq:
        Q.append((("fix", c), vk.fixed_commitments[c], rot(r), fixed_evals[(c, r)]))
    for j in range(n_perm):
        Q.append((("sig", j), vk.permutation_commitments[j], x, sigma_evals[j]))
    Q.append((("h",), h_commitment, x, expected_h_eval))
    Q.append((("rand",), random_c, x, random_eval))

    # ---- VerifierSHPLONK ---------------------------------------------------------------------------------------------------------
    yy = tr.squeeze()
    super_points = sorted({q[2] for q in Q})
    order, info = [], {}
    for key, com, pt, ev in Q:
        if key not in info:
            info[key] = {"c": com, "pts": {}}
            order.append(key)
        info[key]["pts"].setdefault(pt, ev)
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
    h1 = tr.read_point()
    u = tr.squeeze()
    h2 = tr.read_point()
    if tr.pos != len(tr.proof):
        raise ValueError("trailing bytes in proof")
    vanish = lambda roots, z: __import__("functools").reduce(lambda a, r: a * (z - r) % R, roots, 1)
    outer, r_outer = None, 0
    z_0 = z_0_diff_inv = 0
    vpow = 1
    for i, (pts, keys) in enumerate(sets):
        z_diff = vanish([q for q in super_points if q not in pts], u)
        if i == 0:
            z_0 = vanish(pts, u)
            z_0_diff_inv = pow(z_diff, -1, R)
            z_diff = 1
        else:
            z_diff = z_diff * z_0_diff_inv % R
        inner, r_inner, ypow = None, 0, 1
        for key in keys:
            evals = [info[key]["pts"][q] for q in pts]
            r_x = _interpolate(list(pts), evals)
            r_inner = (r_inner + ypow * p.poly_eval(r_x, u)) % R
            inner = p.g1_add(inner, p.g1_mul(info[key]["c"], ypow))
            ypow = ypow * yy % R
        outer = p.g1_add(outer, p.g1_mul(inner, vpow * z_diff % R))
        r_outer = (r_outer + vpow * r_inner % R * z_diff) % R
        vpow = vpow * v % R
    outer = p.g1_add(outer, p.g1_mul(p.G1_GEN, (-r_outer) % R))
    outer = p.g1_add(outer, p.g1_mul(h1, (-z_0) % R))
    outer = p.g1_add(outer, p.g1_mul(h2, u))
    return p.g1_mul(h2, tau) == outer
