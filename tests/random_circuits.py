"""Random constraint systems for differential testing of create_proof: the product's provers (zk_plonk_create_proof and its Python twin, on the emulator or
the GPU) against the independent CPU prover (oracle/prover.py) — byte for byte.

create_proof never checks the gates (halo2 leaves that to MockProver / the verifier), so the witness need not satisfy them: the proof is a deterministic
function of (constraint system, fixed columns, copy constraints, witness, RNG stream) either way, and the numerator of h(X) that is NOT divisible by X^n - 1
exercises the truncation in extended_to_coeff as well.  Only the lookups must hold (an input outside its table is an error in halo2), so the lookup inputs are
drawn from the table rows.  What varies: the degree (3 .. 9, i.e. extended_k - k from 1 to 3 and 1 .. 7 permutation columns per set), rotations (-3 .. 3, which
move the blinding-factor count and the rotation sets of SHPLONK), the number of advice / fixed / instance columns, gates per system, lookups with 1-3
expression pairs that may share a table, equality-enabled columns of all three kinds and random copy cycles.
"""
import random

import numpy as np

from zk_dcap_verifier_amd import plonk
from zk_dcap_verifier_amd.fields import R_MOD, fr_mont_array
from zk_dcap_verifier_amd.plonk import ADVICE, FIXED, INSTANCE, Advice, Fixed, Instance


def random_circuit(k: int, seed: int):
    """-> (cs, fixed columns (int lists), assembly (with .copies logged), advice columns (Montgomery arrays), instances (int lists))"""
    rnd = random.Random(seed)
    n = 1 << k
    n_lookups = rnd.randint(0, 3)
    n_adv = rnd.randint(2, 5)
    n_inst = rnd.randint(0, 2)
    # lookups first decide how many dedicated columns exist: input j of lookup l reads its own advice column; tables are fixed columns, sometimes shared
    lookups, tables = [], []
    adv_next, fix_next = n_adv, 0
    n_gate_fixed = rnd.randint(1, 3)
    fix_next = n_gate_fixed
    for l in range(n_lookups):
        m = rnd.randint(1, 3)
        if tables and rnd.random() < 0.4:
            tcols = rnd.choice(tables)                                       # a second lookup into the same table (its columns are sorted once per proof)
            m = len(tcols)
        else:
            tcols = list(range(fix_next, fix_next + m))
            fix_next += m
            tables.append(tcols)
        sel = fix_next
        fix_next += 1
        lookups.append(dict(inputs=list(range(adv_next, adv_next + m)), table=tcols, selector=sel, product_form=rnd.random() < 0.5))
        adv_next += m
    n_adv_total, n_fix_total = adv_next, fix_next
    cs = plonk.ConstraintSystem(num_fixed_columns=n_fix_total, num_advice_columns=n_adv_total, num_instance_columns=n_inst)

    def query():
        kind = rnd.choice(["a", "a", "a", "f"] + (["i"] if n_inst else []))
        rot = rnd.choice([0, 0, 0, 1, -1, 2, -2, 3, -3])
        if kind == "a":
            return Advice(rnd.randrange(n_adv_total), rot)
        if kind == "f":
            return Fixed(rnd.randrange(n_gate_fixed), rot)
        return Instance(rnd.randrange(n_inst), rot)

    def linear():
        e = query()
        for _ in range(rnd.randint(0, 2)):
            t = query()
            e = e + t if rnd.random() < 0.5 else e - t * rnd.randrange(1, 1 << 20)
        if rnd.random() < 0.3:
            e = e + rnd.randrange(R_MOD)
        return e

    target = rnd.choice([2, 3, 4, 5, 6, 9])
    for _ in range(rnd.randint(1, 4)):
        deg = rnd.randint(1, target - 1)
        e = Fixed(rnd.randrange(n_gate_fixed))                                 # a selector-like factor
        for _ in range(deg):
            e = e * linear()
        if rnd.random() < 0.3:
            e = -e
        cs.create_gate(e)
    for lk in lookups:
        s = Fixed(lk["selector"])
        cs.lookup([((s * Advice(a)) if lk["product_form"] else Advice(a), Fixed(t)) for a, t in zip(lk["inputs"], lk["table"])])
    eq_cols = []
    for c in range(n_adv_total):
        if rnd.random() < 0.6:
            eq_cols.append((ADVICE, c))
    for c in range(n_inst):
        eq_cols.append((INSTANCE, c))
    if rnd.random() < 0.5:
        eq_cols.append((FIXED, rnd.randrange(n_gate_fixed)))
    rnd.shuffle(eq_cols)
    for col in eq_cols:
        cs.enable_equality(*col)

    u = cs.usable_rows(k)
    assert u >= 4, "k too small for this system's blinding factors"
    fixed = [[rnd.randrange(R_MOD) if rnd.random() < 0.7 else rnd.randrange(2) for _ in range(n)] for _ in range(n_fix_total)]
    advice = [[rnd.randrange(R_MOD) for _ in range(n)] for _ in range(n_adv_total)]
    for tcols in tables:                                                       # table rows: few distinct tuples, the all-zero tuple among them
        rows = [tuple(0 for _ in tcols)] + [tuple(rnd.randrange(1 << rnd.choice([4, 16, 200])) for _ in tcols) for _ in range(rnd.randint(1, max(2, u // 2)))]
        for i in range(n):
            tup = rows[i] if i < len(rows) else rows[rnd.randrange(len(rows))]
            for c, v in zip(tcols, tup):
                fixed[c][i] = v
    for lk in lookups:
        for i in range(n):
            if lk["product_form"]:
                on = rnd.random() < 0.7
                fixed[lk["selector"]][i] = 1 if on else 0
            else:
                on = True
            if on:
                t = rnd.randrange(u)                                            # any usable row of the table
                for a, c in zip(lk["inputs"], lk["table"]):
                    advice[a][i] = fixed[c][t]
    instances = [[rnd.randrange(R_MOD) for _ in range(rnd.randint(0, min(u, 6)))] for _ in range(n_inst)]
    asm = plonk.Assembly(cs, k)
    asm.copies = []
    if eq_cols:
        for _ in range(rnd.randint(0, 3 * len(eq_cols))):
            (ta, ca), (tb, cb) = rnd.choice(eq_cols), rnd.choice(eq_cols)

            asm.copy((ta, ca, rnd.randrange(u)), (tb, cb, rnd.randrange(u)))
    return cs, fixed, asm, [fr_mont_array(c) for c in advice], instances


def oracle_proof(k, tau, cs, fixed, asm, advice, instances, seed):
    """the independent CPU prover's bytes for this circuit and RNG stream"""
    import oracle as orc
    import prover as op
    ints = lambda col: orc.fr_to_ints(col) if isinstance(col, np.ndarray) else [int(v) for v in col]
    params = op.Params(k, tau)
    keys = op.keygen(params, cs, [ints(c) for c in fixed], asm.copies)
    return op.create_proof(params, keys, [ints(c) for c in advice], instances, np.random.default_rng(seed), require_satisfied=False)
