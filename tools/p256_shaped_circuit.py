#!/usr/bin/env python3
"""BASELINE configs[0]: the shape of the reference's stack-B circuit (crates/p256-ecdsa at k = 18, the case the reference itself runs on the CPU), as read off its golden
proof bin/assets/proof.bin.  A satisfiable synthetic circuit with that census — the real one needs halo2-lib's ECDSA chip, which is not vendored."""
from zk_dcap_verifier_amd import plonk
from zk_dcap_verifier_amd.fields import R_MOD, fr_mont_array
from zk_dcap_verifier_amd.plonk import ADVICE, INSTANCE, Advice, Fixed


def build(k):
    """The census SURVEY App. B reads off the reference's bin/assets/proof.bin (stack B, crates/p256-ecdsa at k = 18): 2 gate advice
    columns (halo2-lib's vertical gate q * (a + b*c - d) over rotations 0..3) + 1 lookup advice column with a single-expression range
    lookup, fixed = 2 selectors + 1 constants column + 1 table, ONE instance column carrying 15 public limbs (lib.rs:79-89), equality
    on 2 advice + lookup advice + constants + instance => degree 4, 3 permutation sets, 3 h pieces."""
    n = 1 << k
    cs = plonk.ConstraintSystem(num_fixed_columns=4, num_advice_columns=3, num_instance_columns=1)
    for c in (0, 1):
        cs.create_gate(Fixed(c) * (Advice(c, 0) + Advice(c, 1) * Advice(c, 2) - Advice(c, 3)))
    cs.lookup([(Advice(2), Fixed(3))])
    for col in ((ADVICE, 0), (ADVICE, 1), (ADVICE, 2), (plonk.FIXED, 2), (INSTANCE, 0)):
        cs.enable_equality(*col)
    assert cs.degree() == 4
    u = cs.usable_rows(k)
    nblk = u // 4
    cols = []
    for c in (0, 1):
        col = [0] * n
        for t in range(nblk):
            a, b, cc = 3 * t + c + 1, t + 2, 5 * t + 7
            col[4 * t:4 * t + 4] = [a, b, cc, (a + b * cc) % R_MOD]
        cols.append(col)
    T = min(1 << 8, u)
    look = [(7 * i + 3) % T for i in range(n)]
    inst = [look[i] for i in range(15)]                              # 15 public values, each copied from the lookup advice column
    q = [[1 if (i % 4 == 0 and i // 4 < nblk) else 0 for i in range(n)] for _ in range(2)]
    const = [cols[0][1]] + [0] * (n - 1)                             # one constant, copied to the cell that must equal it
    table = [i if i < T else 0 for i in range(n)]
    asm = plonk.Assembly(cs, k)
    for i in range(15):
        asm.copy((INSTANCE, 0, i), (ADVICE, 2, i))
    asm.copy((plonk.FIXED, 2, 0), (ADVICE, 0, 1))
    asm.copy((ADVICE, 0, 5), (ADVICE, 1, 1))                         # b of block 1 in column 0 (= 3) ... must equal b of block 0 in column 1
    cols[1][1] = cols[0][5]
    cols[1][3] = (cols[1][0] + cols[1][1] * cols[1][2]) % R_MOD
    return cs, q + [const, table], asm, [fr_mont_array(cols[0]), fr_mont_array(cols[1]), fr_mont_array(look)], [inst]
