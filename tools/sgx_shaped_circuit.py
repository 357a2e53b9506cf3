"""A SATISFIABLE synthetic circuit with the census of the sgx_dcap_verifier circuit at k = 19 (SURVEY.md §3.1, §8d cfg 2):
25 advice columns, 18 fixed, 11 lookup arguments of 4-5 expressions, 16 equality-enabled columns (6 permutation sets at
degree 5), 24 degree-3 gates of the halo2-lib shape q * (a + b * c - d) over four consecutive rows of one column.

The reference's real circuit (circuits/src/sgx_dcap_verifier.rs:139-238, 351-733) cannot be synthesised here — its chips
live in un-vendored Rust crates — and prover cost does not depend on what the witness means, only on this shape
(SURVEY §0.8).  Unlike the op-mix of bench.py's headline this is a genuine constraint system with a genuine witness, so the
proof `create_proof` emits for it can be (and is) checked by a verifier.

Layout:  advice 0..13   "gate" columns, full-width field elements, blocks of 4 rows (a, b, c, d = a + b*c)
         advice 14..24  "lookup" columns, 16-bit values (range-checked limbs / decomposed bytes in the real circuit)
         fixed 0..4     table columns T_j[i] = (j + 1) * i, i < 2^16 (else 0);  fixed 5..17 selectors
         lookup l       inputs sel * w_l * (j + 1), j < 4|5  against (T_0 .. T_m-1);  degree 2 + 2 + 1 = 5
         equality       gate columns in pairs share their b cells; lookup columns 14/15 agree on the first half of the rows
Witness synthesis uses the device only as a field calculator (d = a + b*c); it is setup, never part of a timed proof.
"""
from __future__ import annotations

import numpy as np

N_GATE_COLS, N_LOOKUP_COLS, N_FIXED, N_TABLE_COLS, N_GATES, TABLE_BITS = 14, 11, 18, 5, 24, 16


def build(z, be, k: int, seed: int = 20241008, table_bits: int = TABLE_BITS):
    """-> (cs, fixed_columns [(n,4) uint64], assembly, advice_columns [(n,4) uint64])"""
    plonk, F = z.plonk, z.fields
    Advice, Fixed = plonk.Advice, plonk.Fixed
    n = 1 << k
    A = N_GATE_COLS + N_LOOKUP_COLS
    n_sel = N_FIXED - N_TABLE_COLS
    cs = plonk.ConstraintSystem(num_fixed_columns=N_FIXED, num_advice_columns=A, num_instance_columns=0)
    for g in range(N_GATES):
        col, sel = g % N_GATE_COLS, N_TABLE_COLS + g % n_sel
        cs.create_gate(Fixed(sel) * (Advice(col, 0) + Advice(col, 1) * Advice(col, 2) - Advice(col, 3)))
    for l in range(N_LOOKUP_COLS):
        col, sel, m = N_GATE_COLS + l, N_TABLE_COLS + l % n_sel, 4 if l % 2 == 0 else 5
        base = Fixed(sel) * Advice(col)
        cs.lookup([(base if j == 0 else base * (j + 1), Fixed(j)) for j in range(m)])
    for c in range(N_GATE_COLS):
        cs.enable_equality(plonk.ADVICE, c)
    cs.enable_equality(plonk.ADVICE, N_GATE_COLS)
    cs.enable_equality(plonk.ADVICE, N_GATE_COLS + 1)
    assert cs.degree() == 5
    u = cs.usable_rows(k)
    nblk = u // 4                                                    # complete 4-row blocks inside the usable rows
    rng = np.random.default_rng(seed)
    one = F.fr_mont(1)

    # ---- fixed columns ------------------------------------------------------------------------------------------------
    T = min(1 << table_bits, u)
    fixed = []
    for j in range(N_TABLE_COLS):
        col = np.zeros((n, 4), dtype=np.uint64)
        col[:T] = F.fr_mont_array([(j + 1) * i for i in range(T)])
        fixed.append(col)
    for s in range(n_sel):
        col = np.zeros((n, 4), dtype=np.uint64)
        rows = np.arange(nblk, dtype=np.int64)
        rows = rows[rows % 2 == s % 2] * 4                            # first row of every other block
        col[rows] = one
        fixed.append(col)

    # ---- advice: gate columns (pairs share b) ---------------------------------------------------------------------------
    advice = []
    b_shared = None
    for c in range(N_GATE_COLS):
        a_, c_ = F.rand_fr_array(rng, nblk), F.rand_fr_array(rng, nblk)
        if c % 2 == 0:
            b_shared = F.rand_fr_array(rng, nblk)
        da, db, dc = be.to_device(a_), be.to_device(b_shared), be.to_device(c_)
        be.fr_mul_dev(db, dc, dc, nblk)
        be.fr_add_dev(da, dc, dc, nblk)                               # d = a + b*c
        d_ = dc.download((nblk, 4))
        for x in (da, db, dc):
            x.free()
        col = F.rand_fr_array(rng, n)                                 # cells outside the blocks: unconstrained
        blk = col[: 4 * nblk].reshape(nblk, 4, 4)
        blk[:, 0], blk[:, 1], blk[:, 2], blk[:, 3] = a_, b_shared, c_, d_
        advice.append(col)
    # ---- advice: lookup columns (16-bit values; 14 and 15 agree on the first half) -----------------------------------------
    small = F.fr_mont_array(list(range(T)))
    half = u // 2
    w_prev = None
    for l in range(N_LOOKUP_COLS):
        w = rng.integers(0, T, size=n)
        if l == 1:
            w[:half] = w_prev[:half]
        w_prev = w
        advice.append(small[w])

    # ---- copy constraints ------------------------------------------------------------------------------------------------------
    asm = plonk.Assembly(cs, k)
    if k <= 10:
        asm.copies = []                                               # small sizes: keep the copy list (the independent CPU prover of oracle/ replays it)
    brow = np.arange(nblk, dtype=np.int64) * 4 + 1
    for c in range(0, N_GATE_COLS, 2):
        asm.copy_rows((plonk.ADVICE, c), (plonk.ADVICE, c + 1), brow)
    asm.copy_rows((plonk.ADVICE, N_GATE_COLS), (plonk.ADVICE, N_GATE_COLS + 1), np.arange(half, dtype=np.int64))
    return cs, fixed, asm, advice
