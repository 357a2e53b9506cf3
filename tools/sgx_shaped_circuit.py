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


def build(z, be, k: int, seed: int = 20241008, table_bits: int = TABLE_BITS, census: str = "chip_estimate"):
    """-> (cs, fixed_columns [(n,4) uint64], assembly, advice_columns [(n,4) uint64]).  census: "chip_estimate" (this function: every column a guess
    from SURVEY §3.1) or "reference_exact" (build_reference_exact below: the base64 part exactly as the reference configures and assigns it)."""
    if census == "reference_exact":
        return build_reference_exact(z, be, k, seed)
    # "full_chain_x4": BASELINE configs[4] / SURVEY 8d cfg 5 — the full DCAP chain (root -> intermediate -> leaf certificate ECDSA + QE3 + isv_report signatures,
    # several SHA-256) does not exist in the reference (README.md:23-47 is a roadmap); its op-mix is cfg 2's with the advice and lookup counts x 4, synthetic, at k = 21
    assert census in ("chip_estimate", "full_chain_x4"), census
    scale = 4 if census == "full_chain_x4" else 1
    return _build_chip_estimate(z, be, k, seed, table_bits, N_GATE_COLS * scale, N_LOOKUP_COLS * scale, N_TABLE_COLS + (N_FIXED - N_TABLE_COLS) * scale, N_GATES * scale)


def _build_chip_estimate(z, be, k, seed, table_bits, N_GATE_COLS, N_LOOKUP_COLS, N_FIXED, N_GATES):
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


# =====================================================================================================================================================
# Census B: what the reference FIXES EXACTLY, built exactly, plus the chip estimate for the rest.
#
# In-repo and exact (circuits/src/sgx_dcap_verifier.rs:139-238 configure, :260-329 assign; circuits/src/table/mod.rs:24-149):
#   advice 0..11   bit_decompositions (12 two-bit columns, :140-143, const 12 at :41)
#   advice 12      encoded_chars     13 decoded_chars     14 decoded_chars_without_gap            (:144-147; equality on all three, :151-153)
#   fixed 0..6     table columns character, value_encoded, value_decoded, bit_decompositions[0..3] (table/mod.rs:24-40; 65 + 257 assigned rows, :65-149;
#                  unassigned table rows take the column's first value, as halo2's table layouter fills them)
#   fixed 7        q_decode_selector (complex selector :149, enabled on every 4th of the first 1696 rows, :320-322)
#   7 lookups      4 "encoded" of 4 expressions, 3 "decoded" of 5 (:216-236 over create_bit_lookup :86-137): q*cell + (1-q)*default against the table columns
#   witness        1696 base64 characters (:40) on rows 0..1695, their 2-bit chunks on every 4th row, the 1272 decoded bytes twice (with and without gap),
#                  copy decoded_chars_without_gap[i] -> decoded_chars[i + i/3] (:281-292).  Rows >= 1696 of all 15 columns are ZERO: at k = 19 these
#                  columns are 99.7 % zeros, so their 15 commitments are almost free — the opposite of census A's 14 full-width random columns.
# [3P-MEM] estimate, same as census A for what the un-vendored chips add at the k = 19 configuration (configs/ecdsa_circuit.tmp.config: 1 advice, 1 lookup
# advice, lookup_bits 18; RangeConfig 3 + 1 advice, lookup_bits 16, :183-192; Sha256DynamicConfig :195-202): advice 15 = Fp-chip gate column (dense,
# 88-bit limbs and their products), 16 = its 18-bit lookup column, 17..19 = range-chip gate columns and 20 = their 16-bit lookup column (the SHA region
# was laid out for k = 17: a quarter of the rows at k = 19), 21..24 = SHA dense / spread pairs (two 2-expression lookups), fixed 8..17 selectors,
# constants and the 2^18 / 2^16 / spread tables; equality on the chip columns as halo2-lib enables it.
# =====================================================================================================================================================
B64_LEN = 1696


def _b64_value(ch: int) -> int:
    """table/mod.rs:54-63 map_character_to_encoded_value"""
    c = chr(ch)
    if c == "=":
        return 0
    if "A" <= c <= "Z":
        return ch - 65
    if "a" <= c <= "z":
        return ch - 71
    if "0" <= c <= "9":
        return ch + 4
    return 62 if c == "+" else 63


def build_reference_exact(z, be, k: int, seed: int = 20241008):
    import base64
    plonk, F = z.plonk, z.fields
    Advice, Fixed = plonk.Advice, plonk.Fixed
    n = 1 << k
    A, NF = 25, 18
    cs = plonk.ConstraintSystem(num_fixed_columns=NF, num_advice_columns=A, num_instance_columns=0)
    BITS, ENC, DEC, DEC_NOGAP = list(range(12)), 12, 13, 14
    T_CHAR, T_VENC, T_VDEC, T_BITS, Q_DECODE = 0, 1, 2, [3, 4, 5, 6], 7
    FP_GATE, FP_LOOK, R_GATE, R_LOOK, SHA = 15, 16, [17, 18, 19], 20, [21, 22, 23, 24]
    Q_FP, C_FP, T18, Q_R, C_R, T16, T_DENSE, T_SPREAD = 8, 9, 10, [11, 12, 13], 14, 15, 16, 17
    for c in (ENC, DEC, DEC_NOGAP):                                   # :151-153
        cs.enable_equality(plonk.ADVICE, c)
    # -- the chips' gates (configured before the base64 lookups, as FpConfig / RangeConfig / Sha256DynamicConfig::configure run first, :169-202) ----------
    cs.create_gate(Fixed(Q_FP) * (Advice(FP_GATE, 0) + Advice(FP_GATE, 1) * Advice(FP_GATE, 2) - Advice(FP_GATE, 3)))
    cs.lookup([(Advice(FP_LOOK), Fixed(T18))])
    for col, q in zip(R_GATE, Q_R):
        cs.create_gate(Fixed(q) * (Advice(col, 0) + Advice(col, 1) * Advice(col, 2) - Advice(col, 3)))
    cs.lookup([(Advice(R_LOOK), Fixed(T16))])
    cs.lookup([(Advice(SHA[0]), Fixed(T_DENSE)), (Advice(SHA[2]), Fixed(T_SPREAD))])
    cs.lookup([(Advice(SHA[1]), Fixed(T_DENSE)), (Advice(SHA[3]), Fixed(T_SPREAD))])
    for c in [FP_GATE, FP_LOOK] + R_GATE + [R_LOOK]:
        cs.enable_equality(plonk.ADVICE, c)
    cs.enable_equality(plonk.FIXED, C_FP)
    cs.enable_equality(plonk.FIXED, C_R)
    # -- the seven base64 lookups, :216-236 / :86-137 ---------------------------------------------------------------------------------------------------
    q = Fixed(Q_DECODE)
    one_minus_q = 1 - q
    for i in range(4):                                               # encoded: ENCODED_BIT_LOOKUP_COLS[i] = [3i, 3i+1, 3i+2] against table bits [2, 1, 0]
        pairs = [(q * Advice(ENC, i) + one_minus_q * 65, Fixed(T_CHAR))]
        for j, tcol in enumerate([2, 1, 0]):
            pairs.append((q * Advice(BITS[3 * i + j]) + one_minus_q * 0, Fixed(T_BITS[tcol])))
        cs.lookup(pairs)
    for i in range(3):                                               # decoded: DECODED_BIT_LOOKUP_COLS[i] = [4i .. 4i+3] against table bits [3, 2, 1, 0]
        pairs = [(q * Advice(DEC, i) + one_minus_q * 0, Fixed(T_VDEC))]
        for j, tcol in enumerate([3, 2, 1, 0]):
            pairs.append((q * Advice(BITS[4 * i + j]) + one_minus_q * 0, Fixed(T_BITS[tcol])))
        cs.lookup(pairs)
    assert cs.degree() == 5 and len(cs.lookups) == 11
    u = cs.usable_rows(k)
    assert u >= 260, "the 257-row byte table needs k >= 9"
    rng = np.random.default_rng(seed)
    one = F.fr_mont(1)
    small = F.fr_mont_array(list(range(1 << 18)))                    # Montgomery forms of every small value used below

    def col_of(values_by_row):                                        # dict / (rows, values) of small ints -> (n, 4) Montgomery column, zero elsewhere
        colm = np.zeros((n, 4), dtype=np.uint64)
        rows, vals = values_by_row
        colm[np.asarray(rows, dtype=np.int64)] = small[np.asarray(vals, dtype=np.int64)]
        return colm

    # -- fixed columns -----------------------------------------------------------------------------------------------------------------------------------
    def table_col(assigned):                                          # halo2 fills the unassigned rows of a table column with its first value
        colm = np.repeat(small[assigned[0]][None, :], n, axis=0)
        colm[: len(assigned)] = small[np.asarray(assigned, dtype=np.int64)]
        return np.ascontiguousarray(colm)
    chars = [61] + [(v + 65 if v < 26 else v + 71 if v < 52 else v - 4 if v < 62 else (43 if v == 62 else 47)) for v in range(64)]
    fixed = [None] * NF
    fixed[T_CHAR] = table_col(chars)
    fixed[T_VENC] = table_col([0] + list(range(64)))
    fixed[T_VDEC] = table_col([0] + list(range(256)))
    for c in range(4):
        fixed[T_BITS[c]] = table_col([0] + [(i >> (2 * c)) % 4 for i in range(256)])
    b64_len = min(B64_LEN, (u // 4) * 4)
    fixed[Q_DECODE] = col_of((np.arange(0, b64_len, 4), np.ones(b64_len // 4, dtype=np.int64)))
    # -- the base64 witness: a synthetic 1696-character string (random bytes; the reference's is the leaf PCK certificate, :769) ---------------------------
    raw = rng.integers(0, 256, size=b64_len * 3 // 4, dtype=np.uint8).tobytes()
    text = base64.b64encode(raw)
    assert len(text) == b64_len
    advice = [None] * A
    advice[ENC] = col_of((np.arange(b64_len), np.frombuffer(text, dtype=np.uint8).astype(np.int64)))
    vals6 = np.array([_b64_value(ch) for ch in text], dtype=np.int64)
    for c in range(12):                                               # column (i % 4) * 3 + j at row i - i % 4 holds (value >> ((2 - j) * 2)) % 4, :308-316
        i_mod, j = divmod(c, 3)
        idx = np.arange(i_mod, b64_len, 4)
        advice[BITS[c]] = col_of((idx - i_mod, (vals6[idx] >> ((2 - j) * 2)) % 4))
    dec = np.frombuffer(raw, dtype=np.uint8).astype(np.int64)
    di = np.arange(dec.size)
    advice[DEC_NOGAP] = col_of((di, dec))
    advice[DEC] = col_of((di + di // 3, dec))
    # -- the chips' part (estimate): Fp gate column dense with 88-bit limbs, the rest on the first quarter of the rows --------------------------------------
    R2 = F.limbs(pow(1 << 256, 2, F.R_MOD))

    def limbs_88(count):
        a = rng.integers(0, 1 << 64, size=(count, 4), dtype=np.uint64)
        a[:, 1] &= np.uint64((1 << 24) - 1)
        a[:, 2:] = 0
        return a

    def gate_column(nblk):
        """blocks (a, b, c, d = a + b*c) with 88-bit a, b, c on the first 4 * nblk rows, zeros after"""
        colm = np.zeros((n, 4), dtype=np.uint64)
        if nblk:
            da, db, dc = (be.to_device(limbs_88(nblk)) for _ in range(3))
            for d_ in (da, db, dc):
                be.fr_scale_dev(d_, R2, d_, nblk)                     # canonical -> Montgomery
            a_, b_, c_ = (d_.download((nblk, 4)).copy() for d_ in (da, db, dc))
            be.fr_mul_dev(db, dc, dc, nblk)
            be.fr_add_dev(da, dc, dc, nblk)
            d4 = dc.download((nblk, 4)).copy()
            for d_ in (da, db, dc):
                d_.free()
            blk = colm[: 4 * nblk].reshape(nblk, 4, 4)
            blk[:, 0], blk[:, 1], blk[:, 2], blk[:, 3] = a_, b_, c_, d4
        return colm

    def selector(nblk):
        return col_of((np.arange(nblk, dtype=np.int64) * 4, np.ones(nblk, dtype=np.int64)))
    nblk_full, quarter = u // 4, max(4, (u // 4) // 4 * 4)
    advice[FP_GATE] = gate_column(nblk_full)
    fixed[Q_FP] = selector(nblk_full)
    t18, t16 = min(1 << 18, u), min(1 << 16, u)
    advice[FP_LOOK] = small[rng.integers(0, t18, size=n)]
    fixed[T18] = col_of((np.arange(t18), np.arange(t18)))
    for col, qc in zip(R_GATE, Q_R):
        advice[col] = gate_column(quarter // 4)
        fixed[qc] = selector(quarter // 4)
    w = np.zeros(n, dtype=np.int64)
    w[:quarter] = rng.integers(0, t16, size=quarter)
    advice[R_LOOK] = small[w]
    fixed[T16] = col_of((np.arange(t16), np.arange(t16)))
    fixed[T_DENSE] = fixed[T16]
    spread = np.zeros(t16, dtype=np.int64)                            # spread(x): the bits of x interleaved with zeros (SHA-256 chips' table), kept below 2^18 here
    for bit in range(9):
        spread |= ((np.arange(t16) >> bit) & 1) << (2 * bit)
    fixed[T_SPREAD] = col_of((np.arange(t16), spread))
    for dcol, scol in ((SHA[0], SHA[2]), (SHA[1], SHA[3])):
        w = np.zeros(n, dtype=np.int64)
        w[:quarter] = rng.integers(0, t16, size=quarter)
        advice[dcol], advice[scol] = small[w], small[spread[w]]
    const_rows = 64
    for ccol in (C_FP, C_R):
        fixed[ccol] = col_of((np.arange(const_rows), rng.integers(0, 1 << 16, size=const_rows)))
    fixed = [f if f is not None else np.zeros((n, 4), dtype=np.uint64) for f in fixed]
    # -- copy constraints ------------------------------------------------------------------------------------------------------------------------------------
    asm = plonk.Assembly(cs, k)
    if k <= 10:
        asm.copies = []
    for i in di.tolist():                                             # decoded_chars_without_gap[i] -> decoded_chars[i + i / 3], :281-292 (different rows: scalar copies)
        asm.copy((plonk.ADVICE, DEC_NOGAP, i), (plonk.ADVICE, DEC, i + i // 3))
    return cs, fixed, asm, advice
