"""permute_expression_pair: oracle restatement vs a direct Python transcription of the rule, product vs oracle."""
import random

import numpy as np
import pytest

import parity_cases as pc
import zk_dcap_verifier_amd as z


def _case(orc, pyref, k, seed, kind):
    rnd = random.Random(seed)
    n, bf, R = 1 << k, 5, pyref.R
    u = n - bf - 1
    if kind == "small":                      # table = small range, inputs drawn from it (typical range-check lookup)
        tab_vals = [rnd.randrange(0, max(4, u // 3)) for _ in range(u)]
        tab_vals[: max(4, u // 3)] = list(range(max(4, u // 3)))[: len(tab_vals[: max(4, u // 3)])]
        inp_vals = [rnd.choice(tab_vals) for _ in range(u)]
    elif kind == "wide":                     # full-width field elements: every radix pass matters
        tab_vals = [rnd.randrange(R) for _ in range(u)]
        inp_vals = [rnd.choice(tab_vals[: max(2, u // 2)]) for _ in range(u)]
    elif kind == "window_ties":              # full-width keys that agree on the top 64 bits in use and differ far below: the 64-bit window
        big = (1 << 252) + (0x1234567 << 200)  # sort cannot order them, the order check must send the call down the every-digit path
        tab_vals = [big + rnd.randrange(1 << 40) for _ in range(u)]
        inp_vals = [rnd.choice(tab_vals[: max(2, u // 2)]) for _ in range(u)]
    elif kind == "theta_last":               # theta-compressed tuples whose LAST expression differs (the base64 lookups of the sgx circuit,
        theta = rnd.randrange(R)              # sgx_dcap_verifier.rs:216-236: a character and three 2-bit chunks): v and v + small tie on the window
        rows = [(61 if r > 64 else 65 + r, (r >> 4) % 4, (r >> 2) % 4, r % 4) if r < 257 else (61, 0, 0, 0) for r in range(u)]
        comp = lambda e: (((e[0] * theta + e[1]) * theta + e[2]) * theta + e[3]) % R
        tab_vals = [comp(e) for e in rows]
        inp_vals = [comp(rows[rnd.randrange(min(64, u))]) if i % 4 == 0 and i < u // 2 else comp((65, 0, 0, 0)) for i in range(u)]
    elif kind == "two_stage_ties":           # window ties that differ in two bit ranges more than 64 bits apart: two refinement stages
        tab_vals = [(5 << 250) + ((i % 7) << 130) + (((i * 31) % 11) << 20) for i in range(u)]
        rnd.shuffle(tab_vals)
        inp_vals = [rnd.choice(tab_vals) for _ in range(u)]
    elif kind == "identical":                # all inputs equal
        tab_vals = [7] + [rnd.randrange(R) for _ in range(u - 1)]
        inp_vals = [7] * u
    else:                                    # permutation: input == table multiset, no repeats needed from leftovers
        tab_vals = [rnd.randrange(R) for _ in range(u)]
        inp_vals = list(tab_vals)
        rnd.shuffle(inp_vals)
    pad = [rnd.randrange(R) for _ in range(bf + 1)]
    return inp_vals + pad, tab_vals + pad[::-1], bf


def test_oracle_lookup_permute_matches_rule(orc, pyref):
    for kind in ("small", "wide", "identical", "perm"):
        inp, tab, bf = _case(orc, pyref, 6, 5, kind)
        n = 64
        u = n - bf - 1
        M = orc.fr_from_ints
        bi, bt = M(list(range(100, 100 + bf + 1))), M(list(range(200, 200 + bf + 1)))
        oi, ot = orc.lookup_permute(M(inp), M(tab), 6, bf, bi, bt)
        got_i, got_t = orc.fr_to_ints(oi), orc.fr_to_ints(ot)
        pin = sorted(inp[:u])
        left = sorted(tab[:u])
        ptab, rep = [None] * u, []
        for r in range(u):
            if r == 0 or pin[r] != pin[r - 1]:
                ptab[r] = pin[r]
                left.remove(pin[r])
            else:
                rep.append(r)
        for v in left:
            ptab[rep.pop()] = v
        assert got_i == pin + list(range(100, 100 + bf + 1)) and got_t == ptab + list(range(200, 200 + bf + 1)), kind
    with pytest.raises(ValueError):
        bad = inp[:]
        bad[3] = 1 << 200
        orc.lookup_permute(M(bad), M(tab), 6, bf, bi, bt)


def _check(be, orc, pyref, k, kind, seed):
    inp, tab, bf = _case(orc, pyref, k, seed, kind)
    n = 1 << k
    M = orc.fr_from_ints
    bi, bt = pc.rand_fr(orc, pyref, bf + 1, seed + 1), pc.rand_fr(orc, pyref, bf + 1, seed + 2)
    A, T = M(inp), M(tab)
    want_i, want_t = orc.lookup_permute(A, T, k, bf, bi, bt)
    da, dt = be.to_device(A), be.to_device(T)
    oa, ot = z.permutation.permute_expression_pair(da, dt, k, bf, bi, bt, backend=be)
    assert (oa.download((n, 4)) == want_i).all() and (ot.download((n, 4)) == want_t).all(), kind
    for d in (da, dt, oa, ot):
        d.free()


@pytest.mark.parametrize("kind", ["small", "wide", "window_ties"])
def test_emulated_lookup_permute(emu, orc, pyref, kind):
    emu.tune(vec_block=64)
    try:
        _check(emu, orc, pyref, 5, kind, seed=3)      # (the emulator spawns a thread per work-item: keep it tiny)
    finally:
        emu.tune(vec_block=32)


def test_lookup_permute_rejects_value_outside_table(emu, orc, pyref):
    inp, tab, bf = _case(orc, pyref, 5, 1, "small")
    inp[2] = 1 << 100
    M = orc.fr_from_ints
    da, dt = emu.to_device(M(inp)), emu.to_device(M(tab))
    b = pc.rand_fr(orc, pyref, bf + 1, 9)
    with pytest.raises(z.ZkError):
        z.permutation.permute_expression_pair(da, dt, 5, bf, b, b, backend=emu)


@pytest.mark.gpu
@pytest.mark.parametrize("k,kind", [(7, "small"), (12, "wide"), (16, "small"), (14, "identical"), (15, "perm"), (13, "window_ties"), (18, "wide"), (19, "small"),
                                    (10, "theta_last"), (19, "theta_last"), (14, "two_stage_ties")])
def test_gpu_lookup_permute(gpu, orc, pyref, k, kind):
    gpu.timing(True)
    _check(gpu, orc, pyref, k, kind, seed=k)
    if kind in ("theta_last", "two_stage_ties", "window_ties"):       # window ties are refined by a few extra radix passes, not re-sorted digit by digit
        # (refined the first time a shape is seen; afterwards the context remembers which columns tie and sorts them in two stages straight away)
        assert gpu.stat_get("lookup_generic_sorts") == 0 and gpu.stat_get("lookup_refined_sorts") + gpu.stat_get("lookup_hinted_sorts") >= 1
    gpu.timing(False)


@pytest.mark.parametrize("kind", ["theta_last", "two_stage_ties", "window_ties"])
def test_emulated_lookup_permute_refines_window_ties(emu, orc, pyref, kind):
    emu.tune(vec_block=64)
    try:
        emu.timing(True)
        _check(emu, orc, pyref, 9, kind, seed=3)
        assert emu.stat_get("lookup_generic_sorts") == 0 and emu.stat_get("lookup_refined_sorts") + emu.stat_get("lookup_hinted_sorts") >= 1
    finally:
        emu.timing(False)
        emu.tune(vec_block=32)


@pytest.mark.gpu
def test_gpu_lookup_permute_generic_sort_path(gpu, orc, pyref):
    """the every-digit sort (taken when rows tie on the 64-bit window) forced on ordinary full-width data: same answer"""
    gpu.tune(lookup_force_generic_sort=1)
    try:
        _check(gpu, orc, pyref, 12, "wide", seed=5)
    finally:
        gpu.tune(lookup_force_generic_sort=0)


def test_emulated_lookup_tie_hint_is_only_a_hint(built, orc, pyref):
    """a context that learnt "this column ties" from one call must stay exact when the next call of the same shape has ties that differ HIGHER up than
    before (the order check sends it through the refinement again) or no ties at all"""
    from conftest import EMU_SO
    be = z.Backend(0, lib_path=EMU_SO)
    be.tune(vec_block=64)
    try:
        be.timing(True)
        _check(be, orc, pyref, 9, "theta_last", seed=3)                 # learns: ties in the lowest bits
        assert be.stat_get("lookup_refined_sorts") >= 1
        _check(be, orc, pyref, 9, "two_stage_ties", seed=4)             # same shape, ties now differ around bits 20 and 130
        _check(be, orc, pyref, 9, "wide", seed=5)                       # no ties: the extra low passes are harmless
        _check(be, orc, pyref, 9, "theta_last", seed=6)
        assert be.stat_get("lookup_generic_sorts") == 0 and be.stat_get("lookup_hinted_sorts") >= 1
    finally:
        be.close()


def _check_batch(be, orc, pyref, k, kinds, seed):
    """several lookups in one zk_lookup_permute_batch_dev call (narrow keys, full-width keys and window ties side by side)"""
    n = 1 << k
    M = orc.fr_from_ints
    ins, tabs, bis, bts, wants = [], [], [], [], []
    for j, kind in enumerate(kinds):
        inp, tab, bf = _case(orc, pyref, k, seed + j, kind)
        bi, bt = pc.rand_fr(orc, pyref, bf + 1, seed + 10 + j), pc.rand_fr(orc, pyref, bf + 1, seed + 20 + j)
        A, T = M(inp), M(tab)
        wants.append(orc.lookup_permute(A, T, k, bf, bi, bt))
        ins.append(be.to_device(A)); tabs.append(be.to_device(T)); bis.append(bi); bts.append(bt)
    outs = z.permutation.permute_expression_pairs(ins, tabs, k, bf, np.stack(bis), np.stack(bts), backend=be)
    for (oa, ot), (wi, wt), kind in zip(outs, wants, kinds):
        assert (oa.download((n, 4)) == wi).all() and (ot.download((n, 4)) == wt).all(), kind


def test_emulated_lookup_permute_batch(emu, orc, pyref):
    emu.tune(vec_block=64)
    try:
        _check_batch(emu, orc, pyref, 5, ["small", "window_ties", "wide"], seed=11)
    finally:
        emu.tune(vec_block=32)


@pytest.mark.gpu
def test_gpu_lookup_permute_batch(gpu, orc, pyref):
    _check_batch(gpu, orc, pyref, 13, ["small", "wide", "window_ties", "identical", "perm", "wide", "small"], seed=31)


def _check_shared_table(be, orc, pyref, k, seed):
    """three lookups against ONE table column (the same device buffer): the batch sorts it once; results per lookup as if it were private"""
    n = 1 << k
    M = orc.fr_from_ints
    _, tab, bf = _case(orc, pyref, k, seed, "wide")
    u = n - bf - 1
    import random
    rnd = random.Random(seed)
    T = M(tab)
    dT = be.to_device(T)
    ins, bis, bts, wants = [], [], [], []
    for j in range(3):
        inp = [rnd.choice(tab[:u]) for _ in range(u)] + [rnd.randrange(pyref.R) for _ in range(bf + 1)]
        bi, bt = pc.rand_fr(orc, pyref, bf + 1, seed + 10 + j), pc.rand_fr(orc, pyref, bf + 1, seed + 20 + j)
        A = M(inp)
        wants.append(orc.lookup_permute(A, T, k, bf, bi, bt))
        ins.append(be.to_device(A)); bis.append(bi); bts.append(bt)
    outs = z.permutation.permute_expression_pairs(ins, [dT, dT, dT], k, bf, np.stack(bis), np.stack(bts), backend=be)
    for (oa, ot), (wi, wt) in zip(outs, wants):
        assert (oa.download((n, 4)) == wi).all() and (ot.download((n, 4)) == wt).all()


def test_emulated_lookup_permute_shared_table(emu, orc, pyref):
    emu.tune(vec_block=64)
    try:
        _check_shared_table(emu, orc, pyref, 5, seed=41)
    finally:
        emu.tune(vec_block=32)


@pytest.mark.gpu
def test_gpu_lookup_permute_shared_table(gpu, orc, pyref):
    _check_shared_table(gpu, orc, pyref, 14, seed=43)
