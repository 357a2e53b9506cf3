"""Kernel index logic on CPU: the product's kernel SOURCES compiled against the test-only emulator
(tests/csrc/emu_rt.h) and compared with the oracle.  This is not a product path — see rt.h."""
import pytest

import parity_cases as pc


def test_vec_ops(emu, orc, pyref):
    pc.check_vec_ops(emu, orc, pyref, 77)


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 4, 5, 7, 9, 10, 12])
def test_ntt(emu, orc, pyref, log_n):
    pc.check_ntt(emu, orc, pyref, log_n)


@pytest.mark.parametrize("tile,radix", [(4, 2), (5, 5), (8, 3)])
def test_ntt_other_plans(emu, orc, pyref, tile, radix):
    emu.tune(ntt_tile_log=tile, ntt_max_radix_log=radix)
    try:
        for log_n in (3, 6, 8):
            pc.check_ntt(emu, orc, pyref, log_n, seed=log_n)
    finally:
        emu.tune(ntt_tile_log=6, ntt_max_radix_log=4)


def test_ntt_two_level_twiddle_path(emu, orc, pyref):
    emu.tune(ntt_full_twiddle_max_log=0)          # the path used above 2^24: twiddles from the two-level tables
    try:
        for log_n in (5, 9, 12):
            pc.check_ntt(emu, orc, pyref, log_n, seed=100 + log_n)
    finally:
        emu.tune(ntt_full_twiddle_max_log=24)


@pytest.mark.parametrize("j,k", [(4, 5), (5, 6), (3, 4), (9, 3), (2, 4)])
def test_domain(emu, orc, pyref, j, k):
    pc.check_domain(emu, orc, pyref, j, k)


def test_batched_transform_in_slices_of_columns(emu, orc, pyref):
    """above the workspace budget (ntt_ws_limit_mb; k >= 22 in production) a batch is transformed in slices of columns: same results"""
    emu.tune(ntt_ws_limit_mb=1)
    try:
        pc.check_domain_batch(emu, orc, pyref, 4, 10, 11)           # 2^12-point extended columns of 128 KiB: eight fit, eleven go in two slices
    finally:
        emu.tune(ntt_ws_limit_mb=24576)


def test_domain_batch(emu, orc, pyref):
    pc.check_domain_batch(emu, orc, pyref, 4, 5, 3)
    pc.check_domain_batch(emu, orc, pyref, 3, 2, 2)          # single-pass transforms, batched


@pytest.mark.parametrize("n,c", [(1, 0), (2, 0), (5, 0), (33, 0), (200, 5), (300, 7), (700, 0), (300, 16)])      # c = 16: the half-limb digit path of the production sizes
def test_msm_uniform(emu, orc, pyref, n, c):
    emu.tune(msm_c=c)
    try:
        pc.check_msm(emu, orc, pyref, n, seed=n)
    finally:
        emu.tune(msm_c=0)


@pytest.mark.parametrize("two_level", [0, 1])
@pytest.mark.parametrize("n,c,kind", [(1, 0, "uniform"), (33, 0, "uniform"), (300, 7, "uniform"), (700, 0, "uniform"), (513, 9, "uniform"),
                                      (257, 0, "ones"), (257, 0, "witness"), (257, 0, "minus_one"), (64, 0, "zeros"), (257, 16, "minus_one"), (130, 16, "uniform")])
def test_msm_both_sorts(emu, orc, pyref, n, c, kind, two_level):
    """The pairs are grouped by bucket either by the one-level counting sort (small / batched inputs) or the two-level one (large inputs):
    force each and compare with the oracle; a batch too."""
    emu.tune(msm_c=c, msm_two_level_sort=two_level, msm_bsort_chunk=64 if n in (257, 513) else 8192)   # (two-level: some cases with bins of several chunks)
    try:
        pc.check_msm(emu, orc, pyref, n, seed=n + 1, kind=kind)
        if n == 300:
            pc.check_msm_batch(emu, orc, pyref, 120, 6)
    finally:
        emu.tune(msm_c=0, msm_two_level_sort=0, msm_bsort_chunk=8192)


@pytest.mark.parametrize("n,c,kind", [(300, 17, "uniform"), (700, 18, "uniform"), (257, 17, "minus_one"), (257, 18, "witness"), (64, 17, "zeros")])
def test_msm_wide_windows(emu, orc, pyref, n, c, kind):
    """windows wider than 16 bits (c = 17 .. 22, the large-n plan: fewer windows per scalar for more buckets) take the two-level sort with bins of 2^8 .. 2^10 buckets"""
    emu.tune(msm_c=c)
    try:
        pc.check_msm(emu, orc, pyref, n, seed=n + c, kind=kind)
        if n == 300:
            pc.check_msm_batch(emu, orc, pyref, 120, 3)
        emu.tune(msm_bsort_chunk=64)                       # bins cut into several chunks (the short top window fills the lowest bin)
        pc.check_msm(emu, orc, pyref, n, seed=n + c + 1, kind=kind)
    finally:
        emu.tune(msm_c=0, msm_bsort_chunk=8192)


@pytest.mark.parametrize("kind", ["ones", "zeros", "witness", "minus_one"])
def test_msm_degenerate_scalar_columns(emu, orc, pyref, kind):
    pc.check_msm(emu, orc, pyref, 257, seed=9, kind=kind)          # one heavy bucket -> several merge rounds


def test_msm_repeated_and_identity_bases(emu, orc, pyref):
    pc.check_msm(emu, orc, pyref, 100, seed=10, repeat_bases=True, with_identity=True)


def test_msm_prefix_of_resident_table(emu, orc, pyref):
    pc.check_msm_prefix_and_handle(emu, orc, pyref, 90)


def test_msm_batch(emu, orc, pyref):
    pc.check_msm_batch(emu, orc, pyref, 120, 6)
    pc.check_msm_batch(emu, orc, pyref, 50, 2, seed=3, device=True)


def test_fixed_base_mul(emu, orc, pyref):
    pc.check_fixed_base(emu, orc, pyref, 40)


def test_error_paths(emu, orc, pyref):
    import numpy as np
    import zk_dcap_verifier_amd as z
    with pytest.raises(z.ZkError):
        emu.msm(12345, np.zeros((1, 4), dtype=np.uint64))           # unknown handle
    with pytest.raises(z.ZkError):
        emu.tune(no_such_key=1)
    sc, bases = pc.msm_inputs(orc, pyref, 8, 1)
    h = z.arithmetic.BasesHandle(emu, bases)
    with pytest.raises(z.ZkError):
        emu.msm(h.handle, np.zeros((9, 4), dtype=np.uint64))        # more scalars than bases
    assert (emu.msm(h.handle, np.zeros((0, 4), dtype=np.uint64)) == 0).all()   # empty MSM = identity
    h.release()
    with pytest.raises(AssertionError):
        z.arithmetic.best_multiexp(sc[:3], bases)                   # halo2: assert_eq!(coeffs.len(), bases.len())


def test_concurrent_callers(emu, orc, pyref):
    pc.check_concurrent_callers(emu, orc, pyref, n=60, threads=3)


def test_shared_base_table_across_contexts(built, orc, pyref):
    """zk_bases_share: a second context on the same device commits against the FIRST context's expanded table (no second HBM copy), and the table
    outlives the release of the owner's handle while the borrower still holds it."""
    import numpy as np
    import zk_dcap_verifier_amd as z
    import parity_cases as pc
    from conftest import EMU_SO
    a, b = z.Backend(0, lib_path=EMU_SO), z.Backend(0, lib_path=EMU_SO)
    for be in (a, b):
        be.tune(msm_sort_threads=64, msm_sort_wgs=3, msm_block=32, msm_target_threads=64, msm_min_chunk=2, vec_block=32)
    sc, bases = pc.msm_inputs(orc, pyref, 300, 5)
    want = orc.g1_to_affine(orc.best_multiexp(sc, bases))[0]
    own = z.arithmetic.BasesHandle(a, bases)
    lent = z.arithmetic.BasesHandle.shared(b, own)
    assert (z.arithmetic.best_multiexp(sc, lent)[:8] == want).all()
    own.release()                                                   # the borrower keeps the memory alive
    assert (z.arithmetic.best_multiexp(sc, lent)[:8] == want).all()
    lent.release()
    a.close()
    b.close()


def _check_run_length_msm(be, orc, pyref, n, seed=3):
    """zk_bases_enable_runs: columns made of long runs of equal values (a sorted lookup column, a constant column, a column with a few distinct
    full-width values) are committed through their adjacent differences against the prefix-sum table — same points as the oracle's best_multiexp,
    also for a prefix of the table and next to ordinary columns in the same batch"""
    import random
    import numpy as np
    import zk_dcap_verifier_amd as z
    import parity_cases as pc
    rnd = random.Random(seed)
    _, bases = pc.msm_inputs(orc, pyref, n, seed)
    bases[5] = 0                                                    # an identity base inside a run
    h = z.arithmetic.BasesHandle(be, bases).enable_runs()
    R = pyref.R
    big = [rnd.randrange(R) for _ in range(8)]
    sorted_col = sorted(rnd.choice(big) for _ in range(n))                            # sorted lookup input: 8 runs
    const_col = [big[0]] * (n - 3) + [rnd.randrange(R) for _ in range(3)]             # constant + blinding-like tail
    steps = [big[(i * 7 // n) % 8] if i % 97 else 0 for i in range(n)]                # runs broken by zeros
    rand_col = [rnd.randrange(R) for _ in range(n)]                                   # no runs: stays on the direct path
    cols = [orc.fr_from_ints(c) for c in (sorted_col, const_col, steps, rand_col)]
    be.timing(True)
    be.tune(msm_runs=2)                                              # (below ~4 M saved additions the default keeps small batches on the direct path)
    got = z.arithmetic.best_multiexp_batch(cols, h)
    for i, c in enumerate(cols):
        want = orc.g1_to_affine(orc.best_multiexp(c, bases))[0]
        assert (got[i, :8] == want).all(), i
    assert be.stat_get("msm_run_columns") >= 2
    m = n - n // 3                                                   # commit() of a shorter polynomial: prefix of the table
    got = z.arithmetic.best_multiexp(cols[1][:m], h)
    assert (got[:8] == orc.g1_to_affine(orc.best_multiexp(cols[1][:m], bases[:m]))[0]).all()
    be.tune(msm_runs=0)                                              # the switch: everything direct, same answers
    try:
        got2 = z.arithmetic.best_multiexp_batch(cols, h)
        assert (got2 == z.arithmetic.best_multiexp_batch(cols, h)).all() and (got2[0, :8] == orc.g1_to_affine(orc.best_multiexp(cols[0], bases))[0]).all()
    finally:
        be.tune(msm_runs=1)
        be.timing(False)
    h.release()


def test_emulated_run_length_msm(emu, orc, pyref):
    _check_run_length_msm(emu, orc, pyref, 2500)


def test_ntt_with_two_level_inter_pass_twiddles(emu, orc, pyref):
    """ntt_full_twiddle_max_log = 0: the strided passes take their inter-pass twiddles from the two-level power tables (what transforms above 2^24 do) instead of a full table"""
    emu.tune(ntt_full_twiddle_max_log=0)
    try:
        for log_n in (5, 9, 12):
            pc.check_ntt(emu, orc, pyref, log_n, seed=7 + log_n)
        pc.check_domain(emu, orc, pyref, 9, 3)
    finally:
        emu.tune(ntt_full_twiddle_max_log=24)
