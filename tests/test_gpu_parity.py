"""GPU parity proper: the HIP kernels, called through the product C ABI (libzkmi355.so), against
the CPU oracle on the same seeded inputs — bit-exact (integer arithmetic).  Larger sizes are
checked through size-independent properties (closed-form MSM, NTT round trips)."""
import numpy as np
import pytest

import parity_cases as pc
import zk_dcap_verifier_amd as z

pytestmark = pytest.mark.gpu


def test_product_library_is_the_hip_build(gpu):
    assert "gfx950" in gpu.version()


def test_vec_ops(gpu, orc, pyref):
    pc.check_vec_ops(gpu, orc, pyref, 100003)


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 8, 9, 11, 13, 16, 17])
def test_ntt_vs_oracle(gpu, orc, pyref, log_n):
    pc.check_ntt(gpu, orc, pyref, log_n)


@pytest.mark.parametrize("tile,radix", [(10, 5), (11, 11), (12, 10), (12, 6)])
def test_ntt_other_plans(gpu, orc, pyref, tile, radix):
    gpu.tune(ntt_tile_log=tile, ntt_max_radix_log=radix)
    try:
        for log_n in (4, 10, 14, 16):
            pc.check_ntt(gpu, orc, pyref, log_n, seed=log_n)
    finally:
        gpu.tune(ntt_tile_log=10, ntt_max_radix_log=8)


def test_ntt_two_level_twiddle_path(gpu, orc, pyref):
    gpu.tune(ntt_full_twiddle_max_log=0)
    try:
        for log_n in (9, 14, 17):
            pc.check_ntt(gpu, orc, pyref, log_n, seed=100 + log_n)
    finally:
        gpu.tune(ntt_full_twiddle_max_log=24)


@pytest.mark.parametrize("j,k", [(4, 10), (5, 12), (3, 9), (9, 8), (2, 11)])
def test_domain_ops(gpu, orc, pyref, j, k):
    pc.check_domain(gpu, orc, pyref, j, k)


@pytest.mark.parametrize("j,k,count", [(5, 12, 9), (4, 6, 3), (3, 14, 4)])
def test_domain_batch(gpu, orc, pyref, j, k, count):
    pc.check_domain_batch(gpu, orc, pyref, j, k, count)


@pytest.mark.parametrize("n", [1, 2, 7, 100, 1000, 4096, 20000])
def test_msm_uniform_vs_oracle(gpu, orc, pyref, n):
    pc.check_msm(gpu, orc, pyref, n, seed=n)


@pytest.mark.parametrize("two_level", [0, 1])
@pytest.mark.parametrize("n,c,kind", [(1, 0, "uniform"), (1000, 0, "uniform"), (3000, 8, "uniform"), (20000, 16, "uniform"), (40000, 0, "uniform"),
                                      (5000, 0, "ones"), (5000, 0, "witness"), (5000, 16, "minus_one"), (5000, 0, "zeros")])
def test_msm_both_sorts(gpu, orc, pyref, n, c, kind, two_level):
    """one-level (small / batched) and two-level (large) bucket sort, each forced, against the oracle"""
    gpu.tune(msm_c=c, msm_two_level_sort=two_level, msm_bsort_chunk=512 if n == 5000 else 8192)   # (two-level: the degenerate columns with bins of several chunks)
    try:
        pc.check_msm(gpu, orc, pyref, n, seed=n + 1, kind=kind)
        if n == 3000:
            pc.check_msm_batch(gpu, orc, pyref, 2000, 7, device=True)
    finally:
        gpu.tune(msm_c=0, msm_two_level_sort=0, msm_bsort_chunk=8192)


@pytest.mark.parametrize("c", [3, 8, 11, 13, 16])
def test_msm_window_sizes(gpu, orc, pyref, c):
    gpu.tune(msm_c=c)
    try:
        pc.check_msm(gpu, orc, pyref, 3000, seed=c)
    finally:
        gpu.tune(msm_c=0)


@pytest.mark.parametrize("n,c,kind", [(3000, 17, "uniform"), (40000, 19, "uniform"), (20000, 22, "uniform"), (5000, 20, "minus_one"), (5000, 22, "witness"), (5000, 21, "ones"),
                                      (1, 22, "uniform"), (5000, 18, "zeros")])
def test_msm_wide_windows(gpu, orc, pyref, n, c, kind):
    """windows wider than 16 bits (c = 17 .. 22: what a 2^24 MSM runs with) take the two-level sort with up to 2048 bins of up to 1024 buckets"""
    gpu.tune(msm_c=c)
    try:
        pc.check_msm(gpu, orc, pyref, n, seed=n + c, kind=kind)
        if n == 3000:
            pc.check_msm_batch(gpu, orc, pyref, 2000, 5, device=True)
        gpu.tune(msm_bsort_chunk=256)                      # bins cut into several chunks (the short top window fills the lowest bins)
        pc.check_msm(gpu, orc, pyref, n, seed=n + c + 1, kind=kind)
    finally:
        gpu.tune(msm_c=0, msm_bsort_chunk=8192)


@pytest.mark.parametrize("kind", ["ones", "zeros", "witness", "minus_one"])
def test_msm_degenerate_scalar_columns(gpu, orc, pyref, kind):
    pc.check_msm(gpu, orc, pyref, 5000, seed=9, kind=kind)


def test_msm_repeated_and_identity_bases(gpu, orc, pyref):
    pc.check_msm(gpu, orc, pyref, 2000, seed=10, repeat_bases=True, with_identity=True)


def test_msm_prefix_of_resident_table(gpu, orc, pyref):
    pc.check_msm_prefix_and_handle(gpu, orc, pyref, 3000)


@pytest.mark.parametrize("n,count,device", [(3000, 7, False), (20000, 25, True), (1, 3, False)])
def test_msm_batch(gpu, orc, pyref, n, count, device):
    pc.check_msm_batch(gpu, orc, pyref, n, count, device=device)


def test_fixed_base_mul(gpu, orc, pyref):
    pc.check_fixed_base(gpu, orc, pyref, 5000)


def test_msm_closed_form_2p18(gpu, orc, pyref):
    """Size-independent property (SURVEY 8d cfg 3): bases P_i = [k_i]G built on the GPU, so
    MSM(s, P) must equal [sum s_i k_i mod r] G — one scalar multiplication on the oracle."""
    n = 1 << 18
    ks = pc.rand_fr(orc, pyref, n, 71)
    sc = pc.rand_fr(orc, pyref, n, 72)
    dk, dpts = gpu.to_device(ks), gpu.alloc(n * 64)
    gpu.g1_fixed_base_mul(dk, n, dpts)
    h = gpu.bases_register((dpts, n))
    got = gpu.msm(h, sc)
    ki = np.array(orc.fr_to_ints(ks), dtype=object)
    si = np.array(orc.fr_to_ints(sc), dtype=object)
    total = int((ki * si).sum() % pyref.R)
    want = orc.g1_to_affine(orc.g1_mul(orc.g1_generator(), orc.fr_from_ints([total])[0]))[0]
    assert (got[:8] == want).all()
    # linearity: MSM(2s) = 2 MSM(s)
    two = orc.fr_from_ints([2])
    sc2 = orc.fr_mul(sc, np.repeat(two, n, axis=0))
    got2 = gpu.msm(h, sc2)
    want2 = orc.g1_to_affine(orc.g1_mul(orc.g1_generator(), orc.fr_from_ints([2 * total % pyref.R])[0]))[0]
    assert (got2[:8] == want2).all()
    gpu.bases_release(h)
    dk.free()
    dpts.free()


def test_ntt_roundtrip_2p20_and_spot_values(gpu, orc, pyref):
    log_n = 20
    n = 1 << log_n
    a = pc.rand_fr(orc, pyref, n, 81)
    w = pyref.omega(log_n)
    d = gpu.to_device(a)
    gpu.ntt_dev(d, log_n, orc.fr_from_ints([w])[0])
    out = d.download((n, 4))
    # oracle on the full size takes ~1 s: compare everything
    assert (out == orc.best_fft(a, orc.fr_from_ints([w])[0], log_n)).all()
    gpu.ntt_dev(d, log_n, orc.fr_from_ints([pow(w, -1, pyref.R)])[0])
    back = d.download((n, 4))
    assert (back == orc.fr_mul(a, np.repeat(orc.fr_from_ints([n]), n, axis=0))).all()
    d.free()


def test_concurrent_callers(gpu, orc, pyref):
    pc.check_concurrent_callers(gpu, orc, pyref, n=5000, threads=6)


def test_msm_linearity_and_ntt_roundtrip_at_bench_sizes(gpu, orc, pyref):
    """BASELINE sizes the oracle cannot reach in seconds, through size-independent properties:
    MSM(s1) + MSM(s2) = MSM(s1 + s2) at 2^22 (the 2^24 bench path differs only in n), and iNTT(NTT(a)) = n*a at 2^23."""
    n = 1 << 22
    ks = pc.rand_fr(orc, pyref, n, 91)
    dk, dpts = gpu.to_device(ks), gpu.alloc(n * 64)
    gpu.g1_fixed_base_mul(dk, n, dpts)
    h = gpu.bases_register((dpts, n))
    dpts.free()
    s1, s2 = pc.rand_fr(orc, pyref, n, 92), pc.rand_fr(orc, pyref, n, 93)
    d1, d2, d3 = gpu.to_device(s1), gpu.to_device(s2), gpu.alloc(n * 32)
    gpu.fr_add_dev(d1, d2, d3, n)
    p1, p2 = gpu.msm_partial(h, d1, n), gpu.msm_partial(h, d2, n)
    both = gpu.g1_sum_xyzz(np.stack([p1, p2]))
    direct = gpu.msm(h, d3, n)
    assert (both == direct).all() and not (direct == 0).all()
    # spot check against the oracle on a prefix (the table prefix semantics of commit())
    m = 1 << 12
    pref = gpu.msm(h, d1, m)
    bases_prefix = np.empty((m, 8), dtype=np.uint64)
    g = orc.g1_generator()
    want = orc.g1_to_affine(orc.g1_mul(g, orc.fr_from_ints([sum(a * b for a, b in zip(orc.fr_to_ints(ks[:m]), orc.fr_to_ints(s1[:m]))) % pyref.R])[0]))[0]
    assert (pref[:8] == want).all()
    gpu.bases_release(h)
    for d in (dk, d1, d2, d3):
        d.free()
    log_n = 23
    nn = 1 << log_n
    a = pc.rand_fr(orc, pyref, nn, 94)
    d = gpu.to_device(a)
    w = pyref.omega(log_n)
    gpu.ntt_dev(d, log_n, orc.fr_from_ints([w])[0])
    gpu.ntt_dev(d, log_n, orc.fr_from_ints([pow(w, -1, pyref.R)])[0])
    back = d.download((nn, 4))
    idx = np.random.default_rng(5).integers(0, nn, size=4096)
    assert (back[idx] == orc.fr_mul(a[idx], np.repeat(orc.fr_from_ints([nn]), idx.size, axis=0))).all()
    d.free()


def test_run_length_msm_gpu(gpu, orc, pyref):
    from test_emu_kernels import _check_run_length_msm
    _check_run_length_msm(gpu, orc, pyref, 40000)


def test_ntt_with_two_level_inter_pass_twiddles(gpu, orc, pyref):
    """ntt_full_twiddle_max_log = 0: inter-pass twiddles from the two-level power tables (the path of transforms above 2^24)"""
    gpu.tune(ntt_full_twiddle_max_log=0)
    try:
        for log_n in (10, 14, 17):
            pc.check_ntt(gpu, orc, pyref, log_n, seed=7 + log_n)
        pc.check_domain(gpu, orc, pyref, 5, 12)
    finally:
        gpu.tune(ntt_full_twiddle_max_log=24)
